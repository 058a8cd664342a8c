"""detectron2.engine.hooks.HookBase (ampis/data_utils.py:25,37): a hook with a `.trainer` back-reference and the
before/after callbacks of the train loop."""


class HookBase:
    trainer = None

    def before_train(self):
        pass

    def after_train(self):
        pass

    def before_step(self):
        pass

    def after_step(self):
        pass
