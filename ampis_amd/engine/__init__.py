from . import hooks
from .defaults import DefaultPredictor, DefaultTrainer, TrainModel
from .hooks import HookBase

__all__ = ["DefaultPredictor", "DefaultTrainer", "TrainModel", "HookBase", "hooks"]
