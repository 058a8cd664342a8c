from . import hooks
from .defaults import DefaultPredictor, DefaultTrainer, TrainModel
from .hooks import HookBase
from .launch import launch

__all__ = ["DefaultPredictor", "DefaultTrainer", "TrainModel", "HookBase", "hooks", "launch"]
