from . import hooks
from .defaults import DefaultPredictor, DefaultTrainer

__all__ = ["DefaultPredictor", "DefaultTrainer", "hooks"]
