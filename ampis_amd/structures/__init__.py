from .boxes import Boxes, BoxMode
from .instances import Instances
from .masks import BitMasks, PolygonMasks, RLEBitMasks

__all__ = ["Boxes", "BoxMode", "Instances", "BitMasks", "PolygonMasks", "RLEBitMasks"]
