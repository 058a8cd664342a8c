"""Host-side mirror of the AMPIS functions that sit directly on the hot path's output:
ampis/data_utils.py:255-279 `compress_pred`, :282-310 `format_outputs` -- same names, arguments, in-place mutation and
return values.  The reference RLE-encodes each N x H x W bool mask after a per-mask D2H copy; here the masks arrive from the
device already as COCO RLE, so compress_pred only unwraps them (bit-identical `counts` bytes)."""
import numpy as np

from . import rle as RLE
from .structures import RLEBitMasks


def compress_pred(pred):
    """pred: Instances from DefaultPredictor. Mutates and returns it: pred_masks -> list of RLE dicts, pred_boxes -> ndarray
    [N,4] f32, scores -> ndarray f32, pred_classes -> ndarray i64 (ampis/data_utils.py:275-278)."""
    m = pred.pred_masks
    if isinstance(m, RLEBitMasks):
        pred.pred_masks = list(m.rle)
    else:   # dense masks (e.g. produced elsewhere): the reference's path
        pred.pred_masks = [RLE.encode(np.asfortranarray(np.asarray(x.to("cpu").numpy() if hasattr(x, "to") else x))) for x in m]
    pred.pred_boxes = pred.pred_boxes.tensor.to("cpu").numpy()
    pred.scores = pred.scores.to("cpu").numpy()
    pred.pred_classes = pred.pred_classes.to("cpu").numpy()
    return pred


def format_outputs(filename, dataset, pred):
    """{'file_name', 'dataset', 'pred'} with pred['instances'] compressed in place (ampis/data_utils.py:305-310)."""
    compress_pred(pred["instances"])
    return {"file_name": filename, "dataset": dataset, "pred": pred}
