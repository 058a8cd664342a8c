"""DefaultPredictor / DefaultTrainer façade over the C-ABI model (notebook cells 22-28).

DefaultPredictor(cfg)(img) follows detectron2's DefaultPredictor.__call__: BGR uint8 HxWx3 -> ResizeShortestEdge
(INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST, PIL bilinear on uint8, exactly detectron2's CPU-side transform) -> the HIP hot path
(amp_model_infer) -> {'instances': Instances(image_size=(H, W), pred_boxes, scores, pred_classes, pred_masks)} rescaled to
the original image.  There is no CPU fallback: cfg.MODEL.DEVICE must name a HIP device.
"""
import logging
import os

import numpy as np
import torch

from .. import _lib, checkpoint, params as P
from ..model import MaskRCNN
from ..structures import Boxes, Instances, RLEBitMasks

logger = logging.getLogger("ampis_amd")


def read_image_bgr(path):
    """cv2.imread stand-in (notebook cells 26/28): 8-bit image -> HxWx3 BGR uint8 (grayscale replicated)."""
    from PIL import Image
    im = Image.open(str(path))
    im = im.convert("RGB")
    return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def resize_shortest_edge(img, min_size, max_size):
    """detectron2 ResizeShortestEdge.get_transform + ResizeTransform.apply_image (PIL bilinear on uint8)."""
    h, w = img.shape[:2]
    if min_size == 0:
        return img
    scale = min_size * 1.0 / min(h, w)
    newh, neww = (min_size, scale * w) if h < w else (scale * h, min_size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    neww, newh = int(neww + 0.5), int(newh + 0.5)
    if (newh, neww) == (h, w):
        return img
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize((neww, newh), Image.BILINEAR))


def _device_index(dev):
    dev = str(dev)
    if dev.startswith("cpu"):
        raise _lib.AmpError("cfg.MODEL.DEVICE='cpu': ampis_amd has no CPU path for the Mask R-CNN hot path "
                            "(the CPU restatement under oracle/ is test infrastructure only)")
    return int(dev.split(":")[1]) if ":" in dev else 0


class DefaultPredictor:
    def __init__(self, cfg):
        self.cfg = cfg.clone()
        if len(cfg.DATASETS.TEST) == 0:
            logger.warning("cfg.DATASETS.TEST is empty")
        assert cfg.INPUT.FORMAT in ("RGB", "BGR"), cfg.INPUT.FORMAT
        self.input_format = cfg.INPUT.FORMAT
        self.num_classes = int(cfg.MODEL.ROI_HEADS.NUM_CLASSES)
        self.ctx = _lib.Context(_device_index(cfg.MODEL.DEVICE))
        w = str(cfg.MODEL.WEIGHTS)
        if w:
            self.params = checkpoint.load_checkpoint(w, self.num_classes)
        else:
            logger.warning("cfg.MODEL.WEIGHTS is empty: using seeded random initialisation (like an un-loaded detectron2 model)")
            self.params = P.init_params(self.num_classes, seed=0, style="d2")
        self._model = None
        self._cap = (0, 0, 0)

    def _ensure(self, h, w, out_hw):
        hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
        if self._model is None or hp > self._cap[0] or wp > self._cap[1] or out_hw > self._cap[2]:
            if self._model is not None:
                self._model.close()
            c = self.cfg
            cap = (max(hp, self._cap[0]), max(wp, self._cap[1]), max(out_hw, self._cap[2], 64))
            self._model = MaskRCNN(self.ctx, self.num_classes, max_batch=1, max_h=cap[0], max_w=cap[1], max_out_hw=cap[2],
                                   detections_per_image=int(c.TEST.DETECTIONS_PER_IMAGE),
                                   pre_nms_topk=int(c.MODEL.RPN.PRE_NMS_TOPK_TEST), post_nms_topk=int(c.MODEL.RPN.POST_NMS_TOPK_TEST),
                                   rpn_nms_thresh=float(c.MODEL.RPN.NMS_THRESH), score_thresh=float(c.MODEL.ROI_HEADS.SCORE_THRESH_TEST),
                                   nms_thresh=float(c.MODEL.ROI_HEADS.NMS_THRESH_TEST), pixel_mean=tuple(c.MODEL.PIXEL_MEAN),
                                   pixel_std=tuple(c.MODEL.PIXEL_STD))
            self._model.load_params(self.params)
            self._cap = cap
        return self._model

    def __call__(self, original_image):
        original_image = np.asarray(original_image)
        assert original_image.ndim == 3 and original_image.shape[2] == 3 and original_image.dtype == np.uint8, \
            "DefaultPredictor expects an HxWx3 uint8 image"
        if self.input_format == "RGB":
            original_image = original_image[:, :, ::-1]
        height, width = original_image.shape[:2]
        image = resize_shortest_edge(np.ascontiguousarray(original_image), int(self.cfg.INPUT.MIN_SIZE_TEST),
                                     int(self.cfg.INPUT.MAX_SIZE_TEST))
        h, w = image.shape[:2]
        model = self._ensure(h, w, max(height, width))
        r = model.infer(np.ascontiguousarray(image)[None], out_sizes=[(height, width)])[0]
        inst = Instances((height, width))
        inst.pred_boxes = Boxes(torch.from_numpy(r["boxes"]))
        inst.scores = torch.from_numpy(r["scores"])
        inst.pred_classes = torch.from_numpy(r["classes"])
        inst.pred_masks = RLEBitMasks(r["masks"], (height, width))
        return {"instances": inst}


class DefaultTrainer:
    """Surface of detectron2's DefaultTrainer that AMPIS subclasses (ampis/data_utils.py:135-177; notebook cell 22).
    The training hot path (losses, backward, SGD, RCCL gradient all-reduce) is not built in this
    round: constructing the trainer works (cfg, hooks list, checkpoint naming), train() fails loudly."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.iter = self.start_iter = 0
        self.max_iter = int(cfg.SOLVER.MAX_ITER)
        self.model = None
        self._hooks = []
        self.register_hooks(self.build_hooks())

    def build_hooks(self):
        from .hooks import HookBase

        class _PeriodicWriter(HookBase):
            pass

        return [_PeriodicWriter()]

    def register_hooks(self, hooks):
        for h in hooks:
            if h is not None:
                h.trainer = self
        self._hooks.extend([h for h in hooks if h is not None])

    def resume_or_load(self, resume=True):
        self.start_iter = 0

    def train(self):
        raise NotImplementedError("ampis_amd: the MI355X training path (SURVEY.md §8a rows a18-a20) is not built yet; "
                                  "inference (DefaultPredictor) is.")
