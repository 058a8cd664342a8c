"""detectron2.utils.visualizer.Visualizer -- the members AMPIS calls (ampis/visualize.py:14,154-164,291,326-328):

    Visualizer(img, metadata=..., scale=1)
        .draw_dataset_dict(ddict)              ground truth of one dataset dict           -> VisImage
        .draw_instance_predictions(instances)  `pred_*` fields of a predictor's Instances  -> VisImage
        .overlay_instances(boxes=, masks=, labels=, assigned_colors=, alpha=)             -> VisImage
    VisImage.get_image() -> uint8 [H*scale, W*scale, 3]

Host-side drawing, outside the hot path (SURVEY §2.1 lists visualisation as out of scope); it exists so that `import ampis` and
the notebook cells that call `visualize.display_ddicts` / `display_iset` run on the façade.  Rendering is plain numpy (alpha-blended
mask fill, 1-px outline, box frame) plus PIL for label text; it does not try to be pixel-identical to detectron2's matplotlib canvas.
Masks arrive in whatever form the caller holds: COCO RLE dicts (what the MI355X path produces and `RLEMasks.rle` carries), bool
arrays [N, H, W], BitMasks, PolygonMasks / lists of polygons; they are decoded one at a time, never as one N x H x W block.
"""
import colorsys
from enum import Enum

import numpy as np


class ColorMode(Enum):
    IMAGE = 0
    SEGMENTATION = 1
    IMAGE_BW = 2


class VisImage:
    def __init__(self, img, scale=1.0):
        self.img = img                      # uint8 RGB, already at output scale
        self.scale = scale
        self.height, self.width = img.shape[:2]

    def get_image(self):
        return self.img

    def save(self, filepath):
        from PIL import Image
        Image.fromarray(self.img).save(str(filepath))


def _palette(n):
    """n visually distinct, reproducible RGB colours in [0, 1] (golden-ratio walk around the hue circle)."""
    return np.array([colorsys.hsv_to_rgb((0.11 + 0.61803398875 * i) % 1.0, 0.85, 1.0) for i in range(n)], dtype=np.float64).reshape(n, 3)


def _meta_get(metadata, key, default=None):
    if metadata is None:
        return default
    if isinstance(metadata, dict):
        return metadata.get(key, default)
    return getattr(metadata, key, default)


class Visualizer:
    def __init__(self, img_rgb, metadata=None, scale=1.0, instance_mode=ColorMode.IMAGE):
        img = np.asarray(img_rgb)
        if img.ndim == 2:
            img = img[:, :, None]
        if img.shape[2] == 1:
            img = np.repeat(img, 3, axis=2)
        self.img = np.clip(img[:, :, :3], 0, 255).astype(np.uint8)
        self.metadata = metadata
        self.scale = float(scale)
        self._instance_mode = instance_mode
        h, w = self.img.shape[:2]
        self._out_hw = (max(1, int(round(h * self.scale))), max(1, int(round(w * self.scale))))
        canvas = self.img
        if self._out_hw != (h, w):
            from PIL import Image
            canvas = np.asarray(Image.fromarray(self.img).resize((self._out_hw[1], self._out_hw[0]), Image.BILINEAR))
        self.output = VisImage(canvas.copy(), self.scale)

    # ---- inputs in their various forms -> per-instance bool masks at image resolution ----
    def _mask_list(self, masks):
        from .. import rle
        from ..structures import BitMasks, PolygonMasks
        h, w = self.img.shape[:2]
        if masks is None:
            return None
        if hasattr(masks, "rle") and not isinstance(masks, (list, tuple)):      # RLEBitMasks / RLEMasks-like
            masks = list(masks.rle)
        if isinstance(masks, BitMasks):
            masks = masks.tensor.numpy()
        if isinstance(masks, PolygonMasks):
            masks = masks.polygons
        if hasattr(masks, "detach"):                                             # torch tensor [N, H, W]
            masks = masks.detach().cpu().numpy()
        out = []
        for m in masks:
            if isinstance(m, dict):                                              # COCO RLE (compressed or uncompressed counts)
                out.append(lambda m=m: rle.decode(m).astype(bool))
            elif isinstance(m, (list, tuple)) or (isinstance(m, np.ndarray) and m.dtype != bool and m.ndim == 1):
                polys = m if isinstance(m, (list, tuple)) and len(m) and not np.isscalar(m[0]) else [m]
                polys = [np.asarray(p, dtype=np.float64).reshape(-1).tolist() for p in polys]
                out.append(lambda polys=polys: rle.decode(rle.merge(rle.frPyObjects(polys, h, w))).astype(bool))
            else:
                arr = np.asarray(m.to_dense() if hasattr(m, "to_dense") else m)
                out.append(lambda arr=arr: arr.astype(bool))
        return out

    @staticmethod
    def _box_array(boxes):
        if boxes is None:
            return None
        if hasattr(boxes, "tensor"):
            boxes = boxes.tensor
        if hasattr(boxes, "detach"):
            boxes = boxes.detach().cpu().numpy()
        return np.asarray(boxes, dtype=np.float64).reshape(-1, 4)

    # ---- drawing primitives on self.output.img (uint8, output scale) ----
    def _to_out(self, mask):
        if mask.shape == self._out_hw:
            return mask
        ys = np.minimum((np.arange(self._out_hw[0]) / self.scale).astype(np.int64), mask.shape[0] - 1)
        xs = np.minimum((np.arange(self._out_hw[1]) / self.scale).astype(np.int64), mask.shape[1] - 1)
        return mask[ys][:, xs]

    def draw_binary_mask(self, mask, color, alpha=0.5, edge=True):
        m = self._to_out(np.asarray(mask, dtype=bool))
        if not m.any():
            return self.output
        img = self.output.img
        col = np.asarray(color, dtype=np.float64)[:3] * 255.0
        img[m] = (img[m].astype(np.float64) * (1.0 - alpha) + col * alpha + 0.5).astype(np.uint8)
        if edge:
            inner = m.copy()
            inner[1:, :] &= m[:-1, :]; inner[:-1, :] &= m[1:, :]; inner[:, 1:] &= m[:, :-1]; inner[:, :-1] &= m[:, 1:]
            inner[0, :] = inner[-1, :] = False; inner[:, 0] = inner[:, -1] = False
            img[m & ~inner] = np.clip(col * 0.7, 0, 255).astype(np.uint8)
        return self.output

    def draw_box(self, box, color, line_width=None):
        img = self.output.img
        H, W = img.shape[:2]
        lw = int(line_width if line_width is not None else max(1, round(max(H, W) / 600)))
        x0, y0, x1, y1 = [v * self.scale for v in box]
        x0, x1 = int(np.clip(round(x0), 0, W - 1)), int(np.clip(round(x1), 0, W - 1))
        y0, y1 = int(np.clip(round(y0), 0, H - 1)), int(np.clip(round(y1), 0, H - 1))
        col = np.clip(np.asarray(color, dtype=np.float64)[:3] * 255.0, 0, 255).astype(np.uint8)
        img[y0:min(y0 + lw, H), x0:x1 + 1] = col
        img[max(y1 - lw + 1, 0):y1 + 1, x0:x1 + 1] = col
        img[y0:y1 + 1, x0:min(x0 + lw, W)] = col
        img[y0:y1 + 1, max(x1 - lw + 1, 0):x1 + 1] = col
        return self.output

    def draw_text(self, text, position, color=(1.0, 1.0, 1.0)):
        if not text:
            return self.output
        from PIL import Image, ImageDraw
        pil = Image.fromarray(self.output.img)
        d = ImageDraw.Draw(pil)
        x, y = position[0] * self.scale, position[1] * self.scale
        l, t, r, b = d.textbbox((x, y), text)
        d.rectangle((l - 1, t - 1, r + 1, b + 1), fill=(0, 0, 0))
        d.text((x, y), text, fill=tuple(int(255 * c) for c in color[:3]))
        self.output.img[:] = np.asarray(pil)
        return self.output

    # ---- the three entry points ----
    def overlay_instances(self, *, boxes=None, labels=None, masks=None, keypoints=None, assigned_colors=None, alpha=0.5):
        boxes = self._box_array(boxes)
        masks = self._mask_list(masks)
        n = len(boxes) if boxes is not None else (len(masks) if masks is not None else (len(labels) if labels is not None else 0))
        if labels is not None:
            assert len(labels) == n, (len(labels), n)
        if masks is not None:
            assert len(masks) == n, (len(masks), n)
        if n == 0:
            return self.output
        colors = _palette(n) if assigned_colors is None else np.asarray(assigned_colors, dtype=np.float64).reshape(n, -1)[:, :3]
        # large instances first so that small ones stay visible (area of the box when there is one, else draw order)
        order = np.argsort(-np.prod(boxes[:, 2:] - boxes[:, :2], axis=1)) if boxes is not None else np.arange(n)
        for i in order:
            if masks is not None:
                self.draw_binary_mask(masks[i](), colors[i], alpha=alpha)
            if boxes is not None:
                self.draw_box(boxes[i], colors[i])
        if labels is not None:
            for i in order:
                if labels[i]:
                    if boxes is not None:
                        pos = (boxes[i][0], boxes[i][1])
                    else:
                        ys, xs = np.nonzero(masks[i]())
                        pos = (float(np.median(xs)), float(np.median(ys))) if len(xs) else (0.0, 0.0)
                    self.draw_text(str(labels[i]), pos)
        return self.output

    def _class_names(self, classes, scores=None):
        names = _meta_get(self.metadata, "thing_classes", None)
        out = []
        for k, c in enumerate(classes):
            s = names[int(c)] if names is not None and 0 <= int(c) < len(names) else str(int(c))
            if scores is not None:
                s = f"{s} {100.0 * float(scores[k]):.0f}%"
            out.append(s)
        return out

    def draw_instance_predictions(self, predictions):
        boxes = predictions.pred_boxes if predictions.has("pred_boxes") else None
        scores = predictions.scores if predictions.has("scores") else None
        classes = predictions.pred_classes if predictions.has("pred_classes") else None
        if hasattr(scores, "detach"):
            scores = scores.detach().cpu().numpy()
        if hasattr(classes, "detach"):
            classes = classes.detach().cpu().numpy()
        labels = self._class_names(classes, scores) if classes is not None else None
        masks = predictions.pred_masks if predictions.has("pred_masks") else None
        return self.overlay_instances(boxes=boxes, masks=masks, labels=labels)

    def draw_dataset_dict(self, dic):
        from ..structures import BoxMode
        annos = dic.get("annotations", None)
        if not annos:
            return self.output
        boxes = [BoxMode.convert(a["bbox"], a["bbox_mode"], BoxMode.XYXY_ABS) if len(a["bbox"]) == 4 else a["bbox"] for a in annos]
        masks = [a["segmentation"] for a in annos] if all("segmentation" in a for a in annos) else None
        labels = self._class_names([a["category_id"] for a in annos])
        return self.overlay_instances(boxes=np.asarray(boxes, dtype=np.float64), masks=masks, labels=labels)
