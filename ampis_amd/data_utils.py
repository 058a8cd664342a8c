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


# ----------------------------------------------------------------------------------------------------------------------
# Dataset ingestion: mirror of ampis/data_utils.py:180-252 (extract_boxes) and :313-532 (get_ddicts) --
# same names, arguments, keys and error behaviour; skimage / pycocotools replaced by numpy, scipy.ndimage, PIL and ampis_amd.rle.
# ----------------------------------------------------------------------------------------------------------------------
def extract_boxes(masks, mask_mode="detectron2", box_mode="detectron2"):
    """Boxes of boolean masks: [x1,y1,x2,y2] float (detectron2) or [y1,y2+1,x1,x2+1] int (matterport); empty mask -> zeros."""
    masks = np.asarray(masks)
    if masks.ndim == 2:
        masks = masks[np.newaxis]
    elif mask_mode == "matterport":
        masks = masks.transpose((2, 0, 1))
    dtype = np.float64 if box_mode == "detectron2" else np.int64
    boxes = np.zeros((masks.shape[0], 4), dtype=dtype)
    for i, m in enumerate(masks):
        xs = np.where(np.any(m, axis=0))[0]
        ys = np.where(np.any(m, axis=1))[0]
        x1, x2, y1, y2 = (xs[0], xs[-1], ys[0], ys[-1]) if xs.shape[0] else (0, 0, 0, 0)
        boxes[i] = [x1, y1, x2, y2] if box_mode == "detectron2" else [y1, y2 + 1, x1, x2 + 1]
    return boxes


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(str(path)))


def get_ddicts(label_fmt, im_root, ann_root=None, pattern="*", dataset_class=None):
    """Images + single-class instance annotations -> detectron2 dataset dicts. label_fmt: 'binary' | 'label' (annotation images /
    .npy next to the images), 'via2' (VIA 2 JSON; im_root is the JSON path), 'rle' (JSON list of {'file_name','segmentations'})."""
    import json
    from pathlib import Path

    from .structures import BoxMode
    cwd = Path()
    im_root = Path(im_root)
    ann_root = Path(ann_root) if ann_root else None
    ddicts = []
    fmt = label_fmt.lower()

    def rel(p):
        try:
            return str(Path(p).relative_to(cwd))
        except ValueError:
            return str(p)

    if fmt in ("binary", "label"):
        from scipy import ndimage
        for idx, p in enumerate(sorted(im_root.glob(pattern))):
            found = list(ann_root.glob("*{}*".format(p.stem)))
            n = len(found)
            assert n == 1, f"There must be exactly 1 annotation file for, {p.name}, but {n} were found"
            ann_path = found[0]
            ann = np.load(str(ann_path)) if ann_path.suffix == ".npy" else _imread(ann_path)
            height, width = ann.shape[:2]
            ddict = {"file_name": rel(p), "annotation_file": rel(ann_path), "height": height, "width": width, "mask_format": "bitmask",
                     "image_id": idx, "dataset_class": dataset_class}
            if fmt == "binary":   # skimage.measure.label default = full (8-) connectivity, labels in raster order
                ann = ndimage.label(ann.astype(bool) if ann.ndim == 2 else ann[..., 0].astype(bool), structure=np.ones((3, 3), int))[0]
            unique = np.unique(ann)
            if unique[0] == 0:
                unique = unique[1:]
            annotations = []
            for u in unique:
                mask = ann == u
                annotations.append({"bbox": extract_boxes(mask)[0], "bbox_mode": BoxMode.XYXY_ABS,
                                    "segmentation": RLE.encode(np.asfortranarray(mask)), "category_id": 0})
            ddict["annotations"] = annotations
            ddict["num_instances"] = len(annotations)
            ddicts.append(ddict)
    elif fmt == "via2":
        with open(im_root, "rb") as f:
            j = json.load(f)
        img_dir = Path(im_root.parent, j["_via_settings"]["core"]["default_filepath"])
        for idx, annos in enumerate(j["_via_img_metadata"].values()):
            filename = Path(img_dir, annos["filename"])
            size = annos["file_attributes"].get("Size (width, height)", None)
            if size:
                width, height = tuple(int(x) for x in size.split(", "))
            else:
                height, width = _imread(filename).shape[:2]
            ddict = {"file_name": rel(filename), "annotation_file": im_root.name, "height": height, "width": width,
                     "mask_format": "polygon", "image_id": idx, "HFW": annos["file_attributes"].get("HFW", None),
                     "dataset_class": dataset_class}
            annotations = []
            for obj in annos["regions"]:
                shape = obj["shape_attributes"]
                px, py = shape["all_points_x"], shape["all_points_y"]
                poly = [v for x, y in zip(px, py) for v in (x + 0.5, y + 0.5)]
                annotations.append({"bbox": np.asarray((np.min(px), np.min(py), np.max(px), np.max(py))), "bbox_mode": BoxMode.XYXY_ABS,
                                    "segmentation": [poly], "category_id": 0})
            ddict["annotations"] = annotations
            ddict["num_instances"] = len(annotations)
            ddicts.append(ddict)
    elif fmt == "rle":
        with open(im_root, "r") as f:
            data = json.load(f)
        for idx, p in enumerate(data):
            ann = [{"size": a["size"], "counts": a["counts"].encode("utf-8") if isinstance(a["counts"], str) else a["counts"]}
                   for a in p["segmentations"]]
            height, width = ann[0]["size"]
            ddict = {"file_name": rel(Path(im_root.parent, Path(p["file_name"]))), "annotation_file": str(im_root), "height": height,
                     "width": width, "mask_format": "bitmask", "image_id": idx, "dataset_class": dataset_class}
            annotations = [{"bbox": extract_boxes(RLE.decode(m))[0], "bbox_mode": BoxMode.XYXY_ABS, "segmentation": m, "category_id": 0}
                           for m in ann]
            ddict["annotations"] = annotations
            ddict["num_instances"] = len(annotations)
            ddicts.append(ddict)
    else:
        raise ValueError("label_fmt must be 'binary','label', or 'via2'")
    return ddicts
