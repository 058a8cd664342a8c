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


def _instance(box, segmentation):
    """One annotation entry of a dataset dict (single-class: category 0, ampis/data_utils.py:428,474,520)."""
    from .structures import BoxMode
    return {"bbox": box, "bbox_mode": BoxMode.XYXY_ABS, "segmentation": segmentation, "category_id": 0}


def _ddict(image_id, file_name, annotation_file, hw, mask_format, instances, dataset_class, **extra):
    """The dataset-dict schema every label format ends in (keys as the reference writes them: ampis/data_utils.py:390-532)."""
    d = {"file_name": file_name, "annotation_file": annotation_file, "height": int(hw[0]), "width": int(hw[1]), "mask_format": mask_format,
         "image_id": image_id, "dataset_class": dataset_class}
    d.update(extra)
    d["annotations"] = instances
    d["num_instances"] = len(instances)
    return d


def _label_image_records(fmt, im_root, ann_root, pattern):
    """'binary' / 'label': one annotation image (or .npy) per micrograph; 'binary' is split into its 8-connected components (the
    default of skimage.measure.label), 'label' already carries one id per instance.  Yields (image path, annotation path, hw, masks)."""
    from scipy import ndimage
    for img_path in im_root.glob(pattern):            # directory order, like the reference (ampis/data_utils.py:394-395): same image_id per file
        found = list(ann_root.glob("*{}*".format(img_path.stem)))
        assert len(found) == 1, f"There must be exactly 1 annotation file for, {img_path.name}, but {len(found)} were found"
        ann = np.load(str(found[0])) if found[0].suffix == ".npy" else _imread(found[0])
        hw = ann.shape[:2]
        if fmt == "binary":
            fg = (ann if ann.ndim == 2 else ann[..., 0]).astype(bool)
            ann = ndimage.label(fg, structure=np.ones((3, 3), int))[0]
        ids = np.unique(ann)
        yield img_path, found[0], hw, [ann == u for u in ids[ids != 0]] if ids.size and ids[0] == 0 else [ann == u for u in ids]


def _via2_records(json_path):
    """VIA 2.x project file: per image the declared size (file attribute 'Size (width, height)', else read from the image) and one
    polygon per region, vertices moved to pixel centres (+0.5).  Yields (image path, hw, HFW attribute, [(box, polygon)])."""
    import json
    from pathlib import Path
    with open(json_path, "rb") as f:
        project = json.load(f)
    img_dir = Path(json_path.parent, project["_via_settings"]["core"]["default_filepath"])
    for entry in project["_via_img_metadata"].values():
        img_path = Path(img_dir, entry["filename"])
        attrs = entry["file_attributes"]
        declared = attrs.get("Size (width, height)", None)
        if declared:
            w, h = (int(v) for v in declared.split(", "))
        else:
            h, w = _imread(img_path).shape[:2]
        regions = []
        for region in entry["regions"]:
            xs, ys = region["shape_attributes"]["all_points_x"], region["shape_attributes"]["all_points_y"]
            outline = np.stack([np.asarray(xs, float) + 0.5, np.asarray(ys, float) + 0.5], axis=1).reshape(-1).tolist()
            regions.append((np.asarray((np.min(xs), np.min(ys), np.max(xs), np.max(ys))), [outline]))
        yield img_path, (h, w), attrs.get("HFW", None), regions


def _rle_records(json_path):
    """JSON list of {'file_name', 'segmentations': [COCO RLE]} (counts as str or bytes).  Yields (image path, hw, [rle])."""
    import json
    from pathlib import Path
    with open(json_path, "r") as f:
        items = json.load(f)
    for item in items:
        rles = [{"size": seg["size"], "counts": seg["counts"].encode("utf-8") if isinstance(seg["counts"], str) else seg["counts"]}
                for seg in item["segmentations"]]
        yield Path(json_path.parent, Path(item["file_name"])), tuple(rles[0]["size"]), rles


def get_ddicts(label_fmt, im_root, ann_root=None, pattern="*", dataset_class=None):
    """Images + single-class instance annotations -> detectron2 dataset dicts. label_fmt: 'binary' | 'label' (annotation images /
    .npy next to the images), 'via2' (VIA 2 JSON; im_root is the JSON path), 'rle' (JSON list of {'file_name','segmentations'})."""
    from pathlib import Path
    im_root = Path(im_root)
    ann_root = Path(ann_root) if ann_root else None
    fmt = label_fmt.lower()

    def rel(path):      # paths relative to the working directory where possible, as the reference stores them
        try:
            return str(Path(path).relative_to(Path()))
        except ValueError:
            return str(path)

    out = []
    if fmt in ("binary", "label"):
        for img, ann_path, hw, masks in _label_image_records(fmt, im_root, ann_root, pattern):
            inst = [_instance(extract_boxes(m)[0], RLE.encode(np.asfortranarray(m))) for m in masks]
            out.append(_ddict(len(out), rel(img), rel(ann_path), hw, "bitmask", inst, dataset_class))
    elif fmt == "via2":
        for img, hw, hfw, regions in _via2_records(im_root):
            inst = [_instance(box, poly) for box, poly in regions]
            out.append(_ddict(len(out), rel(img), im_root.name, hw, "polygon", inst, dataset_class, HFW=hfw))
    elif fmt == "rle":
        for img, hw, rles in _rle_records(im_root):
            inst = [_instance(extract_boxes(RLE.decode(m))[0], m) for m in rles]
            out.append(_ddict(len(out), rel(img), str(im_root), hw, "bitmask", inst, dataset_class))
    else:
        raise ValueError("label_fmt must be 'binary','label', or 'via2'")
    return out
