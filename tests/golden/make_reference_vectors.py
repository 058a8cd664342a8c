"""Golden vectors made BY THE REFERENCE ITSELF, run in the build container (where /root/reference exists), so that they can travel to machines
where the reference cannot: `ampis.analyze._piecewise_iou`, `_piecewise_rle_match` and `det_seg_scores` (ampis/analyze.py:54-339) evaluated on
seeded random RLE sets with exact IoU ties, duplicate predictions and more than 80 masks a side.  The reference is imported UNMODIFIED on the
façade (ampis_amd.install_as_detectron2: detectron2.* and pycocotools.mask resolve to this package; skimage / cv2, absent here and not called on
this path, are import stand-ins; numpy's removed np.int / np.bool aliases are restored for it).  Inputs (counts strings) and the reference's
outputs are written to tests/golden/reference_vectors.json.gz; tests/test_reference_vectors.py holds the product to them.

    python tests/golden/make_reference_vectors.py        # needs /root/reference and the built library
"""
import base64
import gzip
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402

import ampis_amd  # noqa: E402

ampis_amd.install_as_detectron2()
for name, attrs in (("skimage", {}), ("skimage.io", {}), ("skimage.measure", {}), ("skimage.draw", {"polygon2mask": lambda shape, poly: None}), ("cv2", {})):
    if name not in sys.modules:
        m = types.ModuleType(name); m.__dict__.update(attrs); m.__path__ = []
        sys.modules[name] = m
for alias, t in (("int", int), ("float", float), ("bool", bool)):
    if not hasattr(np, alias):
        setattr(np, alias, t)
sys.path.insert(0, REFERENCE)
from ampis import analyze, structures  # noqa: E402
from ampis_amd import rle  # noqa: E402


def b64(b):
    return base64.b64encode(b).decode("ascii")


def main():
    cases = []
    for trial in range(10):
        r = np.random.default_rng(1000 + trial)
        H, W = 24, 36
        cells = [(y, x) for y in range(0, H, 6) for x in range(0, W, 6)]

        def blob(ks):
            m = np.zeros((H, W), np.uint8)
            for k in ks:
                m[cells[k][0]:cells[k][0] + 6, cells[k][1]:cells[k][1] + 6] = 1
            return m
        ng, npred = (int(r.integers(1, 30)), int(r.integers(1, 30))) if trial < 7 else (int(r.integers(85, 120)), int(r.integers(90, 170)))
        g = [rle.encode(np.asfortranarray(blob(r.choice(24, size=int(r.integers(1, 4)), replace=False)))) for _ in range(ng)]
        p = [rle.encode(np.asfortranarray(blob(r.choice(24, size=int(r.integers(1, 4)), replace=False)))) for _ in range(npred)]
        p += p[:2]                                                       # exact duplicates: the first-maximum rule
        if trial % 3 == 0:                                               # irregular masks too
            g += [rle.encode(np.asfortranarray((r.random((H, W)) > 0.6).astype(np.uint8))) for _ in range(5)]
            p += [rle.encode(np.asfortranarray((r.random((H, W)) > 0.6).astype(np.uint8))) for _ in range(5)]
        iou = analyze._piecewise_iou(g, p)
        entry = {"size": [H, W], "gt": [b64(x["counts"]) for x in g], "pred": [b64(x["counts"]) for x in p],
                 "iou": [[float(v) for v in row] for row in iou], "match": {}, "scores": None}
        for thr in (0.0, 1 / 3, 0.5, 0.999):
            m = analyze._piecewise_rle_match(g, p, iou_thresh=thr)
            entry["match"][repr(thr)] = {k: np.asarray(m[k]).reshape(-1).tolist() for k in ("tp", "fn", "fp", "iou")}
        gi = structures.InstanceSet(); gi.instances = structures.Instances((H, W), masks=structures.RLEMasks(g))
        pi = structures.InstanceSet(); pi.instances = structures.Instances((H, W), masks=structures.RLEMasks(p))
        sc = analyze.det_seg_scores(gi, pi, size=(H, W))
        entry["scores"] = {k: (np.asarray(v, float).reshape(-1).tolist()) for k, v in sc.items()}
        cases.append(entry)
    # ---- dataset ingestion: ampis.data_utils.get_ddicts('via2') (ampis/data_utils.py:436-477) on the two-image cut of the reference's own VIA
    # project that tests/golden/via_subset.json holds (sizes are declared in the file attributes: no image is read)
    import tempfile
    from ampis import data_utils
    sub = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "via_subset.json")))["via"]
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "via_2.0.8"))
        jp = os.path.join(td, "via_2.0.8", "subset.json")
        json.dump(sub, open(jp, "w"))
        cwd = os.getcwd(); os.chdir(os.path.join(td, "via_2.0.8"))
        try:
            dd = data_utils.get_ddicts("via2", "subset.json", dataset_class="Train")
        finally:
            os.chdir(cwd)
    via = [{k: (v if k != "annotations" else [{"bbox": np.asarray(a["bbox"], float).tolist(), "bbox_mode": int(a["bbox_mode"]), "category_id": int(a["category_id"]),
                                                "segmentation": [np.asarray(sg, float).tolist() for sg in a["segmentation"]]} for a in v])
            for k, v in d.items()} for d in dd]
    # ---- output container: ampis.data_utils.format_outputs / compress_pred (ampis/data_utils.py:255-310) on dense masks of the pickle cut
    import pickle
    import torch
    from detectron2.structures import Boxes, Instances
    outs = pickle.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "particle_results_subset.pickle"), "rb"))
    src = outs["pred"]["instances"]
    dense = np.stack([rle.decode(m).astype(bool) for m in src.pred_masks[:4]])
    inst = Instances((1024, 1536), pred_boxes=Boxes(torch.as_tensor(src.pred_boxes[:4])), scores=torch.as_tensor(src.scores[:4]),
                     pred_classes=torch.as_tensor(src.pred_classes[:4]), pred_masks=torch.as_tensor(dense))
    res = data_utils.format_outputs("x.png", "d_Train", {"instances": inst})
    ri = res["pred"]["instances"]
    container = {"inputs": {"boxes": np.asarray(src.pred_boxes[:4], float).tolist(), "scores": np.asarray(src.scores[:4], float).tolist(),
                            "classes": np.asarray(src.pred_classes[:4]).tolist(), "masks": [b64(m["counts"]) for m in src.pred_masks[:4]]},
                 "file_name": res["file_name"], "dataset": res["dataset"], "image_size": list(ri.image_size),
                 "fields": sorted(ri._fields), "pred_boxes": np.asarray(ri.pred_boxes, float).tolist(), "boxes_dtype": str(np.asarray(ri.pred_boxes).dtype),
                 "scores": np.asarray(ri.scores, float).tolist(), "classes": np.asarray(ri.pred_classes).tolist(), "classes_dtype": str(np.asarray(ri.pred_classes).dtype),
                 "masks": [{"size": list(m["size"]), "counts": b64(m["counts"])} for m in ri.pred_masks]}
    out = {"via2": via, "container": container, "made_by": "tests/golden/make_reference_vectors.py: ampis.analyze of rccohn/AMPIS imported unmodified on the ampis_amd facade",
           "cases": cases}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json.gz")
    with gzip.open(dst, "wt", compresslevel=9) as f:
        json.dump(out, f)
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
