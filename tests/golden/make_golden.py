"""Generates tests/golden/rle_pickles.json.gz from the data files the reference ships (run in the build container;
/root/reference does not exist on the GPU box).

Source data (fixtures of the reference): the five result pickles
  examples/powder/data/{particle-results,sample_particle_outputs,satellite-results,sample_satellite_outputs}.pickle
  examples/spheroidite/data/sample-spheroidite-results.pickle
Each is the output of ampis.data_utils.format_outputs (data_utils.py:282-310): a list of
{'file_name', 'dataset', 'pred': {'instances': Instances(pred_boxes f32[N,4], scores f32[N], pred_classes i64[N],
pred_masks list[N] of pycocotools RLE dicts)}}.  detectron2 is not installed, so `Instances` is unpickled through a stub
class that only records its state.  Only DATA is extracted (image sizes, boxes, scores, classes, RLE `counts` bytes);
no reference code is copied or executed.
"""
import base64
import gzip
import json
import os
import pickle
import sys
import types

import numpy as np

REF = "/root/reference"
FILES = [
    "examples/powder/data/particle-results.pickle",
    "examples/powder/data/sample_particle_outputs.pickle",
    "examples/powder/data/satellite-results.pickle",
    "examples/powder/data/sample_satellite_outputs.pickle",
    "examples/spheroidite/data/sample-spheroidite-results.pickle",
]


class _StubInstances:
    def __setstate__(self, state):
        self.__dict__.update(state)

    def __getstate__(self):
        return dict(self.__dict__)


def _install_stub():
    for name in ("detectron2", "detectron2.structures", "detectron2.structures.instances"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["detectron2.structures.instances"].Instances = _StubInstances
    # the arrays were pickled by numpy 1.x under numpy.core.*; numpy 2 keeps an alias module
    return None


def main():
    _install_stub()
    out = {"source": "rccohn/AMPIS result pickles (see docstring)", "files": []}
    n_masks = 0
    for rel in FILES:
        with open(os.path.join(REF, rel), "rb") as f:
            data = pickle.load(f)
        images = []
        for item in data:
            inst = item["pred"]["instances"]
            fields = inst._fields
            masks = fields["pred_masks"]
            images.append({
                "file_name": os.path.basename(item["file_name"]),
                "dataset": item["dataset"],
                "image_size": [int(v) for v in inst._image_size],
                "boxes": np.asarray(fields["pred_boxes"], dtype=np.float32).round(4).tolist(),
                "boxes_dtype": str(np.asarray(fields["pred_boxes"]).dtype),
                "scores": np.asarray(fields["scores"], dtype=np.float64).round(6).tolist(),
                "scores_dtype": str(np.asarray(fields["scores"]).dtype),
                "classes": np.asarray(fields["pred_classes"]).tolist(),
                "classes_dtype": str(np.asarray(fields["pred_classes"]).dtype),
                "mask_sizes_equal_image": all(list(m["size"]) == list(inst._image_size) for m in masks),
                "counts_b64": [base64.b64encode(m["counts"]).decode("ascii") for m in masks],
            })
            n_masks += len(masks)
        out["files"].append({"path": rel, "images": images})
    out["n_masks"] = n_masks
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rle_pickles.json.gz")
    with gzip.open(dst, "wt", compresslevel=9) as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes;", n_masks, "masks")


def make_via_subset():
    """tests/golden/via_subset.json: the first two images (all regions) of the reference's VIA particle training annotations plus
    the per-image region counts of its four VIA files -- input / expected output of get_ddicts('via2')."""
    src = os.path.join(REF, "examples/powder/data/via_2.0.8")
    out = {"_via_settings": None, "_via_img_metadata": {}}
    counts = {}
    for fn in ["via_powder_particle_masks_training.json", "via_powder_particle_masks_validation.json",
               "via_powder_satellite_masks_training.json", "via_powder_satellite_masks_validation.json"]:
        j = json.load(open(os.path.join(src, fn)))
        counts[fn] = [len(v["regions"]) for v in j["_via_img_metadata"].values()]
        if fn == "via_powder_particle_masks_training.json":
            out["_via_settings"] = {"core": {"default_filepath": j["_via_settings"]["core"]["default_filepath"]}}
            for k, v in list(j["_via_img_metadata"].items())[:2]:
                out["_via_img_metadata"][k] = v
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "via_subset.json")
    json.dump({"via": out, "region_counts": counts, "source": "rccohn/AMPIS examples/powder/data/via_2.0.8/*.json"}, open(dst, "w"))


def make_result_pickle_subset():
    """tests/golden/particle_results_subset.pickle: the first image of examples/powder/data/particle-results.pickle cut to its first 12
    detections, re-pickled with the ORIGINAL class path (detectron2.structures.instances.Instances, state keys _image_size / _fields):
    the object `InstanceSet.read_from_model_out` receives (ampis/structures.py:359).  Data only."""
    _install_stub()
    _StubInstances.__module__ = "detectron2.structures.instances"
    _StubInstances.__qualname__ = _StubInstances.__name__ = "Instances"
    with open(os.path.join(REF, FILES[0]), "rb") as f:
        item = pickle.load(f)[0]
    inst = item["pred"]["instances"]
    keep = 12
    cut = _StubInstances()
    cut.__dict__.update({"_image_size": tuple(int(v) for v in inst._image_size),
                         "_fields": {"pred_masks": [dict(size=list(m["size"]), counts=bytes(m["counts"])) for m in inst._fields["pred_masks"][:keep]],
                                     "pred_boxes": np.asarray(inst._fields["pred_boxes"])[:keep].copy(),
                                     "scores": np.asarray(inst._fields["scores"])[:keep].copy(),
                                     "pred_classes": np.asarray(inst._fields["pred_classes"])[:keep].copy()}})
    out = {"file_name": item["file_name"], "dataset": item["dataset"], "pred": {"instances": cut}}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "particle_results_subset.pickle")
    with open(dst, "wb") as f:
        pickle.dump(out, f, protocol=4)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
    make_via_subset()
    make_result_pickle_subset()
