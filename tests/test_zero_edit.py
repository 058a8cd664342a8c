"""The zero-edit boundary and the consumer contract of §8a rows a5 / a6.

`ampis_amd.install_as_detectron2()` registers the façade under the module names the reference imports (ampis/data_utils.py:24-29,
ampis/structures.py:12,19, ampis/analyze.py, ampis/visualize.py: detectron2.engine / structures / data / utils.comm, pycocotools.mask),
so AMPIS code and its pickles run unmodified.  Checked in a fresh interpreter (sys.modules is process-wide):

  * every import the reference's modules make resolves, and resolves to the façade;
  * a result pickle whose class path is detectron2.structures.instances.Instances (tests/golden/particle_results_subset.pickle: the
    first 12 detections of the reference's particle-results.pickle, made by tests/golden/make_golden.py) unpickles;
  * what InstanceSet.read_from_model_out does with it (ampis/structures.py:312-371) works on the façade's Instances: a new
    Instances(image_size, masks=RLEMasks(...), boxes=ndarray, class_idx=ndarray, scores=ndarray), a per-instance `colors` field;
  * what the filters do (ampis/structures.py:395-470): bool-ndarray indexing of every field, Instances[bool ndarray] with an
    RLEMasks field, pycocotools-style RLE.area / RLE.merge / RLE.encode on the masks (remove_edge_instances, :441-470).
The RLEMasks class below restates the indexing contract of the reference's class (ampis/structures.py:24-95): it is test scaffolding.
"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import pickle, sys
    import numpy as np
    sys.path.insert(0, ROOT)
    import ampis_amd
    ampis_amd.install_as_detectron2()
    ampis_amd.install_as_detectron2()                       # idempotent

    # ---- the reference's import lines ----
    from detectron2.engine import DefaultTrainer, DefaultPredictor, HookBase          # data_utils.py:24, notebook cell 7
    from detectron2.engine.hooks import HookBase as HB2
    from detectron2.data import DatasetMapper, build_detection_test_loader, DatasetCatalog, MetadataCatalog   # data_utils.py:25-26
    import detectron2.utils.comm as comm                                              # data_utils.py:27
    from detectron2.structures import BoxMode, Instances, Boxes, BitMasks, PolygonMasks   # data_utils.py:28, structures.py:12
    from detectron2.structures.instances import Instances as I2
    from detectron2.config import get_cfg                                             # notebook cell 7
    from detectron2 import model_zoo
    from detectron2.utils.logger import setup_logger
    import pycocotools.mask as RLE                                                    # data_utils.py:29, structures.py:19, analyze.py
    import ampis_amd.engine, ampis_amd.structures, ampis_amd.data, ampis_amd.rle
    assert DefaultTrainer is ampis_amd.engine.DefaultTrainer and Instances is ampis_amd.structures.Instances and I2 is Instances
    assert DatasetCatalog is ampis_amd.data.DatasetCatalog and RLE is ampis_amd.rle and HB2 is HookBase
    assert comm.get_world_size() == 1 and comm.is_main_process()
    comm.synchronize()

    # ---- the pickle: class path detectron2.structures.instances.Instances ----
    with open(FIXTURE, "rb") as f:
        outs = pickle.load(f)
    pred = outs["pred"]["instances"]
    assert type(pred) is Instances and pred.image_size == (1024, 1536) and len(pred) == 12
    assert pred.pred_boxes.dtype == np.float32 and pred.pred_boxes.shape == (12, 4) and pred.pred_classes.dtype == np.int64
    assert np.all(np.diff(pred.scores) <= 0)

    # ---- RLEMasks: the indexing contract of ampis/structures.py:24-95, restated ----
    class RLEMasks:
        def __init__(self, rle):
            self.rle = rle
        def __len__(self):
            return len(self.rle)
        def __getitem__(self, item):
            import torch
            if type(item) == int or type(item) == slice:
                return RLEMasks(self.rle[item])
            if type(item) == torch.BoolTensor or (type(item) == np.ndarray and item.dtype == bool) or (type(item) == list and type(item[0]) == bool):
                assert len(item) == len(self)
                return RLEMasks([m for m, b in zip(self.rle, item) if b])
            return RLEMasks([self.rle[i] for i in item])

    # ---- InstanceSet.read_from_model_out (structures.py:359-366) ----
    inst = Instances(pred.image_size, **{"masks": RLEMasks(pred.pred_masks), "boxes": pred.pred_boxes,
                                         "class_idx": pred.pred_classes, "scores": pred.scores})
    assert len(inst) == 12 and inst.image_size == (1024, 1536)
    inst.colors = np.random.RandomState(0).rand(len(inst), 3)          # visualize.random_colors(len(self.instances), ...)
    assert set(inst._fields) == {"masks", "boxes", "class_idx", "scores", "colors"}
    dataset_class = outs["dataset"].split("_")[-1]
    assert dataset_class == "Train"

    # ---- filter_mask_size (structures.py:395-438): areas from RLE, bool-ndarray selection of every field ----
    areas = RLE.area(inst.masks.rle).astype(float)
    assert areas.shape == (12,) and areas.min() > 0
    inliers = np.logical_and(areas > np.sort(areas)[2], areas < np.sort(areas)[-2])
    n_in = int(inliers.sum())
    assert 0 < n_in < 12
    fields = {k: (inst.masks[inliers] if k == "masks" else v[inliers]) for k, v in inst._fields.items()}
    filt = Instances(inst.image_size, **fields)
    assert len(filt) == n_in and len(filt.masks) == n_in and filt.colors.shape == (n_in, 3)
    assert [m["counts"] for m in filt.masks.rle] == [m["counts"] for m, b in zip(pred.pred_masks, inliers) if b]

    # ---- remove_edge_instances (structures.py:441-470): border mask, merge(intersect), Instances[bool ndarray] ----
    r, c = inst.image_size
    border = np.ones((r, c), dtype=bool); border[1:-1, 1:-1] = 0
    border = RLE.encode(np.asfortranarray(border))
    touching = RLE.area([RLE.merge([border, x], intersect=True) for x in inst.masks.rle]) != 0
    dec = [RLE.decode(x).astype(bool) for x in inst.masks.rle]
    expect = np.array([m[0].any() or m[-1].any() or m[:, 0].any() or m[:, -1].any() for m in dec])
    assert np.array_equal(touching, expect)
    kept = inst[~touching]
    assert type(kept) is Instances and len(kept) == int((~touching).sum()) and isinstance(kept.masks, RLEMasks)
    assert np.array_equal(kept.boxes, inst.boxes[~touching]) and np.array_equal(kept.scores, inst.scores[~touching])
    # int-array / slice / single-int selection of an Instances holding RLEMasks
    assert len(inst[np.array([3, 0, 7])]) == 3 and inst[np.array([3, 0, 7])].masks.rle[0] is inst.masks.rle[3]
    assert len(inst[2:5]) == 3 and len(inst[4]) == 1
    # a round trip through pickle keeps the reference's class path and state layout (App. B)
    blob = pickle.dumps(pred)
    assert b"detectron2.structures.instances" in blob and set(pickle.loads(blob).__dict__) == {"_image_size", "_fields"}
    print("ZERO-EDIT OK", len(kept))
''')


def test_install_as_detectron2_runs_the_reference_consumer_contract_unmodified(tmp_path):
    fixture = os.path.join(ROOT, "tests", "golden", "particle_results_subset.pickle")
    script = f"ROOT = {ROOT!r}\nFIXTURE = {fixture!r}\n" + SCRIPT
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ZERO-EDIT OK" in r.stdout


def test_install_refuses_to_shadow_a_real_detectron2(tmp_path):
    (tmp_path / "detectron2").mkdir()
    (tmp_path / "detectron2" / "__init__.py").write_text("REAL = True\n")
    code = f"import sys; sys.path.insert(0, {str(tmp_path)!r}); sys.path.insert(0, {ROOT!r}); import ampis_amd\ntry:\n    ampis_amd.install_as_detectron2()\nexcept RuntimeError as e:\n    print('REFUSED', e)\n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout, r.stdout + r.stderr
