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


# ---- every import / from line of the six reference modules, executed on the façade ------------------------------------
# The lines are facts about the API surface, listed by module; relative imports of the package itself are exercised by the second test.
PREAMBLE = textwrap.dedent('''
    import os, sys, types
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, ROOT)
    import ampis_amd
    ampis_amd.install_as_detectron2()
    # stand-ins for the two image libraries this image lacks (skimage, cv2): importable, nothing more
    def _stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name); m.__dict__.update(attrs); m.__path__ = []
        sys.modules[name] = m
        return m
    import importlib.util
    if importlib.util.find_spec("skimage") is None:
        sk = _stub("skimage")
        sk.io = _stub("skimage.io"); sk.measure = _stub("skimage.measure"); sk.draw = _stub("skimage.draw", polygon2mask=lambda shape, poly: None)
    if importlib.util.find_spec("cv2") is None:
        def _imread(p, *a):
            import numpy as np
            from PIL import Image
            return np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].copy()
        _stub("cv2", imread=_imread)
''')

IMPORT_LINES = textwrap.dedent('''
    # ampis/data_utils.py:11-29
    import datetime, json, pickle, logging, time
    import numpy as np
    from pathlib import Path
    import pycocotools.mask as RLE
    import skimage
    import skimage.io
    import skimage.measure
    import torch
    from detectron2.data import DatasetMapper, build_detection_test_loader
    from detectron2.engine.hooks import HookBase
    from detectron2.engine.defaults import DefaultTrainer
    import detectron2.utils.comm as comm
    from detectron2.utils.logger import log_every_n_seconds
    from detectron2.structures import BoxMode
    # ampis/structures.py:8-19
    import copy
    import pandas as pd
    from skimage.draw import polygon2mask
    from typing import List, Union
    from detectron2.structures import Boxes, BitMasks, PolygonMasks, Instances
    # ampis/analyze.py:9-14
    import pycocotools.mask as rle
    from detectron2.structures import Instances
    # ampis/visualize.py:7-14
    import colorsys
    import cv2
    import matplotlib.pyplot as plt
    from detectron2.data import MetadataCatalog
    from detectron2.utils.visualizer import Visualizer
    # ampis/applications/powder.py:14-22
    import skimage.io
    from detectron2.structures import Instances
    import ampis_amd.utils.visualizer, ampis_amd.engine.defaults
    assert Visualizer is ampis_amd.utils.visualizer.Visualizer and DefaultTrainer is ampis_amd.engine.defaults.DefaultTrainer
    assert callable(log_every_n_seconds) and BoxMode.XYXY_ABS == 0
    print("IMPORT LINES OK")
''')


def test_every_import_line_of_the_six_reference_modules_resolves_on_the_facade(tmp_path):
    script = f"ROOT = {ROOT!r}\n" + PREAMBLE + IMPORT_LINES
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "IMPORT LINES OK" in r.stdout, r.stderr[-3000:]


REFERENCE = "/root/reference"

REFERENCE_RUN = textwrap.dedent('''
    import pickle
    import numpy as np
    sys.path.insert(0, REFERENCE)
    for _alias, _t in (("int", int), ("float", float), ("bool", bool)):    # the reference targets numpy < 1.24 (env.yml): the aliases numpy 2 removed
        if not hasattr(np, _alias):
            setattr(np, _alias, _t)
    import ampis                                                            # the reference package itself, unmodified
    from ampis import analyze, data_utils, structures, visualize, applications
    import ampis_amd.engine, ampis_amd.analyze as product, ampis_amd.data_utils as product_du, ampis_amd.rle as prle
    assert data_utils.DefaultTrainer is ampis_amd.engine.DefaultTrainer and issubclass(data_utils.AmpisTrainer, ampis_amd.engine.DefaultTrainer)
    assert issubclass(data_utils.LossEvalHook, ampis_amd.engine.HookBase)

    # (1) the reference's own known-answer test (analyze.py:702-728) runs on the C-ABI codec
    enc = lambda a: prle.encode(np.asfortranarray(np.array(a, np.uint8)))
    m1 = enc([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]); m2 = enc([[0, 0, 1, 1], [0, 0, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]])
    m3 = enc([[0, 0, 0, 0], [0, 0, 0, 0], [1, 1, 0, 0], [1, 1, 0, 0]]); m4 = enc([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]])
    gt, pred = [m1, m2, m3, m4], [m3, m2, m4]
    assert np.all(analyze._piecewise_iou(gt, pred) == np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]]))
    match = analyze._piecewise_rle_match(gt, pred)
    assert np.all(match["tp"] == np.array([[1, 1], [2, 0], [3, 2]])) and np.all(match["fn"] == np.array([0])) and len(match["fp"]) == 0

    # (2) the product's evaluation harness against the REFERENCE's, on random masks with exact ties and > 80 masks a side
    for trial in range(6):
        r = np.random.default_rng(trial)
        H, W = 24, 36
        cells = [(y, x) for y in range(0, H, 6) for x in range(0, W, 6)]
        def blob(ks):
            m = np.zeros((H, W), np.uint8)
            for k in ks:
                m[cells[k][0]:cells[k][0] + 6, cells[k][1]:cells[k][1] + 6] = 1
            return m
        ng, npred = (int(r.integers(1, 30)), int(r.integers(1, 30))) if trial < 4 else (100, 170)
        g = [enc(blob(r.choice(24, size=int(r.integers(1, 4)), replace=False))) for _ in range(ng)]
        p = [enc(blob(r.choice(24, size=int(r.integers(1, 4)), replace=False))) for _ in range(npred)]
        p += p[:2]
        assert np.array_equal(product.iou_matrix(g, p), analyze._piecewise_iou(g, p))
        for thr in (0.0, 1 / 3, 0.5, 0.999):
            a, b = product.rle_instance_matcher(g, p, iou_thresh=thr), analyze._piecewise_rle_match(g, p, iou_thresh=thr)
            assert all(np.array_equal(np.asarray(a[k]).reshape(-1), np.asarray(b[k]).reshape(-1)) for k in ("tp", "fn", "fp", "iou")), (trial, thr)
        gi = structures.InstanceSet(); gi.instances = structures.Instances((H, W), masks=structures.RLEMasks(g))
        pi = structures.InstanceSet(); pi.instances = structures.Instances((H, W), masks=structures.RLEMasks(p))
        want = analyze.det_seg_scores(gi, pi, size=(H, W))
        got = product.det_seg_scores(g, p, size=(H, W))
        for k in want:
            assert np.allclose(np.asarray(got[k], float).reshape(-1), np.asarray(want[k], float).reshape(-1), equal_nan=True), k

    # (3) InstanceSet.read_from_model_out (structures.py:312-371) and the filters on the reference's own pickle cut
    with open(FIXTURE, "rb") as f:
        outs = pickle.load(f)
    iset = structures.InstanceSet().read_from_model_out(outs, inplace=False)
    assert len(iset.instances) == 12 and iset.dataset_class == "Train" and isinstance(iset.instances.masks, structures.RLEMasks)
    assert iset.instances.colors.shape == (12, 3)
    areas = structures.mask_areas(iset.instances.masks)
    assert np.array_equal(np.asarray(areas), prle.area(iset.instances.masks.rle))
    small = iset.filter_mask_size(min_thresh=float(np.sort(areas)[3]), max_thresh=float(np.sort(areas)[-1]) + 1)
    assert 0 < len(small) < 12 and isinstance(small.masks, structures.RLEMasks)      # returns an Instances (structures.py:374-438)
    trimmed = iset.copy()
    trimmed.remove_edge_instances()                                                   # in place (structures.py:445-472)
    assert len(trimmed.instances) <= 12

    # (4) compress_pred / format_outputs of the reference on the façade's Instances == the product's mirror
    from detectron2.structures import Instances, Boxes
    import torch
    dense = np.stack([prle.decode(m).astype(bool) for m in outs["pred"]["instances"].pred_masks[:3]])
    def fresh():
        return Instances((1024, 1536), pred_boxes=Boxes(torch.as_tensor(outs["pred"]["instances"].pred_boxes[:3])), scores=torch.as_tensor(outs["pred"]["instances"].scores[:3]),
                         pred_classes=torch.as_tensor(outs["pred"]["instances"].pred_classes[:3]), pred_masks=torch.as_tensor(dense))
    a = data_utils.format_outputs("x.png", "d_Train", {"instances": fresh()})
    b = product_du.format_outputs("x.png", "d_Train", {"instances": fresh()})
    ia, ib = a["pred"]["instances"], b["pred"]["instances"]
    assert [m["counts"] for m in ia.pred_masks] == [m["counts"] for m in ib.pred_masks] == [m["counts"] for m in outs["pred"]["instances"].pred_masks[:3]]
    assert np.array_equal(ia.pred_boxes, ib.pred_boxes) and np.array_equal(ia.scores, ib.scores) and ia.pred_classes.dtype == ib.pred_classes.dtype

    # (5) dataset ingestion: the reference's get_ddicts('via2') against the product's on the reference's four VIA projects
    #     (sizes are declared in the files' attributes, so no image is read)
    via_dir = os.path.join(REFERENCE, "examples", "powder", "data", "via_2.0.8")
    os.chdir(via_dir)
    n_inst = 0
    for js in sorted(f for f in os.listdir(".") if f.endswith(".json")):
        ref_dd = data_utils.get_ddicts("via2", js, dataset_class="Train")
        mine = product_du.get_ddicts("via2", js, dataset_class="Train")
        assert len(mine) == len(ref_dd) > 0
        for x, y in zip(mine, ref_dd):
            assert set(x) == set(y), (set(x) ^ set(y))
            for k in y:
                if k != "annotations":
                    assert x[k] == y[k], (js, k, x[k], y[k])
            for ax_, ay in zip(x["annotations"], y["annotations"]):
                assert set(ax_) == set(ay)
                assert np.array_equal(np.asarray(ax_["bbox"]), np.asarray(ay["bbox"])) and ax_["bbox_mode"] == ay["bbox_mode"]
                assert ax_["category_id"] == ay["category_id"] and len(ax_["segmentation"]) == len(ay["segmentation"]) == 1
                assert np.array_equal(np.asarray(ax_["segmentation"][0], float), np.asarray(ay["segmentation"][0], float))
                n_inst += 1
    assert n_inst > 2000
    print("VIA PARITY", n_inst)

    # (6) the visualisation calls of notebook cells 16 / 28 on the façade's Visualizer
    img = np.full((1024, 1536, 3), 90, np.uint8)
    out_img = visualize.display_iset(img, iset, metadata={"thing_classes": ["particle"]}, get_img=True, apply_correction=True)
    assert out_img.shape == img.shape and out_img.dtype == np.uint8 and (out_img != img).any()
    keep = np.logical_or.reduce([prle.decode(m).astype(bool) for m in iset.instances.masks.rle])
    assert np.array_equal(out_img[~keep], img[~keep])                       # apply_correction: pixels outside every mask are the image's
    print("REFERENCE RUN OK")
''')


def test_the_reference_package_itself_imports_and_runs_on_the_facade(tmp_path):
    """Here (where /root/reference exists; skipped elsewhere): `import ampis` -- the reference, unmodified -- with the façade installed,
    then its own KAT, its evaluation harness against the product's, read_from_model_out + filters on its own pickle cut,
    compress_pred / format_outputs against the product's mirror, and display_iset through the façade's Visualizer."""
    import pytest
    if not os.path.isdir(os.path.join(REFERENCE, "ampis")):
        pytest.skip("the reference tree is not on this machine")
    fixture = os.path.join(ROOT, "tests", "golden", "particle_results_subset.pickle")
    script = f"ROOT = {ROOT!r}\nFIXTURE = {fixture!r}\nREFERENCE = {REFERENCE!r}\n" + PREAMBLE + REFERENCE_RUN
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and "REFERENCE RUN OK" in r.stdout, (r.stdout[-1500:] + r.stderr[-4000:])


NOTEBOOK_REPLAY = textwrap.dedent('''
    import json, traceback
    import numpy as np
    for _alias, _t in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, _alias):
            setattr(np, _alias, _t)
    # stand-ins that do a little more than exist, where the notebook only needs trivia: reading a PNG, grey -> RGB; seaborn is absent here
    from PIL import Image
    sys.modules["skimage.io"].imread = lambda p, as_gray=False: np.asarray(Image.open(str(p)).convert("L"))
    sys.modules["skimage"].color = types.ModuleType("skimage.color")
    sys.modules["skimage"].color.gray2rgb = lambda a: np.stack([a] * 3, -1) if a.ndim == 2 else a
    sys.modules["skimage.color"] = sys.modules["skimage"].color
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    sys.path.insert(0, REFERENCE)
    os.chdir(WORK)                                     # the notebook addresses the checkout as ./AMPIS
    nb = json.load(open(os.path.join(REFERENCE, "colab", "AMPIS Tutorial.ipynb")))
    ns, ok, failed = {}, [], {}
    for i, c in enumerate(nb["cells"]):
        if c["cell_type"] != "code" or i < FIRST_CELL or i > 68 or i in SKIP_CELLS:
            continue
        src = "".join(l for l in c["source"] if not l.lstrip().startswith(("%", "!")))
        try:
            exec(compile(src, f"cell{i}", "exec"), ns)
            ok.append(i)
        except Exception as e:
            failed[i] = f"{type(e).__name__}: {e}"[:200]
    print("CELLS OK", ok)
    print("CELLS FAILED", json.dumps(failed))
    res = ns["results_pred"]
    print("SATELLITES", ns["psi_pred"][0].matches is not None, len(ns["dss_particles"]), float(np.mean([d["det_precision"] for d in ns["dss_particles"]])))
''')


def test_notebook_analysis_cells_run_on_the_facade(tmp_path):
    """SURVEY §8b's acceptance for the analysis half of the tutorial (NB:c33-c68: ground truth and predictions into InstanceSets, alignment,
    det_seg_scores on all five micrographs, detection-performance overlays through display_iset, powder size distribution, particle /
    satellite matching and its summary): the cells' source is read from the reference's notebook AT TEST TIME (where that tree exists) and
    executed unmodified on the façade.  The cells that need what this image lacks fail for exactly that reason and nothing else: seaborn
    (c48, c52: plots), skimage.measure.regionprops_table (c55) and skimage.draw.polygon2mask (c50: polygon ground truth to dense masks)."""
    import json
    import pytest
    if not os.path.isfile(os.path.join(REFERENCE, "colab", "AMPIS Tutorial.ipynb")):
        pytest.skip("the reference tree is not on this machine")
    os.symlink(REFERENCE, tmp_path / "AMPIS")
    script = f"ROOT = {ROOT!r}\nREFERENCE = {REFERENCE!r}\nWORK = {str(tmp_path)!r}\nFIRST_CELL = 33\nSKIP_CELLS = ()\n" + PREAMBLE + NOTEBOOK_REPLAY
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=1200, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    ok = json.loads([l for l in r.stdout.splitlines() if l.startswith("CELLS OK")][0][len("CELLS OK "):])
    failed = json.loads([l for l in r.stdout.splitlines() if l.startswith("CELLS FAILED")][0][len("CELLS FAILED "):])
    assert set(ok) >= {33, 35, 36, 38, 40, 42, 44, 46, 56, 57, 61, 62, 64, 66, 68}, (ok, failed)
    assert set(map(int, failed)) <= {48, 50, 52, 55}, failed
    for i, why in failed.items():
        assert ("seaborn" in why) or ("regionprops_table" in why) or (int(i) == 50 and "broadcast" in why), (i, why)
    assert "number of satellited particles" in r.stdout and "SATELLITES True 5" in r.stdout


def test_notebook_setup_and_training_cells_run_on_the_facade(tmp_path):
    """The FIRST half of the tutorial (NB:c4-c22), read from the reference's notebook at test time and executed unmodified on the façade in this
    GPU-less container: the imports of c7, the VIA paths of c11, `DatasetCatalog.register` with the `get_ddicts` lambdas and the metadata of
    c13, `display_ddicts` on training and validation dicts (c16, c18), and the whole cfg of c20 (`merge_from_file(model_zoo.get_config_file(..))`,
    `get_checkpoint_url`, OUTPUT_DIR) must pass; c4 may fail only on `torchvision` (absent from this image); c22 -- `DefaultTrainer(cfg)`,
    `resume_or_load`, `train()` -- may stop only where the card is asked for (tests/test_trainer_gpu.py runs that half on the GPU box,
    call for call).  c24-c30 need the checkpoints c22 writes and are not run here."""
    import json
    import pytest
    if not os.path.isfile(os.path.join(REFERENCE, "colab", "AMPIS Tutorial.ipynb")):
        pytest.skip("the reference tree is not on this machine")
    os.symlink(REFERENCE, tmp_path / "AMPIS")
    replay = NOTEBOOK_REPLAY.split('res = ns["results_pred"]')[0].replace("i > 68", "i > 22")
    script = f"ROOT = {ROOT!r}\nREFERENCE = {REFERENCE!r}\nWORK = {str(tmp_path)!r}\nFIRST_CELL = 4\nSKIP_CELLS = ()\n" + PREAMBLE + replay
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=1200, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    ok = json.loads([l for l in r.stdout.splitlines() if l.startswith("CELLS OK")][0][len("CELLS OK "):])
    failed = json.loads([l for l in r.stdout.splitlines() if l.startswith("CELLS FAILED")][0][len("CELLS FAILED "):])
    assert set(ok) >= {7, 11, 13, 16, 18, 20}, (ok, failed)
    assert set(map(int, failed)) <= {4, 22}, failed
    if "4" in failed:
        assert "torchvision" in failed["4"], failed
    if "22" in failed:
        assert any(t in failed["22"] for t in ("no HIP device", "hipGetDeviceCount", "no GPU", "libampis_hip")), failed
    assert "Registered Datasets: ['particle_Train', 'particle_Val']" in r.stdout
