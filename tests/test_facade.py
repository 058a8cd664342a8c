"""CPU tests of the detectron2-surface façade: cfg, catalogs, structures, pickling contract, output mirror."""
import base64
import gzip
import json
import os
import pickle

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "rle_pickles.json.gz")


def test_cfg_notebook_cell20_sequence():
    from ampis_amd import model_zoo
    from ampis_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.INPUT.MASK_FORMAT = "polygon"
    cfg.DATASETS.TRAIN = ("particle_Train",)
    cfg.DATASETS.TEST = ("particle_Train", "particle_Val")
    cfg.SOLVER.IMS_PER_BATCH = 1
    cfg.SOLVER.CHECKPOINT_PERIOD = 400
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
    cfg.TEST.DETECTIONS_PER_IMAGE = 400
    cfg.SOLVER.MAX_ITER = 2000
    cfg.MODEL.WEIGHTs = "unknown keys are accepted like yacs does"   # the tutorial's typo
    cfg.OUTPUT_DIR = "particle_output"
    assert cfg.DATASETS.TEST[0] == "particle_Train" and cfg.SOLVER.CHECKPOINT_PERIOD == 400     # what data_utils.py:158,169 read
    assert cfg.MODEL.RPN.PRE_NMS_TOPK_TEST == 1000 and cfg.MODEL.RPN.POST_NMS_TOPK_TEST == 1000
    assert cfg.SOLVER.BASE_LR == 0.02 and cfg.INPUT.MIN_SIZE_TEST == 800 and cfg.INPUT.MAX_SIZE_TEST == 1333
    assert cfg.MODEL.WEIGHTS.startswith("detectron2://")   # untouched by the typo
    assert "model_final_f10217.pkl" in model_zoo.get_checkpoint_url("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml")
    c2 = cfg.clone()
    c2.MODEL.ROI_HEADS.NUM_CLASSES = 5
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 1


def test_catalogs():
    from ampis_amd.data import DatasetCatalog, MetadataCatalog
    DatasetCatalog.clear()
    calls = []
    DatasetCatalog.register("particle_Train", lambda f="x.json": calls.append(f) or [{"file_name": "a.png"}])
    assert list(DatasetCatalog.data.keys()) == ["particle_Train"]
    assert DatasetCatalog.get("particle_Train")[0]["file_name"] == "a.png"
    DatasetCatalog.get("particle_Train")
    assert calls == ["x.json", "x.json"]          # the lambda runs on every get
    with pytest.raises(AssertionError):
        DatasetCatalog.register("particle_Train", lambda: [])
    with pytest.raises(KeyError):
        DatasetCatalog.get("nope")
    MetadataCatalog.get("particle_Train").set(**{"thing_classes": ["particle"]})
    assert MetadataCatalog.get("particle_Train").thing_classes == ["particle"]
    DatasetCatalog.clear()


def test_instances_semantics_and_pickle_contract():
    import ampis_amd
    from ampis_amd.structures import Boxes, Instances
    inst = Instances((10, 20))
    inst.pred_boxes = Boxes(torch.tensor([[0, 0, 5, 5], [1, 1, 4, 9], [2, 2, 3, 3.0]]))
    inst.scores = torch.tensor([0.9, 0.5, 0.2])
    inst.pred_classes = torch.tensor([0, 0, 1])
    inst.pred_masks = [{"size": [10, 20], "counts": b"x"}] * 3
    assert len(inst) == 3 and inst.image_size == (10, 20) and inst.has("scores") and not inst.has("nope")
    sub = inst[np.array([True, False, True])]
    assert len(sub) == 2 and sub.scores.tolist() == pytest.approx([0.9, 0.2]) and len(sub.pred_masks) == 2
    assert len(inst[[2, 0]]) == 2 and len(inst[1]) == 1
    with pytest.raises(AssertionError):
        inst.bad = [1, 2]
    with pytest.raises(AttributeError):
        inst.missing
    st = pickle.loads(pickle.dumps(inst)).__dict__
    assert set(st) == {"_image_size", "_fields"}              # App. B: the state layout of the reference's pickles


def test_golden_results_roundtrip_through_facade_and_compress_pred():
    """Rebuild one reference result (particle-results.pickle image 0) as predictor output, run the output mirror
    (compress_pred / format_outputs) and check the container contract + byte-identical RLE."""
    from ampis_amd import data_utils
    from ampis_amd.structures import Boxes, Instances, RLEBitMasks
    with gzip.open(GOLD, "rt") as f:
        gold = json.load(f)
    im = gold["files"][0]["images"][0]
    h, w = im["image_size"]
    rles = [{"size": [h, w], "counts": base64.b64decode(c)} for c in im["counts_b64"]]
    inst = Instances((h, w))
    inst.pred_boxes = Boxes(torch.tensor(im["boxes"], dtype=torch.float32).reshape(-1, 4))
    inst.scores = torch.tensor(im["scores"], dtype=torch.float32)
    inst.pred_classes = torch.tensor(im["classes"], dtype=torch.int64)
    inst.pred_masks = RLEBitMasks(rles, (h, w))
    # the reference's own expression (data_utils.py:275) on the lazy masks gives the same bytes
    from ampis_amd import rle as RLE
    x = next(iter(inst.pred_masks))
    assert RLE.encode(np.asfortranarray(x.to("cpu").numpy()))["counts"] == rles[0]["counts"]
    out = data_utils.format_outputs(im["file_name"], im["dataset"], {"instances": inst})
    assert set(out) == {"file_name", "dataset", "pred"}
    p = out["pred"]["instances"]
    assert p is inst                                           # mutated in place
    assert isinstance(p.pred_boxes, np.ndarray) and p.pred_boxes.dtype == np.float32 and p.pred_boxes.shape == (len(rles), 4)
    assert p.scores.dtype == np.float32 and p.pred_classes.dtype == np.int64
    assert [m["counts"] for m in p.pred_masks] == [r["counts"] for r in rles]
    assert pickle.loads(pickle.dumps(out))["pred"]["instances"].pred_masks[0]["size"] == [h, w]


def test_checkpoint_roundtrip(tmp_path):
    from ampis_amd import checkpoint, params as P
    p = P.init_params(1, seed=4)
    path = tmp_path / "model_final.pth"
    checkpoint.save_checkpoint(path, p, iteration=1999)
    q = checkpoint.load_checkpoint(path, 1)
    assert set(q) == set(p) and all(np.array_equal(p[k], q[k]) for k in p)
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(path, 2, strict=True)
    # detectron2 semantics by default: the heads of another class count are skipped and keep their initialisation (tests/test_checkpoint.py)
    q2, rep = checkpoint.load_checkpoint(path, 2, with_report=True)
    assert len(rep["shape_mismatch"]) == 6 and q2["roi_heads.box_predictor.cls_score.weight"].shape == (3, 1024)
    with pytest.raises(FileNotFoundError):
        checkpoint.load_checkpoint("https://dl.fbaipublicfiles.com/x.pkl", 1)
    # model-zoo style .pkl: pickled {'model': {name: ndarray}}
    pk = tmp_path / "model_final_f10217.pkl"
    with open(pk, "wb") as f:
        pickle.dump({"model": p, "__author__": "Detectron2 Model Zoo"}, f)
    assert np.array_equal(checkpoint.load_checkpoint(pk, 1)["roi_heads.mask_head.predictor.weight"], p["roi_heads.mask_head.predictor.weight"])


def test_resize_shortest_edge_rule():
    from ampis_amd.engine.defaults import resize_shortest_edge
    img = np.zeros((1024, 1536, 3), np.uint8)
    assert resize_shortest_edge(img, 800, 1333).shape == (800, 1200, 3)
    assert resize_shortest_edge(np.zeros((1024, 1024, 3), np.uint8), 1024, 1024).shape == (1024, 1024, 3)
    assert resize_shortest_edge(np.zeros((483, 645, 3), np.uint8), 800, 1333).shape == (800, 1068, 3)
    assert resize_shortest_edge(np.zeros((400, 2000, 3), np.uint8), 800, 1333).shape == (267, 1333, 3)


def test_trainer_needs_a_gpu_and_says_so():
    """No CPU fallback: without a HIP device the trainer refuses to construct (this container has no GPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ampis_amd import _lib
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog
    from ampis_amd.engine import DefaultTrainer
    DatasetCatalog.clear()
    DatasetCatalog.register("t_Train", lambda: [{"annotations": [{"bbox": [0, 0, 4, 4], "bbox_mode": 0, "segmentation": [[0, 0, 4, 0, 4, 4]], "category_id": 0}],
                                                 "image_bgr": np.zeros((32, 32, 3), np.uint8)}])
    cfg = get_cfg()
    cfg.DATASETS.TRAIN = ("t_Train",)
    cfg.SOLVER.IMS_PER_BATCH = 1
    with pytest.raises(_lib.AmpError):
        DefaultTrainer(cfg)
    DatasetCatalog.clear()


def test_c_abi_exports_every_declared_symbol():
    """Every function include/ampis_hip.h declares is exported by the built library (no compute calls here)."""
    import re
    from ampis_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "ampis_hip.h")).read()
    names = set(re.findall(r"^(?:int|void|size_t|const char\*|void\*)\s+\*?\s*(amp_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    L = _lib.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert len(names) >= 40


def test_bench_reads_the_pmc_summary_of_its_own_leg():
    """bench.py quotes HBM bytes / MFMA utilisation from the committed rocprofv3 --pmc summaries: the inference line from
    pmc_summary_vN.json, the training object from pmc_summary_train_vN.json, newest N by name (file times are equal after a checkout)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    t, src, key = b.pmc_traffic(["conv_split_kernel<128x256>"])
    assert t and "_train_" not in src and key == "conv_split_kernel<128x256>"
    t2, src2, _ = b.pmc_traffic(["wgrad_split_kernel"], leg="train")
    assert t2 and "_train_" in src2
    rp = b.rocprof_reported("infer")
    assert rp and rp["source"] == src and any(k.startswith("roi_align") for k in rp["kernels"])
    assert all(v["hbm_GBps"] is not None for v in rp["kernels"].values())


def test_visualizer_members_ampis_calls():
    """detectron2.utils.visualizer.Visualizer as ampis/visualize.py:154-164,291,326-328 uses it: RLE dicts, polygons and bool arrays in,
    uint8 image out; mask pixels are blended, everything else keeps the input; draw_dataset_dict / draw_instance_predictions."""
    from ampis_amd import rle
    from ampis_amd.structures import Boxes, BoxMode, Instances
    from ampis_amd.utils.visualizer import Visualizer
    img = np.full((60, 80, 3), 100, np.uint8)
    m0 = np.zeros((60, 80), bool); m0[10:30, 10:40] = True
    m1 = np.zeros((60, 80), bool); m1[35:55, 50:75] = True
    boxes = np.array([[10, 10, 39, 29], [50, 35, 74, 54]], np.float32)
    out = Visualizer(img, {"thing_classes": ["a"]}, scale=1).overlay_instances(boxes=boxes, masks=[rle.encode(m0), rle.encode(m1)],
                                                                               labels=["", ""], assigned_colors=np.array([[1, 0, 0], [0, 1, 0]])).get_image()
    assert out.shape == img.shape and out.dtype == np.uint8
    assert np.array_equal(out[~(m0 | m1)], img[~(m0 | m1)])                    # boxes coincide with the masks' extents here
    assert out[20, 20, 0] > 150 and out[20, 20, 1] < 100 and out[45, 60, 1] > 150   # red fill, green fill
    # same picture from bool arrays and from polygons
    out2 = Visualizer(img, None).overlay_instances(masks=np.stack([m0, m1]), assigned_colors=np.array([[1, 0, 0], [0, 1, 0]])).get_image()
    assert np.array_equal(out2[20:25, 20:25], out[20:25, 20:25])
    dd = {"annotations": [{"bbox": [10, 10, 30, 20], "bbox_mode": BoxMode.XYWH_ABS, "segmentation": [[10, 10, 40, 10, 40, 30, 10, 30]], "category_id": 0}]}
    out3 = Visualizer(img, {"thing_classes": [""]}).draw_dataset_dict(dd).get_image()
    assert (out3[12:28, 12:38] != 100).any() and np.array_equal(out3[40:, :], img[40:, :])
    inst = Instances((60, 80), pred_boxes=Boxes(boxes), scores=np.array([0.9, 0.8], np.float32), pred_classes=np.array([0, 0]),
                     pred_masks=[rle.encode(m0), rle.encode(m1)])
    out4 = Visualizer(img, {"thing_classes": ["p"]}, scale=2).draw_instance_predictions(inst).get_image()
    assert out4.shape == (120, 160, 3) and (out4 != 100).any()
    assert BoxMode.convert([10, 10, 30, 20], BoxMode.XYWH_ABS, BoxMode.XYXY_ABS) == [10.0, 10.0, 40.0, 30.0]
