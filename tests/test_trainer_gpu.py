"""End-to-end mini training through the DefaultTrainer façade (notebook cells 13, 20, 22, 24; ampis/data_utils.py:135-177):
register a dataset, subclass the trainer the way AmpisTrainer does (validation-loss hook inserted before the writer), train a
few iterations, check the scalars / checkpoints, then load the final checkpoint into DefaultPredictor."""
import glob
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ddicts(n, h, w, seed):
    from ampis_amd import synth
    out = []
    for i in range(n):
        img, gt = synth.micrograph(i, h, w, seed=seed)
        annos = [{"bbox": b.tolist(), "bbox_mode": 0, "segmentation": [p.tolist()], "category_id": 0}
                 for b, p in list(zip(gt["boxes"], gt["polygons"]))[:50]]
        out.append({"file_name": f"synthetic_{i}.png", "image_bgr": img, "height": h, "width": w, "image_id": i, "annotations": annos,
                    "mask_format": "polygonmask", "num_instances": len(annos)})
    return out


def test_train_validate_checkpoint_predict(tmp_path):
    from ampis_amd import model_zoo
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, DatasetMapper, MetadataCatalog, build_detection_test_loader
    from ampis_amd.engine import DefaultPredictor, DefaultTrainer
    from ampis_amd.engine.hooks import HookBase
    from ampis_amd.utils import comm

    DatasetCatalog.clear()
    train, val = _ddicts(4, 192, 256, 50), _ddicts(2, 192, 256, 60)
    DatasetCatalog.register("particle_Train", lambda: train)
    DatasetCatalog.register("particle_Val", lambda: val)
    for d in ("particle_Train", "particle_Val"):
        MetadataCatalog.get(d).set(**{"thing_classes": ["particle"]})
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.INPUT.MASK_FORMAT = "polygon"
    cfg.DATASETS.TRAIN = ("particle_Train",)
    cfg.DATASETS.TEST = ("particle_Val",)
    cfg.SOLVER.IMS_PER_BATCH = 2
    cfg.SOLVER.CHECKPOINT_PERIOD = 4
    cfg.SOLVER.MAX_ITER = 10
    cfg.SOLVER.BASE_LR = 0.001
    cfg.SOLVER.WARMUP_ITERS = 2
    # like the tutorial, start from a checkpoint on disk (here a well-conditioned seeded one: no pretrained weights offline)
    from ampis_amd import checkpoint, params as P
    os.makedirs(tmp_path / "models", exist_ok=True)
    checkpoint.save_checkpoint(tmp_path / "models" / "init.pth", P.init_params(1, seed=4, style="spread"))
    cfg.MODEL.WEIGHTS = str(tmp_path / "models" / "init.pth")
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
    cfg.TEST.DETECTIONS_PER_IMAGE = 50
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (192,), 256
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 192, 256
    cfg.OUTPUT_DIR = str(tmp_path / "particle_output")

    class LossEval(HookBase):          # the call pattern of ampis.data_utils.LossEvalHook (:62-132)
        def __init__(self, period, model, loader):
            self._model, self._period, self._loader = model, period, loader

        def after_step(self):
            nxt = self.trainer.iter + 1
            if nxt == self.trainer.max_iter or (self._period > 0 and nxt % self._period == 0):
                losses, mds = [], []
                for inputs in self._loader:
                    md = {k: float(v) for k, v in self._model(inputs).items()}
                    losses.append(sum(md.values()))
                    mds.append(md)
                self.trainer.storage.put_scalar("validation_loss", np.mean(losses))
                for k in mds[0]:
                    self.trainer.storage.put_scalar("valid_" + k, np.mean([m[k] for m in mds]))
                comm.synchronize()
            self.trainer.storage.put_scalars(timetest=12)

    class AmpisLikeTrainer(DefaultTrainer):
        def build_hooks(self):
            hooks = super().build_hooks()
            hooks.insert(-1, LossEval(self.cfg.SOLVER.CHECKPOINT_PERIOD, self.model,
                                      build_detection_test_loader(self.cfg, self.cfg.DATASETS.TEST[0], DatasetMapper(self.cfg, True))))
            return hooks

    trainer = AmpisLikeTrainer(cfg)
    assert type(trainer._hooks[-1]).__name__ == "PeriodicWriter" and isinstance(trainer._hooks[-2], LossEval)
    trainer.resume_or_load(resume=False)
    trainer.train()

    st = trainer.storage
    tl = [v for v, _ in st.history("total_loss")]
    assert len(tl) == 10 and all(np.isfinite(tl))
    assert set(["loss_cls", "loss_box_reg", "loss_mask", "loss_rpn_cls", "loss_rpn_loc", "lr", "validation_loss", "valid_loss_mask"]) <= set(st.histories())
    assert [i for _, i in st.history("validation_loss")] == [3, 7, 9]
    assert st.history("lr")[0][0] == pytest.approx(0.001 * 0.001) and st.history("lr")[-1][0] == pytest.approx(0.001)
    assert np.mean(tl[-3:]) < np.mean(tl[:3])          # it learns something on 4 images
    ckpts = sorted(glob.glob(os.path.join(cfg.OUTPUT_DIR, "*.pth")))
    assert [os.path.basename(c) for c in ckpts] == ["model_0000003.pth", "model_0000007.pth", "model_final.pth"]
    assert os.path.isfile(os.path.join(cfg.OUTPUT_DIR, "metrics.json"))
    assert json.loads(open(os.path.join(cfg.OUTPUT_DIR, "metrics.json")).readlines()[-1])["iteration"] == 9

    # notebook cell 24: the last checkpoint feeds the predictor
    cfg.MODEL.WEIGHTS = str(ckpts[-1])
    predictor = DefaultPredictor(cfg)
    outs = predictor(val[0]["image_bgr"])
    assert outs["instances"].image_size == (192, 256)
    DatasetCatalog.clear()


def test_trainer_with_mixed_scales(tmp_path):
    """MIN_SIZE_TRAIN with several choices: the scale is drawn per image, batches mix sizes and go through the per-image-size path;
    training still runs, learns and checkpoints."""
    from ampis_amd import checkpoint, model_zoo, params as P
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, MetadataCatalog
    from ampis_amd.engine import DefaultTrainer
    DatasetCatalog.clear()
    train = _ddicts(4, 192, 256, 70)
    DatasetCatalog.register("particle_Train", lambda: train)
    MetadataCatalog.get("particle_Train").set(thing_classes=["particle"])
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.DATASETS.TRAIN, cfg.DATASETS.TEST = ("particle_Train",), ("particle_Train",)
    cfg.SOLVER.IMS_PER_BATCH, cfg.SOLVER.MAX_ITER, cfg.SOLVER.CHECKPOINT_PERIOD = 3, 8, 100
    cfg.SOLVER.BASE_LR, cfg.SOLVER.WARMUP_ITERS = 0.001, 2
    os.makedirs(tmp_path / "models", exist_ok=True)
    checkpoint.save_checkpoint(tmp_path / "models" / "init.pth", P.init_params(1, seed=4, style="spread"))
    cfg.MODEL.WEIGHTS, cfg.MODEL.ROI_HEADS.NUM_CLASSES = str(tmp_path / "models" / "init.pth"), 1
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (128, 160, 192), 256
    cfg.OUTPUT_DIR = str(tmp_path / "out")
    trainer = DefaultTrainer(cfg)
    trainer.resume_or_load(resume=False)
    seen = []
    orig = trainer.model.__call__
    import types
    def spy(self, batch, **kw):
        from ampis_amd.data import mapped_hw          # the size the network sees (device input path: the resize is deferred to the uploader)
        seen.append(sorted({mapped_hw(d) for d in batch}))
        return orig(batch, **kw)
    trainer.model.__class__ = type("Spy", (trainer.model.__class__,), {"__call__": spy})
    trainer.train()
    tl = [v for v, _ in trainer.storage.history("total_loss")]
    assert len(tl) == 8 and all(np.isfinite(tl))
    assert any(len(s) > 1 for s in seen), "at least one batch mixed scales"
    assert os.path.isfile(os.path.join(cfg.OUTPUT_DIR, "model_final.pth"))
    DatasetCatalog.clear()


def test_finetune_one_class_from_an_80_class_checkpoint_with_larger_validation_images_and_resume(tmp_path):
    """The reference's main workflow (notebook cell 20, GETTING_STARTED.md:30): start from the 80-class COCO zoo checkpoint with
    MODEL.ROI_HEADS.NUM_CLASSES = 1.  The mismatched heads are skipped and initialised detectron2-style (DetectionCheckpointer
    semantics), training is stable and learns; validation images LARGER than every training image go through the validation-loss
    hook (the net is sized once from cfg + dataset sizes); resume=True continues with the iteration and the SGD momentum it stopped with."""
    import pickle
    from ampis_amd import model_zoo, params as P
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, DatasetMapper, MetadataCatalog, build_detection_test_loader
    from ampis_amd.engine import DefaultTrainer
    from ampis_amd.engine.hooks import HookBase
    DatasetCatalog.clear()
    train, val = _ddicts(4, 160, 224, 80), _ddicts(2, 256, 320, 90)          # validation frames exceed every training frame
    DatasetCatalog.register("particle_Train", lambda: train)
    DatasetCatalog.register("particle_Val", lambda: val)
    for d in ("particle_Train", "particle_Val"):
        MetadataCatalog.get(d).set(thing_classes=["particle"])
    zoo = tmp_path / "model_final_f10217.pkl"
    coco = P.init_params(80, seed=4, style="spread")
    with open(zoo, "wb") as f:
        pickle.dump({"model": dict(coco), "__author__": "Detectron2 Model Zoo"}, f)
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.DATASETS.TRAIN, cfg.DATASETS.TEST = ("particle_Train",), ("particle_Val",)
    cfg.SOLVER.IMS_PER_BATCH, cfg.SOLVER.MAX_ITER, cfg.SOLVER.CHECKPOINT_PERIOD = 2, 6, 3
    cfg.SOLVER.BASE_LR, cfg.SOLVER.WARMUP_ITERS = 0.002, 2
    cfg.MODEL.WEIGHTS, cfg.MODEL.ROI_HEADS.NUM_CLASSES = str(zoo), 1
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (0,), 1000          # native sizes: 160x224 training, 256x320 validation
    cfg.OUTPUT_DIR = str(tmp_path / "out")

    class LossEval(HookBase):
        def __init__(self, model, loader):
            self._model, self._loader = model, loader
        def after_step(self):
            if (self.trainer.iter + 1) % 3 == 0:
                self.trainer.storage.put_scalar("validation_loss", np.mean([sum(self._model(b).values()) for b in self._loader]))

    class T(DefaultTrainer):
        def build_hooks(self):
            hooks = super().build_hooks()
            hooks.insert(-1, LossEval(self.model, build_detection_test_loader(self.cfg, "particle_Val", DatasetMapper(self.cfg, True))))
            return hooks

    tr = T(cfg)
    assert tr._cap == (256, 320), tr._cap                                     # both datasets, padded to 32
    tr.resume_or_load(resume=False)
    rep = tr.load_report
    assert {n for n, _, _ in rep["shape_mismatch"]} == {f"roi_heads.{h}.{p}" for h in ("box_predictor.cls_score", "box_predictor.bbox_pred", "mask_head.predictor")
                                                         for p in ("weight", "bias")}
    assert np.array_equal(tr.params["backbone.fpn_output3.weight"], coco["backbone.fpn_output3.weight"])
    assert tr.params["roi_heads.box_predictor.cls_score.weight"].shape == (2, 1024)
    tr.train()
    tl = [v for v, _ in tr.storage.history("total_loss")]
    vl = [v for v, _ in tr.storage.history("validation_loss")]
    assert len(tl) == 6 and all(np.isfinite(tl)) and len(vl) == 2 and all(np.isfinite(vl))
    assert np.mean(tl[-2:]) < np.mean(tl[:2]), tl
    mom_end = tr._net.momentum()
    assert np.abs(mom_end).max() > 0
    w_end = tr._net.get_tensor("backbone.fpn_output3.weight")

    # resume: iteration, weights and momentum continue where the run stopped
    cfg.SOLVER.MAX_ITER = 8
    tr2 = T(cfg)
    tr2.resume_or_load(resume=True)
    assert tr2.start_iter == 6
    assert np.array_equal(tr2.params["backbone.fpn_output3.weight"], w_end)
    tr2.train()
    assert [i for _, i in tr2.storage.history("total_loss")] == [6, 7]
    # the first resumed step started from the stored velocity: with zero momentum the weights after it would differ
    assert tr2._momentum is None
    from ampis_amd import checkpoint
    stored = checkpoint.checkpoint_momentum(os.path.join(cfg.OUTPUT_DIR, open(os.path.join(cfg.OUTPUT_DIR, "last_checkpoint")).read().strip()))
    assert "arena" not in stored and "roi_heads.box_head.fc1.weight" in stored and stored["roi_heads.box_head.fc1.weight"].shape == (1024, 12544)
    tr.close(); tr2.close()
    assert tr.ctx is None and tr._uploader is None and tr.data_loader is None
    DatasetCatalog.clear()


def test_momentum_survives_a_regrown_net(tmp_path):
    """A batch beyond the capacity re-creates the net; weights AND the SGD velocity arena are carried over."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    K = 1
    imgs, gts = synth.batch(2, 128, 160, seed=3)
    gts = [dict(boxes=g["boxes"][:20], classes=np.zeros(min(20, len(g["boxes"])), np.int64), polygons=g["polygons"][:20]) for g in gts]
    p = P.init_params(K, seed=2, style="spread")
    a = MaskRCNN(ctx, K, max_batch=2, max_h=128, max_w=160, max_out_hw=160, train=True, max_gt=512, max_poly_doubles=512 * 64)
    a.load_params(p)
    a.forward_losses(imgs, gts, seed=1, backward=True); a.sgd_step(0.01)
    mom, w = a.momentum(), a.state_dict()
    a.close()
    b = MaskRCNN(ctx, K, max_batch=2, max_h=256, max_w=320, max_out_hw=320, train=True, max_gt=512, max_poly_doubles=512 * 64)
    full = dict(p); full.update(w)
    b.load_params(full)
    b.momentum(mom)
    assert np.array_equal(b.momentum(), mom)
    with pytest.raises(_lib.AmpError):
        b.momentum(mom[:-1])
    # the checkpoint form: per-parameter buffers in torch layout, independent of the arena layout
    a2 = MaskRCNN(ctx, K, max_batch=2, max_h=128, max_w=160, max_out_hw=160, train=True, max_gt=512, max_poly_doubles=512 * 64)
    a2.load_params(full)
    md = b.momentum_dict()
    shapes = P.param_shapes(K)
    assert set(md) == set(b.trainable_names()) and all(md[k].shape == tuple(shapes[k]) for k in md)
    assert not any(k.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2")) or ".norm." in k for k in md)
    assert np.abs(md["roi_heads.box_head.fc1.weight"]).max() > 0 and np.abs(md["proposal_generator.rpn_head.anchor_deltas.weight"]).max() > 0
    assert a2.load_momentum_dict(dict(md, **{"backbone.bottom_up.stem.conv1.weight": np.zeros((64, 3, 7, 7), np.float32)})) == ["backbone.bottom_up.stem.conv1.weight"]
    assert np.array_equal(a2.momentum(), mom)                      # every layout transform inverted exactly, padding untouched
    with pytest.raises(_lib.AmpError):
        a2.load_momentum_dict({"roi_heads.box_head.fc2.bias": np.zeros(7, np.float32)})
    a2.close(); b.close(); ctx.close()


NOTEBOOK_GPU_CELLS = r'''
import os, sys, types, json, pickle
sys.path.insert(0, ROOT)
import numpy as np
import ampis_amd
ampis_amd.install_as_detectron2()
if "cv2" not in sys.modules:                      # this image has no OpenCV: cv2.imread is all the notebook needs of it
    def _imread(p, *a):
        from PIL import Image
        return np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1].copy()
    cv2 = types.ModuleType("cv2"); cv2.imread = _imread; sys.modules["cv2"] = cv2
os.chdir(WORK)
# ---- NB:c7 (imports; `from ampis import data_utils` is the product's mirror here: the reference tree is not on the GPU box) ----
import cv2
from pathlib import Path
from detectron2 import model_zoo
from detectron2.config import get_cfg
from detectron2.data import DatasetCatalog, MetadataCatalog
from detectron2.engine import DefaultTrainer, DefaultPredictor
from ampis_amd import data_utils
# ---- NB:c11 ----
EXPERIMENT_NAME = 'particle'
root = Path('AMPIS','examples','powder')
json_path_train = Path(root,'data','via_2.0.8/', f'via_powder_{EXPERIMENT_NAME}_masks_training.json')
json_path_val = Path(root,'data','via_2.0.8/', f'via_powder_{EXPERIMENT_NAME}_masks_validation.json')
assert json_path_train.is_file(), 'training file not found!'
assert json_path_val.is_file(), 'validation file not found!'
# ---- NB:c13 ----
DatasetCatalog.clear()
dataset_train = f'{EXPERIMENT_NAME}_Train'
dataset_valid = f'{EXPERIMENT_NAME}_Val'
DatasetCatalog.register(dataset_train, lambda f = json_path_train: data_utils.get_ddicts(label_fmt='via2', im_root=f, dataset_class='Train'))
DatasetCatalog.register(dataset_valid, lambda f = json_path_val: data_utils.get_ddicts(label_fmt='via2', im_root=f, dataset_class='Validation'))
print(f'Registered Datasets: {list(DatasetCatalog.data.keys())}')
for d in [dataset_train, dataset_valid]:
    MetadataCatalog.get(d).set(**{'thing_classes': [EXPERIMENT_NAME]})
# ---- NB:c20 ----
cfg = get_cfg()
cfg.merge_from_file(model_zoo.get_config_file('COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml'))
cfg.INPUT.MASK_FORMAT = 'polygon'
cfg.DATASETS.TRAIN = (dataset_train,)
cfg.DATASETS.TEST = (dataset_train, dataset_valid)
cfg.SOLVER.IMS_PER_BATCH = 1
cfg.SOLVER.CHECKPOINT_PERIOD = 400
cfg.MODEL.DEVICE='cuda'
cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
cfg.TEST.DETECTIONS_PER_IMAGE = 400 if EXPERIMENT_NAME == 'particle' else 150
cfg.SOLVER.MAX_ITER = 2000
weights_path = Path('AMPIS','models','model_final_f10217.pkl')
if weights_path.is_file():
    print('Using locally stored weights: {}'.format(weights_path))
else:
    weights_path = model_zoo.get_checkpoint_url("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml")
    print('Weights not found, weights will be downloaded from source: {}'.format(weights_path))
cfg.MODEL.WEIGHTs = str(weights_path)
cfg.OUTPUT_DIR = str(Path(f'{EXPERIMENT_NAME}_output'))
os.makedirs(Path(cfg.OUTPUT_DIR), exist_ok=True)
# ---- the test's only edits: a run of 6 iterations instead of 2000, a checkpoint every 3 ----
cfg.SOLVER.MAX_ITER = 6
cfg.SOLVER.CHECKPOINT_PERIOD = 3
# ---- NB:c22 ----
trainer = DefaultTrainer(cfg)
trainer.resume_or_load(resume=False)
trainer.train()
# ---- NB:c24 ----
model_checkpoints = sorted(Path(cfg.OUTPUT_DIR).glob('*.pth'))
cfg.DATASETS.TEST = (dataset_train, dataset_valid)
cfg.MODEL.WEIGHTS = str(model_checkpoints[-1])
predictor = DefaultPredictor(cfg)
# ---- NB:c26 (without the plot) ----
img_path = Path(root, 'data','images_png','Sc1Tile_001-001-000_0-000.png')
img = cv2.imread(str(img_path))
outs = predictor(img)
data_utils.format_outputs(img_path, dataset='test', pred=outs)
# ---- NB:c28 (without the plots) ----
results = []
for ds in cfg.DATASETS.TEST:
    print(f'Dataset: {ds}')
    for dd in DatasetCatalog.get(ds):
        print(f'\tFile: {dd["file_name"]}')
        img = cv2.imread(dd['file_name'])
        outs = predictor(img)
        results.append(data_utils.format_outputs(dd['file_name'], ds, outs))
prediction_save_path = Path(f'{EXPERIMENT_NAME}-results.pickle')
with open(prediction_save_path, 'wb') as f:
    pickle.dump(results, f)
# ---- what the analysis half (NB:c33 on) expects of that file ----
with open(prediction_save_path, 'rb') as f:
    back = pickle.load(f)
assert len(back) == 3 and [r['dataset'] for r in back] == ['particle_Train', 'particle_Train', 'particle_Val']
for r in back:
    inst = r['pred']['instances']
    assert inst.image_size == (1024, 1536) and isinstance(inst.pred_masks, list) and inst.pred_boxes.dtype == np.float32
    assert len(inst.pred_masks) == len(inst.scores) == len(inst.pred_classes) <= 400 and inst.pred_classes.dtype == np.int64
    assert all(m['size'] == [1024, 1536] and isinstance(m['counts'], bytes) for m in inst.pred_masks)
print('CHECKPOINTS', [p.name for p in model_checkpoints], 'ITER', trainer.iter, 'DETECTIONS', [len(r['pred']['instances'].scores) for r in back])
print('NOTEBOOK GPU CELLS OK')
'''


def test_notebook_training_and_prediction_cells_as_written(tmp_path):
    """NB:c7-c28 of the reference's tutorial as a user runs them on a GPU box -- `DefaultTrainer(cfg)` -> `resume_or_load(resume=False)` ->
    `train()` -> `sorted(glob('*.pth'))[-1]` -> `DefaultPredictor(cfg)` -> `predictor(cv2.imread(..))` -> `format_outputs` -> `pickle.dump` --
    call for call (the plots left out; MAX_ITER 6 instead of 2000), under the module names the notebook imports (`install_as_detectron2`), on
    the two-image cut of the reference's own VIA project with images drawn from its polygons.  cfg.MODEL.WEIGHTS stays the model-zoo URL
    (the notebook's `cfg.MODEL.WEIGHTs = ...` sets nothing): no network, so the trainer starts from its seeded initialisation, with a warning.
    tests/test_zero_edit.py runs c4-c20 and c33-c68 from the notebook file itself where the reference tree exists."""
    import subprocess
    import sys
    from PIL import Image, ImageDraw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    via = json.load(open(os.path.join(root, "tests", "golden", "via_subset.json")))["via"]
    data = tmp_path / "AMPIS" / "examples" / "powder" / "data"
    (data / "via_2.0.8").mkdir(parents=True)
    (data / "images_png").mkdir()
    meta = via["_via_img_metadata"]
    keys = list(meta)
    rng = np.random.RandomState(0)
    for k in keys:                                   # a micrograph-like image under every file name: bright particles where the polygons are
        im = Image.fromarray((40 + 10 * rng.rand(1024, 1536)).astype(np.uint8))
        dr = ImageDraw.Draw(im)
        for reg in meta[k]["regions"]:
            sa = reg["shape_attributes"]
            dr.polygon(list(zip(sa["all_points_x"], sa["all_points_y"])), fill=int(150 + 60 * rng.rand()))
        im.save(data / "images_png" / meta[k]["filename"])
    (data / "via_2.0.8" / "via_powder_particle_masks_training.json").write_text(json.dumps(via))
    val = dict(via, _via_img_metadata={keys[1]: meta[keys[1]]})
    (data / "via_2.0.8" / "via_powder_particle_masks_validation.json").write_text(json.dumps(val))
    script = f"ROOT = {root!r}\nWORK = {str(tmp_path)!r}\n" + NOTEBOOK_GPU_CELLS
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0 and "NOTEBOOK GPU CELLS OK" in r.stdout, (r.stdout[-1500:] + r.stderr[-3000:])
    assert "Registered Datasets: ['particle_Train', 'particle_Val']" in r.stdout
    line = [l for l in r.stdout.splitlines() if l.startswith("CHECKPOINTS")][0]
    assert "model_final.pth" in line and "ITER 6" in line, line
    print(line)
