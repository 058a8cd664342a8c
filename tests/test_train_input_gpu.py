"""Train-time input pipeline on the device (round 4; ampis/data_utils.py:171-175 -> detectron2 DatasetMapper: ResizeShortestEdge + RandomFlip,
then ImageList.from_tensors): the loader uploads the decoded image once and amp_resize_flip_u8 resizes (PIL-exact), mirrors and stacks it in
HBM.  The frame the network sees must be byte for byte the one the host path builds, and a training run must not notice the difference."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg(sizes, max_size):
    from ampis_amd import model_zoo
    from ampis_amd.config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = tuple(sizes), max_size
    return cfg


def _dd(i, h, w):
    from ampis_amd import synth
    img, gt = synth.micrograph(i, h, w, seed=5)
    annos = [{"bbox": [float(v) for v in b], "bbox_mode": 0, "segmentation": [[float(v) for v in p]], "category_id": 0}
             for b, p in list(zip(gt["boxes"], gt["polygons"]))[:40]]
    return {"file_name": f"s{i}.png", "image_bgr": img, "height": h, "width": w, "image_id": i, "annotations": annos}


def test_device_built_frames_equal_the_host_stacked_frames_byte_for_byte():
    from ampis_amd.data import DatasetMapper
    from ampis_amd.engine.defaults import TrainModel, _Uploader
    up = _Uploader(0, 2)
    try:
        cases = [
            ([(300, 420), (300, 420), (300, 420)], (200, 232, 264), 333),     # down-scaling, several target sizes in one batch, the max-size clamp
            ([(192, 256), (256, 192), (224, 224)], (256, 288), 512),          # up-scaling, portrait and landscape in one frame
            ([(256, 320), (256, 320)], (256,), 320),                          # no resize at all: straight copies and mirrored copies
        ]
        for shapes, sizes, max_size in cases:
            cfg = _cfg(sizes, max_size)
            dicts = [_dd(i, h, w) for i, (h, w) in enumerate(shapes)]
            m = DatasetMapper(cfg, True, seed=3)
            for rep in range(4):
                plans = [(d,) + tuple(m.draw()) for d in dicts]
                host = [m.apply(*p) for p in plans]
                dev = [m.apply(*p, True) for p in plans]
                imgs, sizes_h, gt_h = TrainModel.collate(host)
                none, sizes_d, gt_d = TrainModel.collate(dev)
                assert none is None and sizes_d == sizes_h
                for a, b in zip(host, dev):
                    assert b["image_bgr"].shape[:2] == (b["height"], b["width"]) and b["device_plan"][:2] == a["image_bgr"].shape[:2]
                    assert np.array_equal(a["gt"]["boxes"], b["gt"]["boxes"]) and np.array_equal(a["gt"]["poly_flat"], b["gt"]["poly_flat"])
                ptr, shp = up.frames(dev)
                assert shp == imgs.shape[:3]
                got = np.empty_like(imgs)
                up.ctx.d2h(got, ptr)
                assert np.array_equal(got, imgs), f"{int((got != imgs).sum())} bytes differ (sizes {sizes}, flips {[p[2] for p in plans]})"
    finally:
        up.close()


def test_trainer_runs_the_same_losses_with_the_device_and_the_host_input_path(tmp_path, monkeypatch):
    """DefaultTrainer, multi-scale + flip, NUM_WORKERS = 2, six iterations: losses bit for bit the same whether the frames are built on the
    device (default) or on the host (AMP_HOST_TRAIN_INPUT=1) -- a training step is bitwise reproducible, so any differing input byte would show."""
    from ampis_amd import checkpoint, params as P
    from ampis_amd.data import DatasetCatalog, MetadataCatalog
    from ampis_amd.engine import DefaultTrainer
    dicts = [_dd(i, 288, 352) for i in range(6)]
    checkpoint.save_checkpoint(str(tmp_path / "init.pth"), P.init_params(1, seed=4, style="spread"))
    runs = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("AMP_HOST_TRAIN_INPUT", "1")
        else:
            monkeypatch.delenv("AMP_HOST_TRAIN_INPUT", raising=False)
        DatasetCatalog.clear()
        DatasetCatalog.register("particle_Train", lambda: dicts)
        MetadataCatalog.get("particle_Train").set(thing_classes=["particle"])
        cfg = _cfg((224, 256, 288), 352)
        cfg.DATASETS.TRAIN, cfg.DATASETS.TEST = ("particle_Train",), ("particle_Train",)
        cfg.SOLVER.IMS_PER_BATCH, cfg.SOLVER.MAX_ITER, cfg.SOLVER.CHECKPOINT_PERIOD, cfg.SOLVER.BASE_LR = 3, 6, 10 ** 6, 1e-3
        cfg.MODEL.WEIGHTS, cfg.MODEL.ROI_HEADS.NUM_CLASSES = str(tmp_path / "init.pth"), 1
        cfg.DATALOADER.NUM_WORKERS = 2
        cfg.OUTPUT_DIR = str(tmp_path / mode)
        tr = DefaultTrainer(cfg)
        tr.resume_or_load(resume=False)
        seen = []

        class Rec:
            trainer = None
            def before_train(self): pass
            def after_train(self): pass
            def before_step(self): pass
            def after_step(self): seen.append(dict(tr.storage.latest()) if hasattr(tr.storage, "latest") else None)
        tr.register_hooks([Rec()])
        assert (tr._uploader is not None) and tr._uploader.device_resize == (mode == "device")
        tr.train()
        runs[mode] = seen
        del tr
    assert len(runs["device"]) == 6 and runs["device"] == runs["host"], (runs["device"][-1], runs["host"][-1])
