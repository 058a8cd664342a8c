"""CPU tests of the training host logic: LR schedule, EventStorage, annotation transforms, the sharded
train loader, hook ordering, and the N>1 gradient exchange with world_size-2 gloo."""
import os

import numpy as np
import pytest
import torch


def test_warmup_multistep_lr_matches_detectron2_formula():
    from ampis_amd.engine.train_loop import warmup_multistep_lr as lr
    # mask_rcnn_R_50_FPN_3x: base 0.02, steps (210000, 250000), gamma 0.1, warm-up 1000 iters from factor 0.001
    assert lr(0, 0.02, (210000, 250000), 0.1, 1000, 0.001) == pytest.approx(0.02 * 0.001)
    assert lr(500, 0.02, (210000, 250000), 0.1, 1000, 0.001) == pytest.approx(0.02 * (0.001 * 0.5 + 0.5))
    assert lr(1000, 0.02, (210000, 250000), 0.1, 1000, 0.001) == pytest.approx(0.02)
    assert lr(210000, 0.02, (210000, 250000), 0.1, 1000, 0.001) == pytest.approx(0.002)
    assert lr(260000, 0.02, (210000, 250000), 0.1, 1000, 0.001) == pytest.approx(0.0002)


def test_event_storage_surface():
    from ampis_amd.engine.train_loop import EventStorage
    st = EventStorage(5)
    st.put_scalar("validation_loss", 1.5)          # ampis/data_utils.py:104
    st.put_scalars(timetest=12)                    # ampis/data_utils.py:132
    st.step()
    st.put_scalar("validation_loss", 1.0)
    assert st.iter == 6
    assert st.history("validation_loss") == [(1.5, 5), (1.0, 6)]
    assert st.latest()["timetest"] == (12.0, 5)


def test_transform_annotations_scale_and_flip():
    from ampis_amd.data import transform_annotations
    annos = [{"bbox": [10, 20, 50, 60], "bbox_mode": 0, "segmentation": [[10.5, 20.5, 50.5, 20.5, 50.5, 60.5]], "category_id": 0},
             {"bbox": [0, 0, 0.000001, 5], "bbox_mode": 0, "segmentation": [[0, 0, 0, 5, 0, 2]], "category_id": 0}]   # empty after clip
    g = transform_annotations(annos, 0.5, 0.5, False, 100, 100)
    assert g["boxes"].tolist() == [[5, 10, 25, 30]] and len(g["polygons"]) == 1
    assert g["polygons"][0][:2].tolist() == [5.25, 10.25]
    f = transform_annotations(annos[:1], 0.5, 0.5, True, 100, 100)
    assert f["boxes"].tolist() == [[75, 10, 95, 30]]
    assert f["polygons"][0][0] == pytest.approx(100 - 5.25)
    with pytest.raises(ValueError, match="bitmask"):      # an RLE segmentation under MASK_FORMAT='polygon': detectron2's kind of refusal (the bitmask format takes it)
        transform_annotations([{"bbox": [0, 0, 5, 5], "segmentation": {"size": [5, 5], "counts": b"0"}, "category_id": 0}], 1, 1, False, 5, 5)


def _register(n=7):
    from ampis_amd.data import DatasetCatalog
    DatasetCatalog.clear()
    dd = [{"image_id": i, "image_bgr": np.full((40, 64, 3), i, np.uint8), "height": 40, "width": 64,
           "annotations": [{"bbox": [4, 4, 30, 30], "bbox_mode": 0, "segmentation": [[4, 4, 30, 4, 30, 30, 4, 30]], "category_id": 0}]} for i in range(n)]
    DatasetCatalog.register("cpu_Train", lambda: dd)


def test_vectorised_annotation_transform_equals_the_per_instance_form():
    """transform_annotations handles all instances of an image in a few array operations (and the mapper parses a dataset dict once);
    values must equal the per-instance statement of detectron2's transform bit for bit: XYWH boxes, crowd instances, boxes that
    collapse after clipping, flips, anisotropic scales -- and PackedGt must flatten both forms to the same amp_gt arrays."""
    from ampis_amd import data, synth
    from ampis_amd.model import PackedGt
    _, gt = synth.micrograph(3, 512, 512)
    annos = [{"bbox": [float(v) for v in b], "bbox_mode": 0, "segmentation": [[float(v) for v in p]], "category_id": i % 3}
             for i, (b, p) in enumerate(zip(gt["boxes"], gt["polygons"]))]
    assert len(annos) > 50
    annos[5]["bbox_mode"] = 1
    annos[5]["bbox"][2] -= annos[5]["bbox"][0]; annos[5]["bbox"][3] -= annos[5]["bbox"][1]
    annos[7]["iscrowd"] = 1
    annos[9]["bbox"] = [10.0, 10.0, 10.0, 50.0]            # empty: dropped with its polygon
    annos[11]["bbox"] = [600.0, 20.0, 700.0, 90.0]         # outside the frame after clipping
    parsed = data.parse_annotations(annos)
    for flip in (False, True):
        for sx, sy, W, H in ((1.0, 1.0, 512, 512), (0.78125, 0.78125, 400, 400), (1.3, 0.7, 666, 358)):
            ref = data._transform_annotations_loop(annos, sx, sy, flip, W, H)
            for got in (data.transform_annotations(annos, sx, sy, flip, W, H), data.transform_parsed(parsed, sx, sy, flip, W, H)):
                assert got["boxes"].dtype == np.float32 and got["classes"].dtype == np.int64
                assert np.array_equal(got["boxes"], ref["boxes"]) and np.array_equal(got["classes"], ref["classes"])
                assert len(got["polygons"]) == len(ref["polygons"]) == len(ref["boxes"]) < len(annos) - 1
                assert all(np.array_equal(a, b) for a, b in zip(got["polygons"], ref["polygons"]))
                a, b = PackedGt([got, got]), PackedGt([ref, ref])
                assert all(np.array_equal(x, y) for x, y in zip(a._keep, b._keep))
    odd = [dict(annos[0], segmentation=[[1.0, 2.0, 3.0]])]  # a dangling coordinate: the per-instance form decides
    assert data.parse_annotations(odd) is None
    got, ref = data.transform_annotations(odd, 0.5, 0.5, False, 512, 512), data._transform_annotations_loop(odd, 0.5, 0.5, False, 512, 512)
    assert len(got["polygons"]) == len(ref["polygons"]) == 1 and np.array_equal(got["polygons"][0], ref["polygons"][0]) and np.array_equal(got["boxes"], ref["boxes"])
    assert data.transform_annotations([], 1.0, 1.0, False, 8, 8)["boxes"].shape == (0, 4)


def test_train_loader_shards_the_same_stream():
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, build_detection_train_loader
    _register()
    cfg = get_cfg()
    cfg.DATASETS.TRAIN = ("cpu_Train",)
    cfg.SOLVER.IMS_PER_BATCH = 4
    cfg.INPUT.MIN_SIZE_TRAIN = (40,)
    cfg.INPUT.MAX_SIZE_TRAIN = 64
    one = build_detection_train_loader(cfg, rank=0, world_size=1, seed=3)
    r0 = build_detection_train_loader(cfg, rank=0, world_size=2, seed=3)
    r1 = build_detection_train_loader(cfg, rank=1, world_size=2, seed=3)
    for _ in range(5):
        full = [d["image_id"] for d in next(one)]
        a, b = [d["image_id"] for d in next(r0)], [d["image_id"] for d in next(r1)]
        assert len(full) == 4 and len(a) == 2 and len(b) == 2
        assert full == [a[0], b[0], a[1], b[1]]          # rank r takes elements r, r + world, ...
    d = next(one)[0]
    assert d["image_bgr"].shape == (40, 64, 3) and d["gt"]["boxes"].shape == (1, 4) and d["height"] == 40
    DatasetCatalog.clear()


def test_threaded_loader_equals_the_sequential_one():
    """DATALOADER.NUM_WORKERS threads prefetch the next batches; scales, flips, order and pixels are those of the sequential loader."""
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, build_detection_train_loader
    _register(11)
    def loader(workers):
        cfg = get_cfg()
        cfg.DATASETS.TRAIN = ("cpu_Train",)
        cfg.SOLVER.IMS_PER_BATCH = 3
        cfg.INPUT.MIN_SIZE_TRAIN = (32, 40, 48)
        cfg.INPUT.MAX_SIZE_TRAIN = 80
        cfg.DATALOADER.NUM_WORKERS = workers
        return build_detection_train_loader(cfg, rank=0, world_size=1, seed=5)
    seq, thr = loader(0), loader(3)
    flips = 0
    for _ in range(12):
        a, b = next(seq), next(thr)
        assert [d["image_id"] for d in a] == [d["image_id"] for d in b]
        for x, y in zip(a, b):
            assert x["image_bgr"].shape == y["image_bgr"].shape and np.array_equal(x["image_bgr"], y["image_bgr"])
            assert np.array_equal(x["gt"]["boxes"], y["gt"]["boxes"])
            flips += int(x["gt"]["boxes"][0][0] > 10)
    assert 0 < flips < 36, "both orientations occur"
    shapes = {d["image_bgr"].shape[0] for _ in range(6) for d in next(seq)}
    assert len(shapes) > 1, "MIN_SIZE_TRAIN is drawn per image: one batch may mix scales"
    thr.close()
    DatasetCatalog.clear()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ampis_amd.utils import comm
    g = torch.full((1000,), float(rank + 1))
    scale = comm.all_reduce_sum_(g)
    comm.synchronize()
    q.put((rank, float(g[0]), scale, comm.get_world_size(), comm.is_main_process()))
    dist.destroy_process_group()


def test_gradient_all_reduce_world_size_2_gloo():
    """The N>1 data path: SUM all-reduce of the flat gradient tensor + the 1/world factor, two CPU processes over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert [r[1] for r in res] == [3.0, 3.0]             # 1 + 2 on both ranks
    assert all(r[2] == 0.5 and r[3] == 2 for r in res)
    assert [r[4] for r in res] == [True, False]


def test_deferred_mapping_plans_exactly_what_the_host_mapping_does():
    """The device input path (round 4) leaves ResizeShortestEdge + flip + stacking to the uploader: the mapper's deferred form must hand over the
    decoded pixels untouched, the size PIL would produce, the flip, and the SAME transformed annotations; collate then carries no host frame
    but the same per-image sizes.  (The bytes of the device-built frame are compared on the GPU: tests/test_train_input_gpu.py.)"""
    import numpy as np
    from ampis_amd import model_zoo, synth
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetMapper, mapped_hw
    from ampis_amd.engine.defaults import TrainModel
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (160, 192, 224, 300), 260
    dicts = []
    for i, (h, w) in enumerate([(200, 280), (280, 200), (192, 192)]):
        img, gt = synth.micrograph(i, h, w, seed=5)
        annos = [{"bbox": [float(v) for v in b], "bbox_mode": 0, "segmentation": [[float(v) for v in p]], "category_id": 0}
                 for b, p in list(zip(gt["boxes"], gt["polygons"]))[:30]]
        dicts.append({"file_name": f"s{i}.png", "image_bgr": img, "height": h, "width": w, "image_id": i, "annotations": annos})
    m = DatasetMapper(cfg, True, seed=1)
    flips = 0
    for rep in range(12):
        plans = [(d,) + tuple(m.draw()) for d in dicts]
        host = [m.apply(*p) for p in plans]
        dev = [m.apply(*p, True) for p in plans]
        for p, a, b in zip(plans, host, dev):
            assert b["image_bgr"] is p[0]["image_bgr"]                                     # untouched, not even copied
            assert b["device_plan"] == (a["image_bgr"].shape[0], a["image_bgr"].shape[1], bool(p[2])) and mapped_hw(b) == mapped_hw(a)
            assert set(a["gt"]) == set(b["gt"])
            for k in ("boxes", "classes", "poly_flat", "poly_len"):
                assert np.array_equal(a["gt"][k], b["gt"][k]), k
            flips += int(p[2])
        fh, sh, gh = TrainModel.collate(host)
        fd, sd, gd = TrainModel.collate(dev)
        assert fd is None and sd == sh and fh.shape[1:3] == (max(mapped_hw(b)[0] for b in dev), max(mapped_hw(b)[1] for b in dev))
    assert 6 < flips < 30
