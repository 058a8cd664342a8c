"""DefaultTrainer end to end on a synthetic particle dataset (powder-sized 1024 x 1024 micrographs, ~480 polygon instances each):
seconds per iteration with DATALOADER.NUM_WORKERS = 0 (everything on the training thread) and 4 (threads prefetch and collate)."""
import sys, os, time, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import model_zoo, synth, checkpoint, params as P
from ampis_amd.config import get_cfg
from ampis_amd.data import DatasetCatalog, MetadataCatalog
from ampis_amd.engine import DefaultTrainer

N, S, B, ITERS = 32, 1024, 16, 12
# "multiscale": the model zoo's own training config (MIN_SIZE_TRAIN 640..800 sampled per image, MAX_SIZE_TRAIN 1333) on 1024 x 1536 micrographs:
# the batch frame changes from iteration to iteration, so the weight-gradient row tables (wgrad.hip) are re-used only through the cache --
# its hit rate is printed (the fixed 1024 x 1024 bench says nothing about that).  B = 8 there, 40 iterations.
MULTI = len(sys.argv) > 1 and sys.argv[1] == "multiscale"
if MULTI:
    N, B, ITERS = 16, 8, 40
dd = []
for i in range(N):
    img, gt = synth.micrograph(i, S, 1536 if MULTI else S)
    annos = [{"bbox": [float(v) for v in b], "bbox_mode": 0, "segmentation": [[float(v) for v in p]], "category_id": 0}
             for b, p in zip(gt["boxes"], gt["polygons"])]
    dd.append({"file_name": f"synthetic_{i}.png", "image_bgr": img, "height": S, "width": img.shape[1], "image_id": i, "annotations": annos})
tmp = tempfile.mkdtemp()
checkpoint.save_checkpoint(os.path.join(tmp, "init.pth"), P.init_params(1, seed=4, style="spread"))
out = {}
for workers in ((4,) if MULTI else (0, 4)):
    DatasetCatalog.clear()
    DatasetCatalog.register("particle_Train", lambda: dd)
    MetadataCatalog.get("particle_Train").set(thing_classes=["particle"])
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.DATASETS.TRAIN, cfg.DATASETS.TEST = ("particle_Train",), ("particle_Train",)
    cfg.SOLVER.IMS_PER_BATCH, cfg.SOLVER.MAX_ITER, cfg.SOLVER.CHECKPOINT_PERIOD, cfg.SOLVER.BASE_LR = B, ITERS, 10 ** 6, 1e-4
    cfg.MODEL.WEIGHTS, cfg.MODEL.ROI_HEADS.NUM_CLASSES = os.path.join(tmp, "init.pth"), 1
    if not MULTI:
        cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN = (S,), S
    cfg.DATALOADER.NUM_WORKERS = workers
    cfg.OUTPUT_DIR = os.path.join(tmp, f"out{workers}")
    tr = DefaultTrainer(cfg)
    tr.resume_or_load(resume=False)
    times = []
    class Clock:
        trainer = None
        def before_train(self): pass
        def after_train(self): pass
        def before_step(self): self.t = time.perf_counter()
        def after_step(self): times.append(time.perf_counter() - self.t)
    tr.register_hooks([Clock()]) if hasattr(tr, "register_hooks") else tr._hooks.append(Clock())
    # where an iteration goes: waiting for the loader, the forward / backward call, everything else (SGD, scalars, hooks)
    parts = {"loader_wait": [], "model_call": []}
    if tr.data_loader is None: tr._build_loader()
    _it = tr.data_loader
    class TimedLoader:
        def __iter__(self): return self
        def __next__(self):
            t = time.perf_counter(); b = next(_it); parts["loader_wait"].append(time.perf_counter() - t); return b
    tr.data_loader = TimedLoader()
    from ampis_amd.engine import defaults as _D
    if not hasattr(_D.TrainModel, "_orig_call"):
        _D.TrainModel._orig_call = _D.TrainModel.__call__
    def _timed_call(self, *a, **kw):
        t = time.perf_counter(); r = _D.TrainModel._orig_call(self, *a, **kw); parts["model_call"].append(time.perf_counter() - t); return r
    _D.TrainModel.__call__ = _timed_call
    tr.train()
    out[f"NUM_WORKERS={workers}"] = {"ms_per_iter_median": round(float(np.median(times[3:])) * 1e3, 1), "images_per_s": round(B / float(np.median(times[3:])), 1),
                                     "loader_wait_ms_median": round(float(np.median(parts["loader_wait"][3:])) * 1e3, 2),
                                     "model_call_ms_median": round(float(np.median(parts["model_call"][3:])) * 1e3, 2)}
    try:
        st = tr.ctx.rowtab_stats()
        st["hit_rate"] = round(st["hits"] / max(st["hits"] + st["misses"], 1), 4)
        out[f"NUM_WORKERS={workers}"]["wgrad_rowtab_cache"] = st
    except Exception as e:   # noqa: BLE001
        out[f"NUM_WORKERS={workers}"]["wgrad_rowtab_cache"] = repr(e)
    del tr
print(json.dumps(out))
