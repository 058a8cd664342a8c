"""Latency of the notebook flow (cells 24-28): DefaultPredictor(cfg)(img) on one 1024x1536 powder-sized micrograph, batch 1,
host uint8 in -> Instances out (boxes, scores, classes, lazily decoded RLE masks), DETECTIONS_PER_IMAGE=400 as the tutorial sets."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import model_zoo, synth, checkpoint, params as P, data_utils
from ampis_amd.config import get_cfg
from ampis_amd.engine import DefaultPredictor

tmp = "/tmp/_bench_pred_init.pth"
checkpoint.save_checkpoint(tmp, P.init_params(1, seed=4, style="spread"))
cfg = get_cfg()
cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
cfg.TEST.DETECTIONS_PER_IMAGE = 400
cfg.DATASETS.TEST = ("particle_Train",)
cfg.MODEL.WEIGHTS = tmp
for size in ((800, 1333), (1024, 1536)):
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = size
    pred = DefaultPredictor(cfg)
    img, _ = synth.micrograph(0, 1024, 1536)
    for _ in range(3):
        out = pred(img)
    t = time.perf_counter(); n = 10
    for _ in range(n):
        out = pred(img)
    dt = (time.perf_counter() - t) / n
    t = time.perf_counter()
    res = data_utils.format_outputs("a.png", "particle_Train", out)
    t_fmt = time.perf_counter() - t
    print(json.dumps({"min_max_size_test": size, "predictor_ms_per_image": round(dt * 1e3, 2), "detections": len(out["instances"]),
                      "format_outputs_ms": round(t_fmt * 1e3, 2)}), flush=True)
