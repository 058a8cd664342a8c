import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ampis_amd import checkpoint, data_utils, model_zoo, params as P, rle, synth
from ampis_amd.config import get_cfg
from ampis_amd.engine import DefaultPredictor
from ampis_amd.engine.defaults import resize_shortest_edge
from oracle import maskrcnn as O
K, D = 1, 30
npp = P.init_params(K, seed=21, style="spread")
wpath = os.path.join(tempfile.mkdtemp(), "m.pth"); checkpoint.save_checkpoint(wpath, npp)
cfg = get_cfg(); cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
cfg.MODEL.ROI_HEADS.NUM_CLASSES = K; cfg.TEST.DETECTIONS_PER_IMAGE = D; cfg.DATASETS.TEST = ("a",); cfg.MODEL.WEIGHTS = wpath
cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 160, 256
img, _ = synth.micrograph(3, 240, 300)
pr = DefaultPredictor(cfg); outs = pr(img); p = outs["instances"]
small = resize_shortest_edge(img, 160, 256)
st = {}
ref = O.infer(small[None], O.to_torch_params(npp), O.Cfg(num_classes=K, detections_per_image=D), out_sizes=[(240, 300)], stages=st)[0]
rb, rm = ref["boxes"].numpy(), ref["masks"].numpy()
pb = p.pred_boxes.tensor.numpy()
for i in range(len(rb)):
    d = np.abs(pb - rb[i]).max(axis=1); j = int(np.argmin(d))
    gm = rle.decode(p.pred_masks.rle[j]).astype(bool); u = (gm | rm[i]).sum(); iou = 1.0 if u == 0 else (gm & rm[i]).sum() / u
    flag = "" if (d[j] < 1e-3 and iou >= 0.999) else "  <<<<"
    print(i, j, f"dbox {d[j]:.2e} iou {iou:.5f} area {rm[i].sum()} diffpx {(gm ^ rm[i]).sum()} score {ref['scores'][i]:.6f} {p.scores[j]:.6f}{flag}")
    if flag:
        ys, xs = np.nonzero(gm ^ rm[i]); print("   diff at", list(zip(ys[:8], xs[:8])), "box", rb[i], pb[j])
