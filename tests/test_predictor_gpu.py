"""DefaultPredictor façade on the GPU: notebook cells 24-28 (cfg -> DefaultPredictor -> predictor(img) -> format_outputs)
against the oracle run on the same resized input; includes the 800/1333 resize and the rescale to the original size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_predictor_matches_oracle_with_resize(tmp_path):
    from ampis_amd import checkpoint, data_utils, model_zoo, params as P, rle, synth
    from ampis_amd.config import get_cfg
    from ampis_amd.engine import DefaultPredictor
    from ampis_amd.engine.defaults import resize_shortest_edge
    from oracle import maskrcnn as O

    K, D = 1, 30
    npp = P.init_params(K, seed=21, style="spread")
    wpath = tmp_path / "model_final.pth"
    checkpoint.save_checkpoint(wpath, npp)
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = K
    cfg.TEST.DETECTIONS_PER_IMAGE = D
    cfg.DATASETS.TEST = ("particle_Train",)
    cfg.MODEL.WEIGHTS = str(wpath)
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 160, 256      # small net input, exercises the resize + rescale
    img, _ = synth.micrograph(3, 240, 300)
    predictor = DefaultPredictor(cfg)
    outs = predictor(img)
    inst = outs["instances"]
    assert inst.image_size == (240, 300) and len(inst) > 0
    assert inst.pred_boxes.tensor.dtype == torch.float32 and inst.pred_classes.dtype == torch.int64
    assert torch.all(inst.scores[:-1] >= inst.scores[1:])

    small = resize_shortest_edge(img, 160, 256)
    assert small.shape[:2] == (160, 200)
    from oracle import gate
    refs, floor = gate.floor_of(lambda: O.infer(small[None], O.to_torch_params(npp), O.Cfg(num_classes=K, detections_per_image=D), out_sizes=[(240, 300)]), (240, 300))
    ref = refs[0]
    res = data_utils.format_outputs("a.png", "particle_Train", outs)
    p = res["pred"]["instances"]
    # the gate of oracle/gate.py on what format_outputs stores (numpy boxes / scores / classes, RLE dicts)
    from oracle import gate
    hip = dict(boxes=np.asarray(p.pred_boxes), scores=np.asarray(p.scores), classes=np.asarray(p.pred_classes), masks=list(p.pred_masks))
    st = gate.check_image(hip, ref, 240, 300, lambda m: rle.decode(m).astype(bool))
    print("predictor gate:", gate.summary(st))
    assert st["instances"] > 5
    print("predictor gate |", gate.assert_floor(st, floor, sigmas=3.0, floor_sigmas=2.5))      # (30 instances pasted into an upscaled frame: many single-pixel ties, in the oracle's own noise too)


def test_predictor_refuses_cpu_device():
    from ampis_amd import _lib
    from ampis_amd.config import get_cfg
    from ampis_amd.engine import DefaultPredictor
    cfg = get_cfg()
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
    with pytest.raises(_lib.AmpError):
        DefaultPredictor(cfg)


def test_predictor_stream_equals_single_calls(tmp_path):
    """DefaultPredictor.stream (amp_pipeline, two images in flight) against predictor(img) image by image: differently sized
    micrographs, same boxes / scores / classes / RLE bytes, in order."""
    from ampis_amd import checkpoint, model_zoo, params as P, synth
    from ampis_amd.config import get_cfg
    from ampis_amd.engine import DefaultPredictor
    K, D = 1, 25
    wpath = tmp_path / "model_final.pth"
    checkpoint.save_checkpoint(wpath, P.init_params(K, seed=21, style="spread"))
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = K
    cfg.TEST.DETECTIONS_PER_IMAGE = D
    cfg.DATASETS.TEST = ("particle_Train",)
    cfg.MODEL.WEIGHTS = str(wpath)
    cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST = 160, 256
    imgs = [synth.micrograph(i, h, w)[0] for i, (h, w) in enumerate(((240, 300), (200, 200), (180, 320), (256, 256), (150, 290)))]
    predictor = DefaultPredictor(cfg)

    def sig(o):
        i = o["instances"]
        return (i.image_size, i.pred_boxes.tensor.numpy().tobytes(), i.scores.numpy().tobytes(), i.pred_classes.numpy().tobytes(),
                tuple(m["counts"] for m in i.pred_masks.rle))

    want = [sig(predictor(im)) for im in imgs]
    assert sum(len(w[4]) for w in want) > 20
    for depth in (2, 3):
        got = [sig(o) for o in predictor.stream(iter(imgs), depth=depth)]
        assert got == want, depth
    assert [sig(o) for o in predictor.stream(imgs[:1])] == want[:1]
    predictor.close()
