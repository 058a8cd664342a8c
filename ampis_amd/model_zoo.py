"""model_zoo façade (detectron2.model_zoo members AMPIS uses: get_config_file, get_checkpoint_url; notebook cell 20,
GETTING_STARTED.md:30).  No network: get_config_file returns a token that cfg.merge_from_file resolves to the built-in
overrides of that zoo config; get_checkpoint_url returns detectron2's URL string, which the loaders refuse to fetch."""

ZOO_PREFIX = "ampis_amd-zoo://"

# Base-RCNN-FPN.yaml + mask_rcnn_R_50_FPN_3x.yaml ([D2-KNOWLEDGE])
_RCNN_FPN = {
    "MODEL": {
        "META_ARCHITECTURE": "GeneralizedRCNN", "MASK_ON": True,
        "WEIGHTS": "detectron2://ImageNetPretrained/MSRA/R-50.pkl",
        "BACKBONE": {"NAME": "build_resnet_fpn_backbone"},
        "RESNETS": {"OUT_FEATURES": ["res2", "res3", "res4", "res5"], "DEPTH": 50},
        "FPN": {"IN_FEATURES": ["res2", "res3", "res4", "res5"]},
        "ANCHOR_GENERATOR": {"SIZES": [[32], [64], [128], [256], [512]], "ASPECT_RATIOS": [[0.5, 1.0, 2.0]]},
        "RPN": {"IN_FEATURES": ["p2", "p3", "p4", "p5", "p6"], "PRE_NMS_TOPK_TRAIN": 2000, "PRE_NMS_TOPK_TEST": 1000,
                "POST_NMS_TOPK_TRAIN": 1000, "POST_NMS_TOPK_TEST": 1000},
        "ROI_HEADS": {"NAME": "StandardROIHeads", "IN_FEATURES": ["p2", "p3", "p4", "p5"]},
        "ROI_BOX_HEAD": {"NAME": "FastRCNNConvFCHead", "NUM_FC": 2, "POOLER_RESOLUTION": 7},
        "ROI_MASK_HEAD": {"NAME": "MaskRCNNConvUpsampleHead", "NUM_CONV": 4, "POOLER_RESOLUTION": 14},
    },
    "DATASETS": {"TRAIN": ("coco_2017_train",), "TEST": ("coco_2017_val",)},
    "SOLVER": {"IMS_PER_BATCH": 16, "BASE_LR": 0.02, "STEPS": (210000, 250000), "MAX_ITER": 270000},
    "INPUT": {"MIN_SIZE_TRAIN": (640, 672, 704, 736, 768, 800)},
    "VERSION": 2,
}



def _variant(**model):
    import copy
    c = copy.deepcopy(_RCNN_FPN)
    for k, v in model.items():
        if isinstance(v, dict):
            c["MODEL"].setdefault(k, {}).update(v)
        else:
            c["MODEL"][k] = v
    return c


# mask_rcnn_X_101_32x8d_FPN_3x.yaml (BASELINE configs[4]): ResNeXt-101 32x8d, stride in the 3x3, caffe2-style pixel std
_X101_FPN = _variant(WEIGHTS="detectron2://ImageNetPretrained/FAIR/X-101-32x8d.pkl", PIXEL_STD=[57.375, 57.120, 58.395],
                     RESNETS={"DEPTH": 101, "NUM_GROUPS": 32, "WIDTH_PER_GROUP": 8, "STRIDE_IN_1X1": False})
_R101_FPN = _variant(WEIGHTS="detectron2://ImageNetPretrained/MSRA/R-101.pkl", RESNETS={"DEPTH": 101})

_CONFIGS = {"COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml": _RCNN_FPN,
            "COCO-InstanceSegmentation/mask_rcnn_R_101_FPN_3x.yaml": _R101_FPN,
            "COCO-InstanceSegmentation/mask_rcnn_X_101_32x8d_FPN_3x.yaml": _X101_FPN}
_D2 = "https://dl.fbaipublicfiles.com/detectron2/COCO-InstanceSegmentation/"
_CHECKPOINTS = {"COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml": _D2 + "mask_rcnn_R_50_FPN_3x/137849600/model_final_f10217.pkl",
                "COCO-InstanceSegmentation/mask_rcnn_R_101_FPN_3x.yaml": _D2 + "mask_rcnn_R_101_FPN_3x/138205316/model_final_a3ec72.pkl",
                "COCO-InstanceSegmentation/mask_rcnn_X_101_32x8d_FPN_3x.yaml": _D2 + "mask_rcnn_X_101_32x8d_FPN_3x/139653917/model_final_2d9806.pkl"}


def get_config_file(config_path):
    if config_path not in _CONFIGS:
        raise RuntimeError(f"{config_path} not available in this model zoo (supported: {sorted(_CONFIGS)})")
    return ZOO_PREFIX + config_path


def zoo_overrides(config_path):
    import copy
    return copy.deepcopy(_CONFIGS[config_path])


def get_checkpoint_url(config_path):
    if config_path not in _CHECKPOINTS:
        raise RuntimeError(f"{config_path} not available in this model zoo")
    return _CHECKPOINTS[config_path]


def get_config(config_path, trained=False):
    from .config import get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(get_config_file(config_path))
    if trained:
        cfg.MODEL.WEIGHTS = get_checkpoint_url(config_path)
    return cfg
