"""cfg façade: the subset of detectron2.config AMPIS touches.

`get_cfg()` returns a CfgNode tree with detectron2's defaults for the keys the Mask R-CNN R50-FPN path reads;
attribute get/set works on any key, unknown keys are accepted silently (the tutorial's `cfg.MODEL.WEIGHTs` typo,
notebook cell 20, relies on that).  `merge_from_file` understands the model-zoo name AMPIS uses and plain yaml files.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge_from_dict(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        """`path` is what model_zoo.get_config_file returned (a zoo name) or a yaml file with detectron2 keys."""
        from . import model_zoo
        name = str(path)
        if name.startswith(model_zoo.ZOO_PREFIX):
            self.merge_from_dict(model_zoo.zoo_overrides(name[len(model_zoo.ZOO_PREFIX):]))
            return
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        base = d.pop("_BASE_", None)
        if base:
            self.merge_from_file(os.path.join(os.path.dirname(path), base))
        self.merge_from_dict(d)

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v

    def freeze(self):
        pass

    def defrost(self):
        pass

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}
        return yaml.safe_dump(plain(self))


_DEFAULTS = {
    "VERSION": 2,
    "MODEL": {
        "META_ARCHITECTURE": "GeneralizedRCNN", "DEVICE": "cuda", "WEIGHTS": "", "MASK_ON": False,
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0],
        "BACKBONE": {"NAME": "build_resnet_backbone", "FREEZE_AT": 2},
        "RESNETS": {"DEPTH": 50, "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_IN_1X1": True,
                    "OUT_FEATURES": ["res4"], "NORM": "FrozenBN"},
        "FPN": {"IN_FEATURES": [], "OUT_CHANNELS": 256},
        "ANCHOR_GENERATOR": {"SIZES": [[32, 64, 128, 256, 512]], "ASPECT_RATIOS": [[0.5, 1.0, 2.0]]},
        "RPN": {"IN_FEATURES": ["res4"], "PRE_NMS_TOPK_TRAIN": 12000, "PRE_NMS_TOPK_TEST": 6000,
                "POST_NMS_TOPK_TRAIN": 2000, "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7,
                "BATCH_SIZE_PER_IMAGE": 256, "POSITIVE_FRACTION": 0.5, "IOU_THRESHOLDS": [0.3, 0.7]},
        "ROI_HEADS": {"NAME": "Res5ROIHeads", "NUM_CLASSES": 80, "BATCH_SIZE_PER_IMAGE": 512,
                      "POSITIVE_FRACTION": 0.25, "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5,
                      "IOU_THRESHOLDS": [0.5], "IN_FEATURES": ["res4"]},
        "ROI_BOX_HEAD": {"NAME": "", "NUM_FC": 0, "FC_DIM": 1024, "POOLER_RESOLUTION": 14,
                         "BBOX_REG_WEIGHTS": [10.0, 10.0, 5.0, 5.0]},
        "ROI_MASK_HEAD": {"NAME": "MaskRCNNConvUpsampleHead", "NUM_CONV": 0, "POOLER_RESOLUTION": 14},
    },
    "INPUT": {"MIN_SIZE_TRAIN": (800,), "MAX_SIZE_TRAIN": 1333, "MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333,
              "FORMAT": "BGR", "MASK_FORMAT": "polygon", "RANDOM_FLIP": "horizontal"},
    "DATASETS": {"TRAIN": (), "TEST": ()},
    "DATALOADER": {"NUM_WORKERS": 4},
    "SOLVER": {"IMS_PER_BATCH": 16, "BASE_LR": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0001, "WEIGHT_DECAY_NORM": 0.0,
               "MAX_ITER": 40000, "STEPS": (30000,), "GAMMA": 0.1, "WARMUP_ITERS": 1000, "WARMUP_FACTOR": 0.001,
               "CHECKPOINT_PERIOD": 5000},
    "TEST": {"DETECTIONS_PER_IMAGE": 100, "EVAL_PERIOD": 0},
    "OUTPUT_DIR": "./output",
    "SEED": -1,
}


def get_cfg():
    return CfgNode(copy.deepcopy(_DEFAULTS))
