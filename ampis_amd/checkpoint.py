"""Weights I/O with detectron2's DetectionCheckpointer semantics.

What the reference's workflow loads (notebook cell 20 / 24, GETTING_STARTED.md:30):
  * model-zoo `model_final_*.pkl`: pickled {'model': {detectron2 name: ndarray}, '__author__': ...} -- the COCO Mask R-CNN with 80
    classes, fine-tuned with MODEL.ROI_HEADS.NUM_CLASSES = 1: the 81-way cls_score, 320-row bbox_pred and 80-channel mask predictor
    do not fit and detectron2 SKIPS them with a warning, keeping their fresh initialisation;
  * `detectron2://ImageNetPretrained/MSRA/R-50.pkl` (what the tutorial's cfg.MODEL.WEIGHTs typo leaves in effect):
    a caffe2-named backbone ({'conv1_w', 'res_conv1_bn_s', 'res2_0_branch2a_w', ..., 'fc1000_w'}), renamed by
    convert_basic_c2_names and matched to `backbone.bottom_up.*` by suffix; FPN, RPN and both heads keep their initialisation;
  * `*.pth` (torch.save'd {'model': state_dict, 'iteration': n, ...}): what training writes and notebook cell 24 reads back.

So: a tensor that is missing from the file, or whose shape differs, keeps the value of `init` (detectron2-style initialisation by
default) and is reported -- never an exception; tensors of the file the model has no use for are reported as unexpected.  `strict=True`
restores the all-or-nothing behaviour (used where a partial load would be a bug, e.g. resuming one's own checkpoint).
No network: URLs / detectron2:// paths are refused with a clear message.
"""
import logging
import os
import pickle
import re

import numpy as np
import torch

from . import params as P

logger = logging.getLogger("ampis_amd")


class LoadReport(dict):
    """{'missing': [...], 'shape_mismatch': [(name, file shape, model shape)], 'unexpected': [...], 'renamed': n, 'source': 'd2'|'d2-suffix'|'caffe2'}"""


def convert_c2_backbone_names(keys):
    """detectron2 checkpoint/c2_model_loading.py convert_basic_c2_names, restated: caffe2 / Detectron blob names of a ResNet backbone
    -> detectron2 module names (without the `backbone.bottom_up.` prefix).  Returns {old: new}."""
    out = {}
    for old in keys:
        k = old.replace("_", ".")
        k = re.sub(r"\.b$", ".bias", k)
        k = re.sub(r"\.w$", ".weight", k)
        k = re.sub(r"bn\.s$", "norm.weight", k)
        k = re.sub(r"bn\.bias$", "norm.bias", k)
        k = re.sub(r"bn\.rm", "norm.running_mean", k)
        k = re.sub(r"bn\.running\.mean$", "norm.running_mean", k)
        k = re.sub(r"bn\.riv$", "norm.running_var", k)
        k = re.sub(r"bn\.running\.var$", "norm.running_var", k)
        k = re.sub(r"bn\.gamma$", "norm.weight", k)
        k = re.sub(r"bn\.beta$", "norm.bias", k)
        k = re.sub(r"gn\.s$", "norm.weight", k)
        k = re.sub(r"gn\.bias$", "norm.bias", k)
        k = re.sub(r"^res\.conv1\.norm\.", "conv1.norm.", k)       # the stem's affine: res_conv1_bn_s
        k = re.sub(r"^conv1\.", "stem.conv1.", k)
        k = k.replace(".branch1.", ".shortcut.").replace(".branch2a.", ".conv1.").replace(".branch2b.", ".conv2.").replace(".branch2c.", ".conv3.")
        out[old] = k
    return out


def _to_numpy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def read_state(path):
    """File -> (state dict name -> ndarray in detectron2 names, source tag, number of renamed keys).

    Two independent switches, as in detectron2's DetectionCheckpointer._load_file / _load_model: blob names are converted only when the
    file says `__author__ == 'Caffe2'` (or is a bare blob dict, which detectron2 tags that way); `matching_heuristics` only turns on
    matching by name suffix.  source: 'd2' exact names | 'd2-suffix' detectron2 names matched by suffix (e.g. the torchvision-converted
    ImageNetPretrained R-50.pkl, whose keys are `stem.conv1.norm.running_mean` ...) | 'caffe2' converted names matched by suffix."""
    path = str(path)
    if path.startswith(("http://", "https://", "detectron2://")):
        raise FileNotFoundError(f"{path}: fetching weights needs a network; put the file on disk and set cfg.MODEL.WEIGHTS to its path")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
    else:
        data = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(data, dict) and "model" in data and isinstance(data["model"], dict):
        state, caffe2 = data["model"], data.get("__author__") == "Caffe2"
        heuristics = caffe2 or bool(data.get("matching_heuristics"))
    else:                                                          # a bare blob dict: Detectron / MSRA ImageNet files
        state = data["blobs"] if isinstance(data, dict) and "blobs" in data else data
        caffe2 = heuristics = True
    state = {k: _to_numpy(v) for k, v in state.items() if not str(k).endswith("_momentum")}
    if not caffe2:
        return state, ("d2-suffix" if heuristics else "d2"), 0
    ren = convert_c2_backbone_names(state.keys())
    return {ren[k]: v for k, v in state.items()}, "caffe2", sum(1 for k, v in ren.items() if k != v)


def frozen_bn_defaults(name, shape):
    """detectron2 FrozenBatchNorm2d buffers as constructed: weight 1, bias 0, running_mean 0, running_var 1 - eps (so that an affine-only
    caffe2 'bn' -- scale and bias, no statistics -- is applied exactly: weight * rsqrt(1 - eps + eps) = weight)."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "weight":
        return np.ones(shape, np.float32)
    if leaf == "running_var":
        return np.full(shape, np.float32(1.0) - np.float32(P.BN_EPS), np.float32)
    return np.zeros(shape, np.float32)


def load_checkpoint(path, num_classes, arch="R50", init=None, strict=False, seed=0, with_report=False):
    """-> OrderedDict name -> ndarray for amp_model_load_tensor (and the LoadReport when with_report).  See the module docstring."""
    state, source, renamed = read_state(path)
    want = P.param_shapes(num_classes, arch)
    if init is None:
        init = P.init_params(num_classes, seed=seed, style="d2", arch=arch)
    # caffe2 names carry no module prefix: match by suffix, as detectron2's align_and_update_state_dicts does (longest suffix wins)
    if source in ("caffe2", "d2-suffix"):
        by_suffix = {}
        for k in state:
            cands = [n for n in want if n == k or n.endswith("." + k)]
            if cands:
                by_suffix[max(cands, key=len)] = k
        lookup = by_suffix
    else:
        lookup = {n: n for n in want if n in state}
    rep = LoadReport(missing=[], shape_mismatch=[], unexpected=[], renamed=renamed, source=source)
    out = type(want)()
    used = set()
    for name, shape in want.items():
        src = lookup.get(name)
        a = None
        if src is not None:
            used.add(src)
            a = state[src]
            if tuple(a.shape) != tuple(shape):
                rep["shape_mismatch"].append((name, tuple(a.shape), tuple(shape)))
                a = None
        else:
            rep["missing"].append(name)
        if a is None:
            a = frozen_bn_defaults(name, shape) if ".norm." in name else init[name]
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    rep["unexpected"] = sorted(k for k in state if k not in used)
    if strict and (rep["missing"] or rep["shape_mismatch"]):
        first = rep["missing"][:1] or [f"{n}: file {fs}, model {ms}" for n, fs, ms in rep["shape_mismatch"][:1]]
        raise ValueError(f"{path}: strict load failed ({len(rep['missing'])} missing, {len(rep['shape_mismatch'])} shape mismatches; first: {first[0]}; "
                         f"MODEL.ROI_HEADS.NUM_CLASSES={num_classes}?)")
    if rep["shape_mismatch"]:
        logger.warning("Skip loading parameters with a different shape (kept at their initialisation): "
                       + "; ".join(f"'{n}' checkpoint {fs} vs model {ms}" for n, fs, ms in rep["shape_mismatch"]))
    if rep["missing"]:
        groups = sorted({n.rsplit(".", 2)[0] if ".norm." in n else n.rsplit(".", 1)[0] for n in rep["missing"]})
        logger.warning(f"Some model parameters or buffers are not found in the checkpoint (kept at their initialisation): "
                       f"{len(rep['missing'])} tensors under {', '.join(groups[:12])}{' ...' if len(groups) > 12 else ''}")
    if rep["unexpected"]:
        logger.warning(f"The checkpoint contains {len(rep['unexpected'])} tensors the model does not use: {', '.join(rep['unexpected'][:8])}"
                       f"{' ...' if len(rep['unexpected']) > 8 else ''}")
    return (out, rep) if with_report else out


def save_checkpoint(path, np_params, iteration=None, optimizer=None):
    """detectron2-shaped `.pth`: {'model': state_dict, 'iteration': n, 'optimizer': {...}} (the optimizer entry carries the SGD
    momentum buffers under the parameters' names, so that resume=True continues with the velocity it stopped with)."""
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_params.items()}
    obj = {"model": state, "__author__": "ampis_amd"}
    if iteration is not None:
        obj["iteration"] = int(iteration)
    if optimizer is not None:
        obj["optimizer"] = {"momentum_buffers": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in optimizer.items()}}
    torch.save(obj, str(path))


def checkpoint_iteration(path):
    """iteration stored in a .pth written by save_checkpoint (None when absent)."""
    try:
        d = torch.load(str(path), map_location="cpu", weights_only=False)
        return int(d["iteration"]) if isinstance(d, dict) and "iteration" in d else None
    except Exception:
        return None


def checkpoint_momentum(path):
    """name -> ndarray of the SGD momentum buffers stored by save_checkpoint ({} when the file has none)."""
    try:
        d = torch.load(str(path), map_location="cpu", weights_only=False)
        opt = d.get("optimizer", {}) if isinstance(d, dict) else {}
        if isinstance(opt, dict) and "momentum_buffers" not in opt and "state" in opt and "param_groups" in opt:
            # a detectron2 / reference checkpoint: torch's SGD.state_dict() indexes the buffers by parameter POSITION in an optimizer this
            # process never built, so they cannot be mapped to names reliably -- say so instead of dropping them silently
            import logging
            logging.getLogger("ampis_amd").warning(
                f"{path}: the optimizer state is torch's SGD.state_dict() ({len(opt['state'])} buffers by parameter index); momentum is NOT "
                "restored from it (weights and iteration are) -- the first steps after this resume run with zero momentum")
            return {}
        mb = opt.get("momentum_buffers", {}) if isinstance(opt, dict) else {}
        return {k: _to_numpy(v).astype(np.float32) for k, v in mb.items()}
    except Exception:
        return {}
