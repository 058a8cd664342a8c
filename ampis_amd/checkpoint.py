"""Weights I/O: read detectron2 checkpoints -- model-zoo `model_final_*.pkl` (pickled dict with a 'model'
dict of numpy arrays under detectron2 names, notebook cell 20) and `*.pth` (torch.save'd {'model': state_dict}, the files
notebook cell 24 picks with sorted(glob('*.pth'))[-1]) -- into the name -> ndarray dict amp_model_load_tensor consumes; write
`.pth` files of the same shape.  No network: URLs / detectron2:// paths are refused with a clear message."""
import os
import pickle

import numpy as np
import torch

from . import params as P


def load_checkpoint(path, num_classes, arch="R50"):
    path = str(path)
    if path.startswith(("http://", "https://", "detectron2://")):
        raise FileNotFoundError(f"{path}: fetching weights needs a network; put the file on disk and set cfg.MODEL.WEIGHTS to its path")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        state = data["model"] if isinstance(data, dict) and "model" in data else data
    else:
        data = torch.load(path, map_location="cpu", weights_only=False)
        state = data["model"] if isinstance(data, dict) and "model" in data else data
    out = {}
    want = P.param_shapes(num_classes, arch)
    for name, shape in want.items():
        if name not in state:
            raise KeyError(f"{path}: checkpoint has no tensor '{name}'")
        a = state[name]
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{path}: '{name}' has shape {tuple(a.shape)}, the model needs {tuple(shape)} "
                             f"(MODEL.ROI_HEADS.NUM_CLASSES={num_classes}?)")
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def save_checkpoint(path, np_params, iteration=None):
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_params.items()}
    obj = {"model": state, "__author__": "ampis_amd"}
    if iteration is not None:
        obj["iteration"] = int(iteration)
    torch.save(obj, str(path))


def checkpoint_iteration(path):
    """iteration stored in a .pth written by save_checkpoint (None when absent)."""
    try:
        d = torch.load(str(path), map_location="cpu", weights_only=False)
        return int(d["iteration"]) if isinstance(d, dict) and "iteration" in d else None
    except Exception:
        return None
