"""ampis_amd — MI355X (gfx950) native Mask R-CNN R50-FPN hot path behind the AMPIS / detectron2 surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/ampis_hip.h), the ctypes
binding, and the host-side mirror of the detectron2 members AMPIS touches.
"""
__version__ = "0.1.0"


def install_as_detectron2():
    """Register this package's façade under the `detectron2.*` (and `pycocotools.mask`) module names, so code written against
    detectron2's API -- the AMPIS modules, its notebooks, and pickles whose class path is
    detectron2.structures.instances.Instances -- runs unmodified on the MI355X path.  Refuses to shadow a real detectron2."""
    import importlib.util
    import sys
    import types

    existing = sys.modules.get("detectron2")
    if existing is not None and "ampis_amd" in (getattr(existing, "__doc__", "") or ""):
        return   # already installed
    if existing is not None or importlib.util.find_spec("detectron2") is not None:
        raise RuntimeError("a real detectron2 is importable; refusing to shadow it (import ampis_amd.* directly instead)")
    from . import config, data, engine, model_zoo, rle, structures
    from .engine import defaults, hooks
    from .structures import boxes, instances, masks
    from .utils import comm, logger, visualizer

    root = types.ModuleType("detectron2")
    root.__doc__ = "ampis_amd façade registered as detectron2"
    root.__path__ = []
    utils = types.ModuleType("detectron2.utils")
    utils.__path__ = []
    utils.comm, utils.logger, utils.visualizer = comm, logger, visualizer
    root.config, root.data, root.engine, root.model_zoo, root.structures, root.utils = config, data, engine, model_zoo, structures, utils
    mods = {
        "detectron2": root, "detectron2.config": config, "detectron2.data": data, "detectron2.engine": engine,
        "detectron2.engine.defaults": defaults, "detectron2.engine.hooks": hooks, "detectron2.model_zoo": model_zoo,
        "detectron2.structures": structures, "detectron2.structures.boxes": boxes,
        "detectron2.structures.instances": instances, "detectron2.structures.masks": masks,
        "detectron2.utils": utils, "detectron2.utils.comm": comm, "detectron2.utils.logger": logger,
        "detectron2.utils.visualizer": visualizer,
    }
    for k, v in mods.items():
        sys.modules.setdefault(k, v)
    if "pycocotools" not in sys.modules and importlib.util.find_spec("pycocotools") is None:
        pc = types.ModuleType("pycocotools")
        pc.__path__ = []
        pc.mask = rle
        sys.modules["pycocotools"] = pc
        sys.modules["pycocotools.mask"] = rle
    structures.Instances.__module__ = "detectron2.structures.instances"
