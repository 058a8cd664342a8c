"""ampis_amd — MI355X (gfx950) native Mask R-CNN R50-FPN hot path behind the AMPIS / detectron2 surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/ampis_hip.h), the ctypes
binding, and the host-side mirror of the detectron2 members AMPIS touches.
"""
__version__ = "0.1.0"
