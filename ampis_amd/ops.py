"""Per-stage entry points of libampis_hip.so on torch CUDA tensors (torch = device memory + stream only).

Used by the parity tests and by tools; the end-to-end path (ampis_amd.engine) calls amp_infer instead.
Every function launches on the stream of the Context it is given and never falls back to torch math.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, lib, ptr


def torch_context(device=0):
    """Context bound to torch's current HIP stream on `device` (so torch ops and ours are ordered)."""
    if not torch.cuda.is_available():
        raise _lib.AmpError("no HIP device visible: the ampis_amd hot path has no CPU fallback")
    with torch.cuda.device(device):
        return _lib.Context(device, borrow_stream=torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()), "need contiguous fp32 CUDA tensor"
    return t


def conv2d_nhwc(ctx, x, w, scale=None, shift=None, res=None, stride=1, pad=0, relu=False, res_mode=None,
                deconv2x2=False):
    """x [B,H,W,Cin], w [Cout,KH,KW,Cin] -> y [B,Ho,Wo,Cout] (or [B,2Ho,2Wo,Cout/4] when deconv2x2)."""
    _f32c(x), _f32c(w), _f32c(scale), _f32c(shift), _f32c(res)
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if res_mode is None:
        res_mode = 0 if res is None else 1
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), int(res_mode), int(deconv2x2))
    if deconv2x2:
        y = torch.empty((B, 2 * Ho, 2 * Wo, Cout // 4), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    check(lib().amp_conv2d_nhwc(ctx.handle, C.byref(d), ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(res), ptr(y)),
          "amp_conv2d_nhwc")
    return y
