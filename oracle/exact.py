"""Noise floor of the fp32 oracle (TEST INFRASTRUCTURE).  `exact_convs()` re-runs oracle.maskrcnn with every convolution, linear
layer and transposed convolution evaluated in fp64 and rounded ONCE to fp32: the same network and the same fp32 tensors, without
the summation-order noise of an fp32 GEMM.  The distance between that run and the plain fp32 oracle is how precisely the
reference's own arithmetic (torch CPU fp32 = detectron2's CPU path) defines an output; a gate against the fp32 oracle cannot
meaningfully be tighter than that.  Measured on 1024x1024 micrographs with 200 detections (tools/oracle_noise_floor.py): boxes up
to 1.25e-3 px apart (a 224 x 741 px box: 1.7 ppm of its side), 13 of 200 boxes more than 5e-4 px apart."""
import contextlib

import torch.nn.functional as F

from . import maskrcnn as O


class _ExactF:
    def __getattr__(self, name):
        return getattr(F, name)

    @staticmethod
    def _d(t):
        return None if t is None else t.double()

    def conv2d(self, x, w, b=None, **kw):
        return F.conv2d(x.double(), w.double(), self._d(b), **kw).float()

    def linear(self, x, w, b=None):
        return F.linear(x.double(), w.double(), self._d(b)).float()

    def conv_transpose2d(self, x, w, b=None, **kw):
        return F.conv_transpose2d(x.double(), w.double(), self._d(b), **kw).float()


@contextlib.contextmanager
def exact_convs():
    keep = O.F
    O.F = _ExactF()
    try:
        yield
    finally:
        O.F = keep
