"""ctypes binding of libampis_hip.so (the C ABI declared in include/ampis_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libampis_hip.so")


class AmpError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad", "relu", "res_mode", "out_mode")]


_lib = None


def lib():
    """Load (once) and return the shared library; raises AmpError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmpError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ampis_amd/csrc`). There is no CPU fallback for the hot path.")
    L = C.CDLL(LIB_PATH)
    L.amp_last_error.restype = C.c_char_p
    L.amp_stream.restype = C.c_void_p
    _declare(L)
    _lib = L
    return L


def _declare(L):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    sig = {
        "amp_version": ([], i),
        "amp_init": ([i, vp, C.POINTER(vp)], i),
        "amp_destroy": ([vp], None),
        "amp_sync": ([vp], i),
        "amp_malloc": ([vp, C.c_size_t, C.POINTER(vp)], i),
        "amp_free": ([vp, vp], i),
        "amp_memcpy_h2d": ([vp, vp, vp, C.c_size_t], i),
        "amp_memcpy_d2h": ([vp, vp, vp, C.c_size_t], i),
        "amp_memset": ([vp, vp, i, C.c_size_t], i),
        "amp_conv2d_nhwc": ([vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp], i),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    L._amp_sig = sig


def check(status, what=""):
    if status != 0:
        msg = lib().amp_last_error().decode(errors="replace")
        raise AmpError(f"{what or 'libampis_hip'} failed with status {status}: {msg}")


class Context:
    """One amp_ctx (one HIP stream). `stream` may be a raw hipStream_t (e.g. torch's current stream)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        check(lib().amp_init(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)), "amp_init")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    @property
    def stream(self):
        return lib().amp_stream(self._h)

    def sync(self):
        check(lib().amp_sync(self._h), "amp_sync")

    def close(self):
        if self._h:
            lib().amp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return C.c_void_p()
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    if hasattr(t, "ctypes"):
        return C.c_void_p(t.ctypes.data)
    return C.c_void_p(int(t))
