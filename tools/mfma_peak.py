"""What the matrix pipe sustains on THIS chip (register-only back-to-back MFMAs, every CU busy, ~0.5 s per point): the practical
ceiling next to the data-sheet peaks bench.py divides by (fp32 157.3, f16 2500 TFLOP/s).  On constant operands the chip holds its
full clock and reaches the data sheet; on random operands (what a GEMM feeds it) it lowers the clock -- that rate is the one a real
kernel can approach."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ampis_amd import _lib
ctx = _lib.Context(0)
L = _lib.lib()
L.amp_debug_mfma_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
out = {}
for kind, name in ((0, "v_mfma_f32_32x32x2_f32 constant operands"), (2, "v_mfma_f32_32x32x2_f32 random operands"),
                   (1, "v_mfma_f32_32x32x16_f16 constant operands"), (3, "v_mfma_f32_32x32x16_f16 random operands")):
    for wps in (1, 2):
        t = C.c_float()
        _lib.check(L.amp_debug_mfma_peak(ctx.handle, kind, 200000 if kind % 2 == 0 else 400000, wps, C.byref(t)), "amp_debug_mfma_peak")
        out[f"{name}, {wps} wave/SIMD"] = round(t.value, 1)
print(json.dumps(out, indent=1))
