// Shared helpers for the ampis_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/ampis_hip.h"

namespace amp {

void set_error(const char* fmt, ...);

#define AMP_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            amp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                           hipGetErrorString(_e));                                  \
            return AMP_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

#define AMP_REQUIRE(cond, ...)                                                      \
    do {                                                                            \
        if (!(cond)) {                                                              \
            amp::set_error(__VA_ARGS__);                                            \
            return AMP_ERR_ARG;                                                     \
        }                                                                           \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD under the observed
// round-robin dispatch, so hand each XCD a contiguous chunk of the logical grid
// (neighbouring tiles share operand panels in that XCD's L2). Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace amp

struct amp_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
};
