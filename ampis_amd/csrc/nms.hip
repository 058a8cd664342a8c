// Batched greedy NMS on score-sorted boxes (torchvision.ops.batched_nms, vanilla semantics:
// boxes of different categories never suppress each other, exact coordinates, suppress when IoU > thresh).
//
// Two kernels per call, all images of the batch at once:
//   nms_mask_kernel : 64x64 tiles of the upper-triangular suppression matrix; lane i of a wave owns box i of the row tile
//                     and builds one 64-bit word (bit j = "i suppresses j", j > i) against the 64 boxes of the column tile
//                     staged in LDS.  IoU arithmetic in torchvision's op order, fp32, no contraction -> bit-exact vs the oracle.
//   nms_scan_kernel : ONE wavefront per image walks the rows in score order.  The `removed` bitmask (<= 256 words) lives in
//                     up to four registers per lane; per 64-box chunk the wave resolves intra-chunk suppression with a scalar loop
//                     over set bits (v_readlane of the diagonal words), then every lane ORs the kept rows' words of its own
//                     column(s).  Exits as soon as max_keep boxes are kept (later rows cannot change earlier decisions).
#include "common.h"

namespace {

struct NmsArgs {
    const float* boxes;   // [B][cap][4] sorted by descending score
    const int* cats;      // [B][cap]
    const int* counts;    // [B] valid boxes per image (prefix of the sorted list)
    unsigned long long* mask;  // [B][cap][W]
    int cap, W;
    float thresh;
};

__global__ __launch_bounds__(64) void nms_mask_kernel(const NmsArgs a) {
    const int b = blockIdx.z;
    const int n = min(a.counts[b], a.cap);
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (rb * 64 >= n || cb * 64 >= n) return;
    const int lane = threadIdx.x;
    const int i = rb * 64 + lane;
    unsigned long long* mrow = a.mask + ((size_t)b * a.cap + i) * a.W + cb;
    if (cb < rb) {
        if (i < n) *mrow = 0ull;
        return;
    }
    __shared__ float sb[64][4];
    __shared__ int sc[64];
    const int j0 = cb * 64;
    {
        const int j = j0 + lane;
        if (j < n) {
            const float* p = a.boxes + ((size_t)b * a.cap + j) * 4;
            sb[lane][0] = p[0]; sb[lane][1] = p[1]; sb[lane][2] = p[2]; sb[lane][3] = p[3];
            sc[lane] = a.cats[(size_t)b * a.cap + j];
        }
    }
    __syncthreads();
    if (i >= n) return;
    const float* p = a.boxes + ((size_t)b * a.cap + i) * 4;
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const int ci = a.cats[(size_t)b * a.cap + i];
    const float area_i = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    unsigned long long bits = 0ull;
    const int jn = min(64, n - j0);
    for (int jj = 0; jj < jn; ++jj) {
        const int j = j0 + jj;
        if (j <= i) continue;
        const float bx1 = sb[jj][0], by1 = sb[jj][1], bx2 = sb[jj][2], by2 = sb[jj][3];
        const float area_j = __fmul_rn(__fsub_rn(bx2, bx1), __fsub_rn(by2, by1));
        const float xx1 = fmaxf(x1, bx1), yy1 = fmaxf(y1, by1);
        const float xx2 = fminf(x2, bx2), yy2 = fminf(y2, by2);
        const float w = fmaxf(__fsub_rn(xx2, xx1), 0.f), h = fmaxf(__fsub_rn(yy2, yy1), 0.f);
        const float inter = __fmul_rn(w, h);
        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, area_j), inter));
        if (iou > a.thresh && sc[jj] == ci) bits |= (1ull << jj);
    }
    *mrow = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const NmsArgs a, int max_keep, int* keep_idx, int* keep_count) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int n = min(a.counts[b], a.cap);
    const int Wn = (n + 63) >> 6;   // words in use (<= a.W <= 256)
    const unsigned long long* M = a.mask + (size_t)b * a.cap * a.W;
    unsigned long long rem[4] = {0ull, 0ull, 0ull, 0ull};   // removed-bit words lane, lane+64, lane+128, lane+192
    int nkept = 0;
    int* out = keep_idx + (size_t)b * max_keep;
    // the diagonal word of the NEXT chunk does not depend on this chunk's decisions: keep its load in flight
    unsigned long long diag_next = (lane < n) ? M[(size_t)lane * a.W] : 0ull;
    for (int c = 0; c < Wn && nkept < max_keep; ++c) {
        const unsigned long long rsel = (c < 64) ? rem[0] : (c < 128) ? rem[1] : (c < 192) ? rem[2] : rem[3];
        const unsigned int rlo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(rsel & 0xffffffffu), c & 63);
        const unsigned int rhi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(rsel >> 32), c & 63);
        const unsigned long long removed = ((unsigned long long)rhi << 32) | rlo;
        const int row = c * 64 + lane;
        const unsigned long long diag = diag_next;
        {
            const int nrow = row + 64;
            diag_next = (c + 1 < Wn && nrow < n) ? M[(size_t)nrow * a.W + c + 1] : 0ull;
        }
        const unsigned int dlo = (unsigned int)(diag & 0xffffffffu), dhi = (unsigned int)(diag >> 32);
        const int nb = min(64, n - c * 64);
        const unsigned long long valid = (nb == 64) ? ~0ull : ((1ull << nb) - 1ull);
        unsigned long long cur = ~removed & valid;
        unsigned long long keep = 0ull;
        while (cur != 0ull) {   // wave-uniform scalar loop over alive boxes of this chunk, in score order
            const int bit = __builtin_ctzll(cur);
            keep |= (1ull << bit);
            // readlane returns a signed int: cast both halves to unsigned before widening (no sign extension)
            const unsigned int dm_hi = (unsigned int)__builtin_amdgcn_readlane((int)dhi, bit);
            const unsigned int dm_lo = (unsigned int)__builtin_amdgcn_readlane((int)dlo, bit);
            const unsigned long long dm = ((unsigned long long)dm_hi << 32) | (unsigned long long)dm_lo;
            cur &= ~dm;
            cur &= ~(1ull << bit);
        }
        // emit kept indices in order
        if ((keep >> lane) & 1ull) {
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int pos = nkept + __popcll(keep & below);
            if (pos < max_keep) out[pos] = row;
        }
        nkept += __popcll(keep);
        if (nkept >= max_keep) break;
        // OR the kept rows into the removed words of later chunks (lane owns words lane, lane + 64)
        unsigned long long kk = keep;
        const int nw = (Wn + 63) >> 6;   // 64-word groups in use (wave-uniform)
        while (kk != 0ull) {   // NU independent row loads in flight per trip (the ORs are associative): the scan is a chain of L2 round trips
            constexpr int NU = 64;      // (16: four dependent round trips per chunk; 64: one -- the wave owns 512 registers)
            int bits[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                bits[u] = kk ? __builtin_ctzll(kk) : -1;
                if (kk) kk &= kk - 1ull;
            }
#pragma unroll
            for (int wgrp = 0; wgrp < 4; ++wgrp) {
                if (wgrp >= nw) break;
                const int word = lane + 64 * wgrp;
                unsigned long long v[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) v[u] = (bits[u] >= 0 && word < Wn) ? M[(size_t)(c * 64 + bits[u]) * a.W + word] : 0ull;
                unsigned long long o = 0ull;
#pragma unroll
                for (int u = 0; u < NU; ++u) o |= v[u];
                rem[wgrp] |= o;
            }
        }
    }
    if (lane == 0) keep_count[b] = min(nkept, max_keep);
}

}  // namespace

extern "C" int amp_nms(amp_ctx* ctx, int B, int cap, const float* boxes, const int* cats, const int* counts, float thresh,
                       int max_keep, unsigned long long* mask_scratch, int* keep_idx, int* keep_count) {
    AMP_REQUIRE(ctx && boxes && cats && counts && mask_scratch && keep_idx && keep_count, "amp_nms: null argument");
    AMP_REQUIRE(B >= 1 && cap >= 1 && cap <= 16384 && max_keep >= 1, "amp_nms: cap=%d must be in [1,16384]", cap);
    NmsArgs a;
    a.boxes = boxes; a.cats = cats; a.counts = counts; a.mask = mask_scratch;
    a.cap = cap; a.W = amp::cdiv(cap, 64); a.thresh = thresh;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(a.W, a.W, B), dim3(64), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, ctx->stream, a, max_keep, keep_idx, keep_count);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}
