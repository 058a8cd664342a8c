// Backward of the grouped 3x3 convolution of a ResNeXt bottleneck (detectron2 `conv2` with RESNETS.NUM_GROUPS > 1; BASELINE configs[4]):
// weights live in the window layout of the forward kernel, [Cout][KH][KW][64] -- output channel o sees the 64 input channels of its own
// 64-wide tile, with zeros outside its group -- and so do their gradients (the arenas share offsets).
//   * data gradient: the SAME grouped forward kernel on transposed / flipped / FrozenBN-scaled windows (amp_group_dgrad_weights); a
//     stride-2 layer first spreads dy over the even positions of a zeroed map (amp_subsample2_bwd), then convolves with stride 1;
//   * weight gradient (this file): per 64-channel tile and tap a 64 x 64 product dY^T X over the pixels on the fp32 MFMA, operands
//     straight from global memory (a lane's A / B value is one float of a 128-byte channel run: coalesced), split over row slices whose
//     partial sums are added in slice order (bitwise reproducible); entries outside a channel's group are zeroed by the reduce pass, which
//     also applies the FrozenBN scale of the output channel.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GWArgs {
    const float* x;       // [B][H][W][C]
    const float* dy;      // [B][Ho][Wo][C]
    float* partial;       // [nslices][C/64][KH*KW][64 co][64 ci]
    int B, H, W, C, Ho, Wo, KH, KW, stride, pad;
    int rows_per_slice;   // output rows (b, oy) per slice
    int nslices;
    int diag_only;        // <= 32 channels per group: the two off-diagonal 32 x 32 quadrants of a tile hold no in-group pair -- two waves, not four
    int x_split;          // x rows are in the split hi|lo' row format (the native trunk of a training step): decoded on the load, hi + lo' * 2^-11 (exact)
    int dy_split;         // dy rows likewise (a 2^dy_shift riding on dy, split or fp32, is undone by the reduce pass: exact, one multiply per weight)
};

typedef _Float16 f16;
// value of channel c of a split row (per 32 channels: 64 B of hi halves, then 64 B of lo' halves)
__device__ __forceinline__ float split_at(const float* row, int c) {
    const f16* h = reinterpret_cast<const f16*>(row) + ((c >> 5) << 6) + (c & 31);
    return __fadd_rn((float)h[0], __fmul_rn((float)h[32], 1.0f / 2048.0f));
}

// grid: (C/64) * KH*KW * nslices workgroups of 256 (128 with diag_only) threads; wave w computes the 32 x 32 quadrant (co half w & 1, ci half
// w >> 1), or the diagonal quadrant w when the groups are at most 32 channels wide (the reduce pass never reads the other two)
__global__ __launch_bounds__(256) void grouped_wgrad_kernel(const GWArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntaps = a.KH * a.KW, ntiles = a.C >> 6;
    int id = blockIdx.x;
    const int slice = id % a.nslices; id /= a.nslices;
    const int tap = id % ntaps;
    const int tile = id / ntaps;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int i = lane & 31, k = lane >> 5;                        // MFMA 32x32x2: a lane holds A[i][k] and B[k][i]
    const int cob = a.diag_only ? (wave << 5) : ((wave & 1) << 5), cib = a.diag_only ? (wave << 5) : ((wave >> 1) << 5);
    const int co = (tile << 6) + cob + i;
    const int ci = (tile << 6) + cib + i;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row0 = slice * a.rows_per_slice, row1 = min(row0 + a.rows_per_slice, a.B * a.Ho);
    for (int row = row0; row < row1; ++row) {
        const int b = row / a.Ho, oy = row - b * a.Ho;
        const int iy = oy * a.stride + ky - a.pad;
        if (iy < 0 || iy >= a.H) continue;                         // the whole row of taps is padding (wave-uniform)
        const float* dyr = a.dy + ((size_t)row * a.Wo) * a.C + co;
        const float* xr = a.x + (((size_t)b * a.H + iy) * a.W) * a.C + ci;
        for (int ox0 = 0; ox0 < a.Wo; ox0 += 8) {                  // four MFMAs (8 pixels) per trip, loads first
            float av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ox = ox0 + 2 * u + k;
                const int ix = ox * a.stride + kx - a.pad;
                const bool ok = ox < a.Wo;
                const bool okx = ok && ix >= 0 && ix < a.W;
                if (a.dy_split) av[u] = ok ? split_at(a.dy + ((size_t)row * a.Wo + ox) * a.C, co) : 0.f;
                else av[u] = ok ? dyr[(size_t)ox * a.C] : 0.f;
                if (a.x_split) bv[u] = okx ? split_at(a.x + (((size_t)b * a.H + iy) * a.W + ix) * a.C, ci) : 0.f;
                else bv[u] = okx ? xr[(size_t)ix * a.C] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    // acc[r]: row (co) = (r / 4) * 8 + (lane / 32) * 4 + r % 4, column (ci) = lane % 32
    float* out = a.partial + ((((size_t)slice * ntiles + tile) * ntaps + tap) << 12);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rco = cob + (r >> 2) * 8 + k * 4 + (r & 3);
        out[(rco << 6) + cib + i] = acc[r];
    }
}

// Round 4: all nine taps in ONE workgroup.  The per-tap kernel above reads dY and the three shifted X rows once per tap -- nine workgroups fetch the
// same operands, 2.4 GB of loads per res3 layer for 9.7 GFLOP -- and was 7.4 ms of a 51-ms X-101 step.  Here a wave keeps nine 32 x 32 accumulators
// (144 registers), loads its dY values once per pixel pair and walks the 3 x 3 taps over the three input rows (the shifted X loads hit the cache: the
// same workgroup touched those lines microseconds ago): one dY load and nine X loads per nine MFMAs instead of eighteen loads, and the operands
// leave HBM about once.  Same partial layout, same slice order in the reduce pass: bitwise reproducible; the sums differ from the per-tap kernel's
// only through the slice boundaries.  3 x 3 windows only (KH = KW = 3: the ResNeXt conv2); dy may be a split row tensor (decoded on the load).
__global__ __launch_bounds__(256, 2) void grouped_wgrad9_kernel(const GWArgs a) {      // stride 1 only (the three stride-2 layers of a ResNeXt keep the per-tap kernel)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntiles = a.C >> 6;
    const int slice = blockIdx.x % a.nslices;
    const int tile = blockIdx.x / a.nslices;
    const int i = lane & 31, k = lane >> 5;
    // <= 32 channels per group: only the two diagonal quadrants matter; the four waves then work as two PAIRS that take alternate trips of the
    // slice (sub-slice 0 / 1: twice the waves per CU for the same grid; the reduce pass adds 2 x nslices partials in order)
    const int sub = a.diag_only ? (wave >> 1) : 0, nsub = a.diag_only ? 2 : 1, q = a.diag_only ? (wave & 1) : wave;
    const int cob = a.diag_only ? (q << 5) : ((q & 1) << 5), cib = a.diag_only ? (q << 5) : ((q >> 1) << 5);
    const int co = (tile << 6) + cob + i;
    const int ci = (tile << 6) + cib + i;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int row0 = slice * a.rows_per_slice, row1 = min(row0 + a.rows_per_slice, a.B * a.Ho);
    // A trip = 8 output pixels of one output row: 4 dY values and 3 x 5 input pixel pairs per lane (the ten input pixels ox0 - 1 .. ox0 + 8 under
    // the eight outputs, loaded ONCE per input row: half k of the wave holds pixel ox0 - 1 + 2 j + k; the three kx taps are views of them --
    // kx = 0 takes pair u as it is, kx = 2 pair u + 1, kx = 1 the two halves crossed, one cross-half shuffle per pair).  The loads of trip
    // t + 1 are issued before the 36 MFMAs of trip t (two register sets): with two or three waves per SIMD nothing else hides their latency.
    const int tpr = (a.Wo + 7) >> 3, ntrips = (row1 - row0) * tpr;
    auto load = [&](int t, float (&av)[4], float (&P)[3][5]) {
        const int row = row0 + t / tpr, ox0 = (t % tpr) << 3;
        const int b = row / a.Ho, oy = row - b * a.Ho;
        const float* dyr = a.dy + ((size_t)row * a.Wo) * a.C;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ox = ox0 + 2 * u + k;
            av[u] = (ox < a.Wo) ? (a.dy_split ? split_at(dyr + (size_t)ox * a.C, co) : dyr[(size_t)ox * a.C + co]) : 0.f;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + ky - 1;
            const bool rv = iy >= 0 && iy < a.H;
            const float* xr = a.x + (((size_t)b * a.H + (rv ? iy : 0)) * a.W) * a.C + ci;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int px = ox0 - 1 + 2 * j + k;
                P[ky][j] = (rv && px >= 0 && px < a.W) ? xr[(size_t)px * a.C] : 0.f;
            }
        }
    };
    auto compute = [&](const float (&av)[4], const float (&P)[3][5]) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float S[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) S[j] = __shfl_xor(P[ky][j], 32);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float m1 = k ? S[u + 1] : S[u];
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], P[ky][u], acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], m1, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], P[ky][u + 1], acc[ky * 3 + 2], 0, 0, 0);
            }
        }
    };
    float av0[4], av1[4], P0[3][5], P1[3][5];
    if (sub < ntrips) load(sub, av0, P0);
    for (int t = sub; t < ntrips; t += 2 * nsub) {
        if (t + nsub < ntrips) load(t + nsub, av1, P1);
        compute(av0, P0);
        if (t + nsub < ntrips) {
            if (t + 2 * nsub < ntrips) load(t + 2 * nsub, av0, P0);
            compute(av1, P1);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* out = a.partial + ((((size_t)(slice * nsub + sub) * ntiles + tile) * 9 + t) << 12);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rco = cob + (r >> 2) * 8 + k * 4 + (r & 3);
            out[(rco << 6) + cib + i] = acc[t][r];
        }
    }
}

// grad[co][ky][kx][slot] = scale[co] * sum over slices (in slice order) of the partials, zero outside the group of co
__global__ void grouped_wgrad_reduce_kernel(const float* partial, const float* scale, float* grad, int C, int ntaps, int nslices, int cpg, float out_scale) {
    const size_t total = (size_t)C * ntaps * 64;
    const int ntiles = C >> 6;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(idx & 63);
        const int tap = (int)((idx >> 6) % ntaps);
        const int co = (int)(idx / ((size_t)ntaps * 64));
        const int tile = co >> 6, cl = co & 63;
        float s = 0.f;
        if (cl / cpg == slot / cpg) {
            for (int sl = 0; sl < nslices; ++sl) s = __fadd_rn(s, partial[((((size_t)sl * ntiles + tile) * ntaps + tap) << 12) + (cl << 6) + slot]);
            s = __fmul_rn(s, out_scale);                 // 2^-dy_shift: exact
            if (scale) s = __fmul_rn(s, scale[co]);
        }
        grad[idx] = s;
    }
}

// wt[ci][ky][kx][b] = w[tile*64 + b][KH-1-ky][KW-1-kx][ci & 63] * scale[tile*64 + b]   (ci and tile*64 + b share the 64-wide tile)
__global__ void group_dgrad_weights_kernel(const float* w, const float* scale, float* wt, int C, int KH, int KW) {
    const size_t total = (size_t)C * KH * KW * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx & 63);
        size_t t = idx >> 6;
        const int kx = (int)(t % KW); t /= KW;
        const int ky = (int)(t % KH);
        const int ci = (int)(t / KH);
        const int co = (ci & ~63) + b;
        float v = w[(((size_t)co * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)) * 64 + (ci & 63)];
        if (scale) v = __fmul_rn(v, scale[co]);
        wt[idx] = v;
    }
}

}  // namespace

extern "C" {

size_t amp_grouped_wgrad_scratch_floats(const amp_conv_desc* d) {
    if (!d || d->Cin <= 0) return 0;
    const size_t per_slice = (size_t)d->Cin * d->KH * d->KW * 64;
    return std::min<size_t>(64, std::max<size_t>(16, ((size_t)48 << 20) / per_slice)) * per_slice;          // 16 .. 64 slices, at most 48 Mi floats beyond 16
}

int amp_conv2d_grouped_wgrad(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* dy, const float* scale,
                             float* scratch, float* grad_win) {
    return amp_conv2d_grouped_wgrad_fmt(ctx, d, groups, x, dy, scale, scratch, grad_win, 0, 0);
}

int amp_conv2d_grouped_wgrad_fmt(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* dy, const float* scale,
                                 float* scratch, float* grad_win, int fmt, int dy_shift) {
    AMP_REQUIRE(ctx && d && x && dy && scratch && grad_win && groups > 1 && fmt >= 0 && fmt <= 3 && dy_shift >= 0 && dy_shift <= 24, "amp_conv2d_grouped_wgrad: bad argument");
    AMP_REQUIRE(d->Cin == d->Cout && d->Cin % 64 == 0 && d->Cin % groups == 0, "amp_conv2d_grouped_wgrad: needs Cin == Cout, a multiple of 64 and of groups");
    const int cpg = d->Cin / groups;
    AMP_REQUIRE(cpg == 8 || cpg == 16 || cpg == 32 || cpg == 64, "amp_conv2d_grouped_wgrad: %d channels per group (8/16/32/64 supported)", cpg);
    GWArgs a;
    a.x = x; a.dy = dy; a.partial = scratch;
    a.x_split = fmt & 1; a.dy_split = (fmt >> 1) & 1;
    a.B = d->B; a.H = d->H; a.W = d->W; a.C = d->Cin; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    a.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    AMP_REQUIRE(a.Ho > 0 && a.Wo > 0, "amp_conv2d_grouped_wgrad: empty output");
    const int rows = a.B * a.Ho, pairs = (a.C >> 6) * a.KH * a.KW;
    a.diag_only = cpg <= 32 ? 1 : 0;
    static const bool v1 = getenv("AMP_GROUPED_WGRAD_V1") != nullptr;      // EXPERIMENT switch: the per-tap kernel of round 3
    if (!v1 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.x_split) {      // (a split x is decoded by the per-tap kernel only: the model decodes it in one pass instead)
        // nine taps per workgroup: (tile, slice) workgroups -- enough slices to fill the chip (~3 workgroups per CU), within the scratch
        const int ntiles = a.C >> 6;
        const int cap = (int)(amp_grouped_wgrad_scratch_floats(d) / ((size_t)a.C * 9 * 64)) / (a.diag_only ? 2 : 1);
        int nslices = std::max(1, std::min({cap, rows, (768 + ntiles - 1) / ntiles}));
        a.rows_per_slice = (rows + nslices - 1) / nslices;
        nslices = (rows + a.rows_per_slice - 1) / a.rows_per_slice;
        a.nslices = nslices;
        hipLaunchKernelGGL(grouped_wgrad9_kernel, dim3(ntiles * nslices), dim3(256), 0, ctx->stream, a);
        if (a.diag_only) a.nslices = 2 * nslices;          // two sub-slices per workgroup (the reduce pass below)
    } else {
        const int target = (cpg <= 32) ? 3072 : 1024;                                        // ~four (two-wave: twelve) workgroups per CU
        int nslices = std::max(1, std::min({16, rows, (target + pairs - 1) / pairs}));
        a.rows_per_slice = (rows + nslices - 1) / nslices;
        nslices = (rows + a.rows_per_slice - 1) / a.rows_per_slice;
        a.nslices = nslices;
        hipLaunchKernelGGL(grouped_wgrad_kernel, dim3(pairs * nslices), dim3(a.diag_only ? 128 : 256), 0, ctx->stream, a);
    }
    const int nslices = a.nslices;
    const size_t total = (size_t)a.C * a.KH * a.KW * 64;
    hipLaunchKernelGGL(grouped_wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, ctx->stream,
                       scratch, scale, grad_win, a.C, a.KH * a.KW, nslices, cpg, ldexpf(1.0f, -dy_shift));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_group_dgrad_weights(amp_ctx* ctx, const float* w_win, const float* scale, int C, int KH, int KW, float* wt_win) {
    AMP_REQUIRE(ctx && w_win && wt_win && C > 0 && C % 64 == 0 && KH > 0 && KW > 0, "amp_group_dgrad_weights: bad argument");
    const size_t total = (size_t)C * KH * KW * 64;
    hipLaunchKernelGGL(group_dgrad_weights_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, ctx->stream, w_win, scale, wt_win, C, KH, KW);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
