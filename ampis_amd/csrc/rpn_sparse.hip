// Sparse backward of the RPN head (round 4).
//
// detectron2's RPN losses (modeling/proposal_generator/rpn.py losses(): objectness BCE over the <= 256 sampled anchors of an image,
// smooth-L1 over the sampled positives) leave a gradient in <= BATCH_SIZE_PER_IMAGE entries of the predictor maps per image and exact zeros in the
// other ~800 000: the gradient of the head's hidden layer (3x3 conv + ReLU) is non-zero in at most 256 PIXELS of an image's five maps.  The dense
// backward pass ran the head's four gradient GEMMs over every pixel all the same -- at B = 16 / 1024^2 the 3x3 conv's weight gradient (2.7 ms at p2)
// and data gradient (2.9 ms at p2), the predictor's data-gradient pass and weight gradient: ~9 ms of a 74-ms step spent on products with an exact zero.
// Here the sampled pixels are compacted (<= B x 256 rows, all levels in ONE list: the head's weights are shared) and the same gradients are
//   d_t[r]   = relu'(t[p_r]) * (d_pred[p_r] . W_pred)                               one row of 256 per sampled pixel
//   dW_pred  = d_pred^T . t,   db_pred = colsum(d_pred),   db_conv = colsum(d_t)    sums over the rows, ascending
//   dW_conv  = d_t^T . X      with X[r] = the 3x3 patch of the FPN feature under p_r (zero outside the map): a [256 x R] . [R x 2304] GEMM
//   d_feat  += scatter of G = d_t . W_conv^T ([R x 256] . [256 x 2304]): row r, tap (ky, kx) goes to pixel p_r + (ky - 1, kx - 1)
// -- the two GEMMs on the library's own kernels (amp_conv2d_wgrad_scaled / conv_run as 1x1 layers over a 1 x R "image"), everything else a handful of
// small kernels.  Every omitted term of the dense sums is an exact zero, so the results are the dense results up to the order of the fp32 additions
// (held to the dense path and to torch autograd in tests/test_backward_gpu.py, tests/test_train_bwd_gpu.py); the order here is fixed (rows sorted by
// (image, level, pixel); the scatter runs tap by tap, and within one tap distinct rows of a map hit distinct pixels): bitwise reproducible, no atomics.
#include "common.h"

#include <hip/hip_runtime.h>

#include <cstdio>

namespace {

constexpr int NLV = 5;
constexpr unsigned int NOROW = 0xffffffffu;

struct SparseGeom {
    int hw[NLV], fh[NLV], fw[NLV], off[NLV + 1];      // off: first anchor index of a level (3 anchors per pixel)
};

// one workgroup per image: the sampled anchors' pixels, sorted by (level, pixel), duplicates removed -> rows[b * batch + i], NOROW beyond the count
__global__ __launch_bounds__(512) void rpn_nz_rows_kernel(const SparseGeom g, const int* __restrict__ sampled, const int* __restrict__ counts, int batch,
                                                          unsigned int* __restrict__ rows, int* __restrict__ nrows) {
    __shared__ unsigned int keys[512];
    __shared__ int scan[512];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts[2 * b] + counts[2 * b + 1], batch);
    unsigned int key = NOROW;
    if (tid < n) {
        const int an = sampled[(size_t)b * batch + tid];
        if (an >= 0 && an < g.off[NLV]) {
            int lvl = 0;
            while (lvl + 1 < NLV && an >= g.off[lvl + 1]) ++lvl;
            key = ((unsigned int)lvl << 26) | (unsigned int)((an - g.off[lvl]) / 3);
        }
    }
    keys[tid] = key;
    __syncthreads();
    for (int k = 2; k <= 512; k <<= 1)                  // bitonic sort, ascending (NOROW last)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = tid ^ j;
            if (p > tid) {
                const unsigned int x = keys[tid], y = keys[p];
                const bool up = (tid & k) == 0;
                if ((x > y) == up) { keys[tid] = y; keys[p] = x; }
            }
            __syncthreads();
        }
    const unsigned int mine = keys[tid];
    const int flag = (mine != NOROW && (tid == 0 || keys[tid - 1] != mine)) ? 1 : 0;
    scan[tid] = flag;
    __syncthreads();
    for (int d = 1; d < 512; d <<= 1) {                 // inclusive scan
        const int v = tid >= d ? scan[tid - d] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const int total = scan[511];
    if (tid < batch) rows[(size_t)b * batch + tid] = NOROW;
    __syncthreads();
    if (flag) rows[(size_t)b * batch + scan[tid] - 1] = mine;
    if (tid == 0) nrows[b] = total;
}

struct RowArgs {
    SparseGeom g;
    const unsigned int* rows;
    int batch, ld, K;                          // predictor rows of ld floats, K of them real
    const float* dpred[NLV];                   // [B * hw][ld]
    const float* t[NLV];                       // hidden activation (after ReLU) [B * hw][256], split rows or fp32
    int t_split;
    const float* t_rows;                       // the rows' hidden activations recomputed ([R][256] fp32) or null: read from t
    const float* w_pred;                       // [K][256]
    float* dpred_rows;                         // [R][16]
    float* act_rows;                           // [R][256]
    float* dt_rows;                            // [R][256]
};

__device__ __forceinline__ float load_ch(const float* base, size_t row, int c, int split) {      // channel c of a 256-channel row
    if (!split) return base[row * 256 + c];
    const char* p = reinterpret_cast<const char*>(base + row * 256) + (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2;
    const _Float16 h = *reinterpret_cast<const _Float16*>(p), l = *reinterpret_cast<const _Float16*>(p + 64);
    return __fadd_rn((float)h, __fmul_rn((float)l, 1.0f / 2048.0f));
}

// one workgroup per row, one thread per hidden channel: d_t = (t > 0) ? sum_k d_pred[k] * W_pred[k][c] : 0 (the sums of small_k_dgrad_kernel, in its order)
__global__ __launch_bounds__(256) void rpn_dt_rows_kernel(const RowArgs a) {
    const int r = blockIdx.x, c = threadIdx.x;
    const unsigned int key = a.rows[r];
    float dt = 0.f, act = 0.f, dp = 0.f;
    if (key != NOROW) {
        const int b = r / a.batch, lvl = (int)(key >> 26);
        const size_t row = (size_t)b * a.g.hw[lvl] + (key & 0x3ffffffu);
        const float* dl = a.dpred[lvl] + row * a.ld;
        act = a.t_rows ? a.t_rows[(size_t)r * 256 + c] : load_ch(a.t[lvl], row, c, a.t_split);
        float v = 0.f;
        for (int k = 0; k < a.K; ++k) v = __fadd_rn(v, __fmul_rn(dl[k], a.w_pred[(size_t)k * 256 + c]));
        dt = act > 0.f ? v : 0.f;
        if (c < a.K) dp = dl[c];
    }
    a.dt_rows[(size_t)r * 256 + c] = dt;
    a.act_rows[(size_t)r * 256 + c] = act;
    if (c < 16) a.dpred_rows[(size_t)r * 16 + c] = dp;
}

// the predictor's gradients and the conv's bias gradient: sums over the rows in ascending order -- 32 row slices in parallel (8 workgroups x 4 quarter
// slices each), added in slice order by rpn_head_sums_final_kernel.  blockIdx.x 0 .. 15: dW_pred[k][c]; 16: db_conv[c]; 17: db_pred[k]
constexpr int SUM_SLICES = 8;
__global__ __launch_bounds__(1024) void rpn_head_sums_kernel(const float* __restrict__ dpred_rows, const float* __restrict__ act_rows, const float* __restrict__ dt_rows,
                                                             int R, int K, float* __restrict__ partial) {      // partial [18][SUM_SLICES * 4][256]
    const int c = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int blk = blockIdx.x, s = blockIdx.y * 4 + q, ns = SUM_SLICES * 4;
    const int r0 = (int)((long long)R * s / ns), r1 = (int)((long long)R * (s + 1) / ns);
    float acc = 0.f;
    if (blk < 16) {
        if (blk < K) for (int r = r0; r < r1; ++r) acc = __fadd_rn(acc, __fmul_rn(dpred_rows[(size_t)r * 16 + blk], act_rows[(size_t)r * 256 + c]));
    } else if (blk == 16) {
        for (int r = r0; r < r1; ++r) acc = __fadd_rn(acc, dt_rows[(size_t)r * 256 + c]);
    } else {
        if (c < K) for (int r = r0; r < r1; ++r) acc = __fadd_rn(acc, dpred_rows[(size_t)r * 16 + c]);
    }
    partial[((size_t)blk * ns + s) * 256 + c] = acc;
}
__global__ __launch_bounds__(256) void rpn_head_sums_final_kernel(const float* __restrict__ partial, int K, float* __restrict__ gw_pred, float* __restrict__ gb_pred,
                                                                   float* __restrict__ gb_conv) {
    const int c = threadIdx.x, blk = blockIdx.x, ns = SUM_SLICES * 4;
    float t = partial[((size_t)blk * ns) * 256 + c];
    for (int s = 1; s < ns; ++s) t = __fadd_rn(t, partial[((size_t)blk * ns + s) * 256 + c]);
    if (blk < 16) { if (blk < K) gw_pred[(size_t)blk * 256 + c] = t; }
    else if (blk == 16) gb_conv[c] = t;
    else if (c < K) gb_pred[c] = t;
}

struct GatherArgs {
    SparseGeom g;
    const unsigned int* rows;
    int batch;
    const float* feat[NLV];                    // FPN features [B * hw][256], split rows or fp32
    int feat_split;
    float* xg;                                 // [R][9][256]
};

// X[r][tap][c] = feature at p_r + (ky - 1, kx - 1), 0 outside the map
__global__ __launch_bounds__(256) void rpn_gather_x_kernel(const GatherArgs a) {
    const int r = blockIdx.x, tap = blockIdx.y, c = threadIdx.x;
    const unsigned int key = a.rows[r];
    float v = 0.f;
    if (key != NOROW) {
        const int b = r / a.batch, lvl = (int)(key >> 26), pix = (int)(key & 0x3ffffffu);
        const int W = a.g.fw[lvl], H = a.g.fh[lvl];
        const int y = pix / W + tap / 3 - 1, x = pix % W + tap % 3 - 1;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = load_ch(a.feat[lvl], (size_t)b * a.g.hw[lvl] + (size_t)y * W + x, c, a.feat_split);
    }
    a.xg[((size_t)r * 9 + tap) * 256 + c] = v;
}

// Wt[(tap, c)][n] = scale[n] * W[n][tap][c]: the conv's weights as the [2304 x 256] matrix of the data-gradient GEMM
__global__ __launch_bounds__(256) void rpn_wt_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ wt, int Cout, int Cin) {
    const int j = blockIdx.x, n = threadIdx.x;                     // j = tap * Cin + c
    for (int nn = n; nn < Cout; nn += 256) {
        const float v = w[(size_t)nn * 9 * Cin + j];
        wt[(size_t)j * Cout + nn] = scale ? __fmul_rn(v, scale[nn]) : v;
    }
}

struct ScatterArgs {
    SparseGeom g;
    const unsigned int* rows;
    int batch, tap;
    const float* G;                            // [R][9][256]
    float* dfeat[NLV];                         // [B * hw][256] fp32
};

// one tap: d_feat[p_r + (ky - 1, kx - 1)] += G[r][tap]; distinct rows of a map are distinct pixels, so no two rows meet in one tap
__global__ __launch_bounds__(256) void rpn_scatter_tap_kernel(const ScatterArgs a) {
    const int r = blockIdx.x, c = threadIdx.x;
    const unsigned int key = a.rows[r];
    if (key == NOROW) return;
    const int b = r / a.batch, lvl = (int)(key >> 26), pix = (int)(key & 0x3ffffffu);
    const int W = a.g.fw[lvl], H = a.g.fh[lvl];
    const int y = pix / W + a.tap / 3 - 1, x = pix % W + a.tap % 3 - 1;
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return;
    float* p = a.dfeat[lvl] + ((size_t)b * a.g.hw[lvl] + (size_t)y * W + x) * 256 + c;
    *p = __fadd_rn(*p, a.G[((size_t)r * 9 + a.tap) * 256 + c]);
}

}  // namespace

namespace amp {

int rpn_sparse_backward(amp_ctx* ctx, const RpnSparseArgs& A) {
    AMP_REQUIRE(A.B >= 1 && A.batch >= 1 && A.batch <= 512 && A.K >= 1 && A.K <= 16 && A.ld == 16, "rpn_sparse_backward: batch <= 512, K <= 16, ld == 16");
    SparseGeom g;
    g.off[0] = 0;
    for (int l = 0; l < NLV; ++l) {
        g.fh[l] = A.fh[l]; g.fw[l] = A.fw[l]; g.hw[l] = A.fh[l] * A.fw[l];
        g.off[l + 1] = g.off[l] + g.hw[l] * 3;
        AMP_REQUIRE(g.hw[l] < (1 << 26), "rpn_sparse_backward: a level of %d pixels", g.hw[l]);
    }
    const int R = A.B * A.batch;
    hipLaunchKernelGGL(rpn_nz_rows_kernel, dim3(A.B), dim3(512), 0, ctx->stream, g, A.sampled, A.counts, A.batch, A.rows, A.nrows);
    GatherArgs ga;
    ga.g = g; ga.rows = A.rows; ga.batch = A.batch; ga.feat_split = A.feat_split; ga.xg = A.xg;
    for (int l = 0; l < NLV; ++l) ga.feat[l] = A.feat[l];
    hipLaunchKernelGGL(rpn_gather_x_kernel, dim3(R, 9), dim3(256), 0, ctx->stream, ga);
    AMP_HIP_CHECK(hipGetLastError());
    if (A.recompute_t) {
        // the rows' hidden activations: relu(patch . W_conv + shift) -- the 3x3 conv's own products in its own K order ((tap, channel): the patch row IS the
        // conv's im2col row, decoded from the split feature it was computed from), as a 1x1 layer 2304 -> 256 over the 1 x R image; G holds them until the
        // row kernel has copied them into act_rows
        amp_conv_desc dt;
        dt.B = 1; dt.H = 1; dt.W = R; dt.Cin = 9 * A.C; dt.Cout = A.C; dt.KH = 1; dt.KW = 1; dt.stride = 1; dt.pad = 0; dt.relu = 1; dt.res_mode = 0; dt.out_mode = 0;
        AMP_TRY_STATUS(amp::conv_run(ctx, &dt, 1, A.xg, A.w_conv, A.w_conv_split, 0, A.conv_scale, A.conv_shift, nullptr, nullptr, A.G, 0, 0));
    }
    RowArgs ra;
    ra.g = g; ra.rows = A.rows; ra.batch = A.batch; ra.ld = A.ld; ra.K = A.K; ra.t_split = A.t_split; ra.w_pred = A.w_pred;
    ra.t_rows = A.recompute_t ? A.G : nullptr;
    ra.dpred_rows = A.dpred_rows; ra.act_rows = A.act_rows; ra.dt_rows = A.dt_rows;
    for (int l = 0; l < NLV; ++l) { ra.dpred[l] = A.dpred[l]; ra.t[l] = A.t[l]; }
    hipLaunchKernelGGL(rpn_dt_rows_kernel, dim3(R), dim3(256), 0, ctx->stream, ra);
    // (the partial sums live in G, which the data-gradient GEMM overwrites further down: 18 x 32 x 256 floats)
    hipLaunchKernelGGL(rpn_head_sums_kernel, dim3(18, SUM_SLICES), dim3(1024), 0, ctx->stream, A.dpred_rows, A.act_rows, A.dt_rows, R, A.K, A.G);
    hipLaunchKernelGGL(rpn_head_sums_final_kernel, dim3(18), dim3(256), 0, ctx->stream, A.G, A.K, A.gw_pred, A.gb_pred, A.gb_conv);
    // dW_conv[n][tap][c] = sum_r d_t[r][n] * X[r][tap][c]: the weight gradient of a 1x1 layer 2304 -> 256 over a 1 x R image (gradients of 1e-9 .. 1e-4: * 2^16 in front of the f16 split)
    amp_conv_desc dw;
    dw.B = 1; dw.H = 1; dw.W = R; dw.Cin = 9 * A.C; dw.Cout = A.C; dw.KH = 1; dw.KW = 1; dw.stride = 1; dw.pad = 0; dw.relu = 0; dw.res_mode = 0; dw.out_mode = 0;
    AMP_REQUIRE(amp_conv_wgrad_scratch_floats(&dw) <= A.wg_scratch_floats, "rpn_sparse_backward: wgrad scratch too small");
    AMP_TRY_STATUS(amp_conv2d_wgrad_scaled(ctx, &dw, A.xg, A.dt_rows, A.conv_scale, A.wg_scratch, A.gw_conv, 0, 16, 0));
    // G = d_t . Wt: a 1x1 layer 256 -> 2304 over the same image
    hipLaunchKernelGGL(rpn_wt_kernel, dim3(9 * A.C), dim3(256), 0, ctx->stream, A.w_conv, A.conv_scale, A.wt, A.C, A.C);
    amp_conv_desc dg;
    dg.B = 1; dg.H = 1; dg.W = R; dg.Cin = A.C; dg.Cout = 9 * A.C; dg.KH = 1; dg.KW = 1; dg.stride = 1; dg.pad = 0; dg.relu = 0; dg.res_mode = 0; dg.out_mode = 0;
    AMP_TRY_STATUS(amp::conv_run(ctx, &dg, 1, A.dt_rows, A.wt, nullptr, 0, nullptr, nullptr, nullptr, nullptr, A.G, 16, 0));
    ScatterArgs sa;
    sa.g = g; sa.rows = A.rows; sa.batch = A.batch; sa.G = A.G;
    for (int l = 0; l < NLV; ++l) sa.dfeat[l] = A.dfeat[l];
    for (int tap = 0; tap < 9; ++tap) {
        sa.tap = tap;
        hipLaunchKernelGGL(rpn_scatter_tap_kernel, dim3(R), dim3(256), 0, ctx->stream, sa);
    }
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // namespace amp
