// Weight gradient of a convolution on the fp32 MFMA (replaces cuDNN wgrad under DefaultTrainer.run_step):
//   dW[n][ky][kx][c] = sum_m dY[m][n] * X[pixel(m) shifted by (ky,kx)][c]           (m over B*Ho*Wo output pixels)
// GEMM view: [N x M] * [M x K'], K' = KH*KW*Cin, reduction over the pixels.  Both operands sit in memory with the reduction
// index m as the slow one (NHWC rows), and that is also how the f32 MFMA wants them in LDS: lane (i, kh) of
// v_mfma_f32_32x32x2_f32 reads A[i][k] / B[k][j], i.e. for a fixed k 32 CONSECUTIVE n (or c) -- a conflict-free ds_read_b32 from
// a tile stored [m][n].  So the LDS tiles are plain copies of 32 rows x 128 columns of dY and of the shifted X rows, filled by
// LDS-DMA (2 rows of 512 B per wave-instruction), no transpose anywhere.
// Work split: output tile 128(n) x 128(c') per workgroup (a c' tile lies inside one tap because Cin % 128 == 0), the pixel
// range is cut into S slices (split-K) so that the grid fills the chip; every slice writes its partial tile to a scratch slab
// [S][N][K'] and wgrad_reduce_kernel adds the slabs in fixed order (bitwise reproducible, no float atomics), applies the
// FrozenBN scale of the output channel and stores or accumulates into the gradient.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr unsigned int OOB = 0x80000000u;
constexpr int BKW = 32;     // pixels per step
constexpr int TN = 128;     // output-channel tile
constexpr int TC = 128;     // (tap, c) tile

struct WgradArgs {
    const float* dy;        // [M][N]
    const float* x;         // [B][H][W][Cin]
    float* partial;         // [S][N][Kp]
    int B, H, W, Cin, Ho, Wo, N;
    int KH, KW, stride, pad;
    int M, Kp;              // Kp = KH*KW*Cin
    int rows_per_split;     // multiple of 32
    int ntn, ntc, nsplit;
    unsigned int dy_bytes, x_bytes;
};

__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BKW * 128];   // 2 buffers x (P 32x128 + Q 32x128)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave tile: 64 (n) x 64 (c')
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    const int tile_c = bid % a.ntc; bid /= a.ntc;
    const int tile_n = bid % a.ntn; bid /= a.ntn;
    const int split = bid;
    const int n0 = tile_n * TN;
    const int kp0 = tile_c * TC;
    const int tap = kp0 / a.Cin;
    const int c0 = kp0 - tap * a.Cin;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BKW - 1) / BKW : 0;

    const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

    // staging: DMA instruction g (0..3) of this wave fills tile rows 8*wave + 2g + (lane>>5), 16-B chunk lane&31
    const int chunk = lane & 31;
    const int HoWo = a.Ho * a.Wo;
    int r_b[4], r_oy[4], r_ox[4];          // pixel of this lane's 4 rows (advanced by 32 pixels per step)
    unsigned int p_voff[4];                // dY row byte offset (+ n0 + chunk), OOB when n is out of range
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 8 * wave + 2 * g + lh;
        const int m = m_begin + r;
        r_b[g] = m / HoWo;
        const int rem = m - r_b[g] * HoWo;
        r_oy[g] = rem / a.Wo;
        r_ox[g] = rem - r_oy[g] * a.Wo;
        const int n = n0 + 4 * chunk;
        p_voff[g] = (n < a.N) ? (unsigned int)(((size_t)m * a.N + n) * 4) : OOB;   // N % 4 == 0: a chunk is all-in or all-out
    }
    const unsigned int p_step = (unsigned int)(BKW * a.N * 4);
    int staged = 0;                        // steps staged so far

    auto stage = [&](int buf) {
        float* P = lds + buf * (2 * BKW * 128);
        float* Q = P + BKW * 128;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 8 * wave + 2 * g + lh;
            const int m = m_begin + staged * BKW + r;
            // dY rows: m beyond the slice / tensor contribute nothing
            const unsigned int pv = (m < m_end && p_voff[g] != OOB) ? p_voff[g] + (unsigned int)staged * p_step : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (__attribute__((address_space(3))) void*)(P + (8 * wave + 2 * g) * 128), 16,
                                                     (int)pv, 0, 0, 0);
            const int iy = r_oy[g] * a.stride - a.pad + ky, ix = r_ox[g] * a.stride - a.pad + kx;
            const bool v = m < m_end && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const unsigned int qv = v ? (unsigned int)((((size_t)(r_b[g] * a.H + iy) * a.W + ix) * a.Cin + c0 + 4 * chunk) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(Q + (8 * wave + 2 * g) * 128), 16,
                                                     (int)qv, 0, 0, 0);
            // advance this row by 32 pixels
            r_ox[g] += BKW;
            while (r_ox[g] >= a.Wo) { r_ox[g] -= a.Wo; if (++r_oy[g] == a.Ho) { r_oy[g] = 0; ++r_b[g]; } }
        }
        ++staged;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) {
        stage(0);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            if (step + 1 < nsteps) stage(cur ^ 1);
            const float* P = lds + cur * (2 * BKW * 128) + wm * 64 + l31;
            const float* Q = lds + cur * (2 * BKW * 128) + BKW * 128 + wn * 64 + l31;
#pragma unroll
            for (int t = 0; t < BKW / 2; ++t) {
                const int k = 2 * t + lh;
                const float a0 = P[k * 128], a1 = P[k * 128 + 32];
                const float b0 = Q[k * 128], b1 = Q[k * 128 + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    // partial tile -> slab [split][n][k']
    float* out = a.partial + (size_t)split * a.N * a.Kp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = kp0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < a.N && kp < a.Kp) out[(size_t)n * a.Kp + kp] = acc[i][j][e];
            }
        }
}

// grad[n][k'] (= or +=) scale[n] * sum_s partial[s][n][k']   (fixed order -> reproducible)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t nk, int Kp, const float* __restrict__ scale,
                                    float* __restrict__ grad, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s = __fadd_rn(s, partial[(size_t)k * nk + i]);
        if (scale) s = __fmul_rn(s, scale[i / Kp]);
        grad[i] = accumulate ? __fadd_rn(grad[i], s) : s;
    }
}

// column sums of dY [M][N] (bias gradients): two passes, fixed order.  Pass 1: a workgroup owns COLSUM_ROWS rows; a thread owns one
// 16-byte column group and every (256 / N4)-th row, so a wave reads whole 1 KiB row segments; row lanes are combined through LDS.
constexpr int COLSUM_ROWS = 512;
typedef float cs_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, int M, int N4, float* __restrict__ partial) {
    __shared__ cs_f4 red[256];
    const int tid = threadIdx.x;
    const int cols = min(N4, 256);               // column groups handled per sweep
    const int nrl = 256 / cols;                  // row lanes
    const int c4 = tid % cols, rl = tid / cols;
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
    const cs_f4* src = reinterpret_cast<const cs_f4*>(dy);
    for (int cbase = 0; cbase < N4; cbase += cols) {
        const int c = cbase + c4;
        cs_f4 acc = {0.f, 0.f, 0.f, 0.f};
        if (rl < nrl && c < N4)
            for (int r = r0 + rl; r < r1; r += nrl) {
                const cs_f4 v = src[(size_t)r * N4 + c];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __fadd_rn(acc[q], v[q]);
            }
        __syncthreads();
        red[tid] = acc;
        __syncthreads();
        if (rl == 0 && c < N4) {
            cs_f4 s = red[c4];
            for (int k = 1; k < nrl; ++k) {
                const cs_f4 v = red[k * cols + c4];
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = __fadd_rn(s[q], v[q]);
            }
            reinterpret_cast<cs_f4*>(partial)[(size_t)blockIdx.x * N4 + c] = s;
        }
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int nparts, int N, float* __restrict__ out, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s = __fadd_rn(s, partial[(size_t)p * N + n]);
    out[n] = accumulate ? __fadd_rn(out[n], s) : s;
}

// dgrad weights: Wt[c][KH-1-ky][KW-1-kx][n] = W[n][ky][kx][c] * scale[n]   (scale may be null)
__global__ void dgrad_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, int N, int KH, int KW, int C,
                                    float* __restrict__ wt) {
    const size_t total = (size_t)N * KH * KW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes wt: [c][ky'][kx'][n]
        const int n = (int)(i % N);
        size_t t = i / N;
        const int kxp = (int)(t % KW); t /= KW;
        const int kyp = (int)(t % KH);
        const int c = (int)(t / KH);
        const int ky = KH - 1 - kyp, kx = KW - 1 - kxp;
        float v = w[(((size_t)n * KH + ky) * KW + kx) * C + c];
        if (scale) v = __fmul_rn(v, scale[n]);
        wt[i] = v;
    }
}

}  // namespace

extern "C" {

/* dW = conv-wgrad(dy, x). scratch: >= nsplit*N*Kp floats (see amp_conv_wgrad_scratch_floats). grad is [N][KH][KW][Cin]. */
size_t amp_conv_wgrad_scratch_floats(const amp_conv_desc* d) {
    if (!d) return 0;
    const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    const long long M = (long long)d->B * Ho * Wo;
    const int Kp = d->KH * d->KW * d->Cin;
    const int tiles = amp::cdiv(d->Cout, TN) * amp::cdiv(Kp, TC);
    int nsplit = (int)std::max(1LL, std::min((long long)amp::cdiv(2048, tiles), (M + 1023) / 1024));
    return (size_t)nsplit * d->Cout * Kp;
}

int amp_conv2d_wgrad(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                     float* grad, int accumulate) {
    AMP_REQUIRE(ctx && d && x && dy && scratch && grad, "amp_conv2d_wgrad: null argument");
    AMP_REQUIRE(d->Cin % TC == 0, "amp_conv2d_wgrad: Cin=%d must be a multiple of %d", d->Cin, TC);
    AMP_REQUIRE(d->Cout % 4 == 0, "amp_conv2d_wgrad: Cout=%d must be a multiple of 4", d->Cout);
    WgradArgs a;
    a.dy = dy; a.x = x; a.partial = scratch;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.N = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    a.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    const long long M = (long long)a.B * a.Ho * a.Wo;
    a.M = (int)M;
    a.Kp = a.KH * a.KW * a.Cin;
    a.ntn = amp::cdiv(a.N, TN);
    a.ntc = amp::cdiv(a.Kp, TC);
    const int tiles = a.ntn * a.ntc;
    a.nsplit = (int)std::max(1LL, std::min((long long)amp::cdiv(2048, tiles), (M + 1023) / 1024));
    a.rows_per_split = amp::cdiv(amp::cdiv(a.M, a.nsplit), BKW) * BKW;
    const size_t dyb = (size_t)M * a.N * 4, xb = (size_t)a.B * a.H * a.W * a.Cin * 4;
    AMP_REQUIRE(dyb < (size_t)OOB && xb < (size_t)OOB, "amp_conv2d_wgrad: operand larger than 2 GiB (split the batch)");
    a.dy_bytes = (unsigned int)dyb; a.x_bytes = (unsigned int)xb;
    hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(tiles * a.nsplit), dim3(256), 0, ctx->stream, a);
    const size_t nk = (size_t)a.N * a.Kp;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((nk + 255) / 256, 4096)), dim3(256), 0, ctx->stream, scratch,
                       a.nsplit, nk, a.Kp, scale, grad, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* out[n] (= or +=) sum_m dy[m][n]; N % 4 == 0; scratch: >= ceil(M/512)*N floats */
int amp_colsum(amp_ctx* ctx, const float* dy, int M, int N, float* scratch, float* out, int accumulate) {
    AMP_REQUIRE(ctx && dy && scratch && out && M >= 0 && N > 0 && N % 4 == 0, "amp_colsum: bad argument (N %% 4 != 0?)");
    const int parts = std::max(1, amp::cdiv(M, COLSUM_ROWS));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(parts), dim3(256), 0, ctx->stream, dy, M, N / 4, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(amp::cdiv(N, 64)), dim3(64), 0, ctx->stream, scratch, parts, N, out, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* wt [Cin][KH][KW][Cout] = spatially flipped, channel-transposed, optionally scaled copy of w [Cout][KH][KW][Cin] */
int amp_dgrad_weights(amp_ctx* ctx, const float* w, const float* scale, int Cout, int KH, int KW, int Cin, float* wt) {
    AMP_REQUIRE(ctx && w && wt, "amp_dgrad_weights: null argument");
    const size_t total = (size_t)Cout * KH * KW * Cin;
    hipLaunchKernelGGL(dgrad_weight_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, ctx->stream, w, scale,
                       Cout, KH, KW, Cin, wt);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
