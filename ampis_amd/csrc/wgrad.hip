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
#include <string.h>

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
    const unsigned int* rowtab;   // [KH*KW][Mpad] X-row byte offsets (wgrad_rowtab_kernel)
    int Mpad;
    float* bias_partial;    // [S][N] column sums of dY per slice (bias gradient), written by the tile_c == 0 workgroups; nullptr: off
    unsigned int div_howo_mul, div_wo_mul;   // x / d == (x * mul) >> shr for x < 2^29 (mul = ceil(2^shr / d))
    int div_howo_shr, div_wo_shr;
};

__device__ __forceinline__ unsigned int fastdiv(unsigned int x, unsigned int mul, int shr) {
    return (unsigned int)(((unsigned long long)x * mul) >> shr);
}
void set_fastdiv(unsigned int d, unsigned int* mul, int* shr) {
    int l = 0;
    while ((1ull << l) < d) ++l;
    *shr = 29 + l;
    *mul = (unsigned int)(((1ull << *shr) + d - 1) / d);
}

__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BKW * 128];   // 2 buffers x (P 32x128 + Q 32x128)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave tile: 64 (n) x 64 (c')
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = amp::xcd_remap(blockIdx.x, gridDim.x);   // splits sharing a pixel range (same dY / X rows) stay inside one XCD's L2
    const int tile_c = bid % a.ntc; bid /= a.ntc;
    const int tile_n = bid % a.ntn; bid /= a.ntn;
    const int split = bid;
    const int n0 = tile_n * TN;
    const int kp0 = tile_c * TC;
    const int tap = kp0 / a.Cin;
    const int c0 = kp0 - tap * a.Cin;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BKW - 1) / BKW : 0;

    const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

    // staging: DMA instruction g (0..3) of this wave fills tile rows 8*wave + 2g + (lane>>5), 16-B chunk lane&31.
    // dY: the row byte offset is linear in m -> the step part rides in the scalar soffset, the lane part is constant.  Rows past
    // the slice need no predicate: their X rows are zero-filled, and rows past the tensor are out of the buffer (zero fill).
    // X: pixel -> (b,oy,ox) -> tap shift -> bounds -> byte offset is ~40 integer operations per row.  Done per lane in front of
    // the MFMAs it cost 20 % of the kernel, done on the scalar unit 9 %; it depends only on (pixel, tap), so wgrad_rowtab_kernel
    // tabulates it once per call ([tap][Mpad] byte offsets, 0x80000000 = zero row) and a lane fetches its row's entry one step
    // ahead: 4 dword loads + 4 adds per step.
    const int chunk = lane & 31;
    unsigned int p_voff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + 4 * chunk;               // N % 4 == 0: a chunk is all-in or all-out
        p_voff[g] = (n < a.N) ? (unsigned int)(((8 * wave + 2 * g + lh) * a.N + n) * 4) : OOB;
    }
    const unsigned int q_lane = (unsigned int)((c0 + 4 * chunk) * 4);
    const unsigned int* tab = a.rowtab + (size_t)tap * a.Mpad + m_begin + 8 * wave + lh;   // + 32*step + 2g
    unsigned int q_voff[4];                // table entries of the step whose DMA is issued next
    auto fetch_rows = [&](int step) {      // step < nsteps: m_begin + 32*step + 31 < Mpad
#pragma unroll
        for (int g = 0; g < 4; ++g) q_voff[g] = tab[step * BKW + 2 * g];
    };
    auto stage_dma = [&](int buf, int step) {
        float* P = lds + buf * (2 * BKW * 128);
        float* Q = P + BKW * 128;
        const int p_soff = (m_begin + step * BKW) * a.N * 4;              // uniform; < 2^31: dy_bytes < 2 GiB
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (__attribute__((address_space(3))) void*)(P + (8 * wave + 2 * g) * 128), 16,
                                                     (int)p_voff[g], p_soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(Q + (8 * wave + 2 * g) * 128), 16,
                                                     (int)(q_voff[g] + q_lane), 0, 0, 0);   // 0x80000000 + (< 64 KiB) stays out of range
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) {
        fetch_rows(0);
        stage_dma(0, 0);
        if (nsteps > 1) fetch_rows(1);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the LDS-DMA has landed (explicit: do not rely on the fence lowering)
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < nsteps;
            if (more) {
                stage_dma(cur ^ 1, step + 1);
                if (step + 2 < nsteps) fetch_rows(step + 2);
            }
            const float* P = lds + cur * (2 * BKW * 128) + wm * 64 + l31 + lh * 128;
            const float* Q = lds + cur * (2 * BKW * 128) + BKW * 128 + wn * 64 + l31 + lh * 128;
            // fragments of 4 k-pairs per group (8 ds_read2 feed 16 MFMAs), fetched one group ahead into the other register set.
            // The sched_barriers pin that order: left alone the scheduler sinks every read to just before its first use
            // (2 reads / wait / 4 MFMAs), which exposes the LDS latency after every 4th MFMA (104 TF).
            float fa[2][4][2], fb[2][4][2];
            auto fetch = [&](int tg, int bufi) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* p = P + (tg * 4 + u) * 256;
                    const float* q = Q + (tg * 4 + u) * 256;
                    fa[bufi][u][0] = p[0]; fa[bufi][u][1] = p[32];
                    fb[bufi][u][0] = q[0]; fb[bufi][u][1] = q[32];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int tg = 0; tg < BKW / 8; ++tg) {
                const int c = tg & 1;
                if (tg + 1 < BKW / 8) fetch(tg + 1, c ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][u][0], fb[c][u][0], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][u][0], fb[c][u][1], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][u][1], fb[c][u][0], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][u][1], fb[c][u][1], acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next tile's LDS-DMA has landed
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

// wgrad_f16x3_kernel (conv mode AMP_CONV_F16X3): the same GEMM on the f16 matrix pipe with the 3-MFMA operand split of
// conv_f16x3_kernel (conv.hip): x = hi + lo'*2^-11, dW ~= sum hi*hi + (hi*lo' + lo'*hi) * 2^-11, fp32 accumulation.
// v_mfma_f32_32x32x16_f16 wants 8 CONSECUTIVE reduction indices (pixels) per lane, but in memory the pixel is the slow index of
// both operands.  The transposition happens in the register stage that has to exist anyway for the split: wave w of a step owns
// pixels 8w..8w+7, a lane two adjacent columns; it loads its 8 x 2 block of each operand (8 coalesced 512-B row loads), splits it
// and writes per column the 8 hi halves and the 8 lo' halves as two 16-byte LDS stores into a column-major tile
// [128 columns][32 pixels] (144-byte column pitch: 64 B hi | 64 B lo' | pad; the pitch makes the ds_read_b128 fragment reads
// conflict-free).  X rows come from the same [tap][pixel] offset table as the fp32 kernel, read through the scalar unit (a wave's
// 8 rows are wave-uniform).  scale_p / scale_q (powers of two) lift a small operand (loss gradients) out of the f16 subnormals
// before the split; out_scale undoes it.  |operand * scale| must stay below 65504: a violation makes an accumulator non-finite,
// which raises *range_flag (the caller redoes the step on the fp32 MFMA).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr float LO_SCALE = 2048.0f;
constexpr int WPITCH = 144;                 // bytes per tile column
// A tile of C columns is two regions, even columns then odd columns, each C/2 columns of WPITCH bytes plus 128 B: a lane's two
// columns go to the same slot of the two regions, so the 8 lanes of a ds_write_b128 group hit 8 consecutive slots (pitch 9 x 16 B:
// all 32 banks once), and the 128-B skew puts the odd lanes of a ds_read_b128 group on the 32 banks its even lanes leave free.
constexpr int wreg(int cols) { return cols / 2 * WPITCH + 128; }
constexpr int WOP = 2 * wreg(128);          // bytes of a 128-column operand tile

// (An 8-wave variant with a 256-channel dY tile -- 24 instead of 32 split values per lane and step -- ran at the same speed.)
// XS = true: X is stored in the split hi|lo' row format (the trunk's native activation format; same row offsets as fp32): a lane
// fetches the 4 B of hi halves and the 4 B of lo' halves of its channel pair and hands them to the tile as they are -- no split
// arithmetic for that operand (the halves are what the in-kernel split of the fp32 value would produce: same MFMA inputs, same sums).
// PS = true: dY is stored in the split row format too, ALREADY multiplied by `scale` (the scaled loss gradients a data-gradient convolution
// on the ring kernel writes): passed through like XS; the sums still get out_scale.
template <int SC, bool XS = false, bool BIAS = false, bool PS = false>   // SC 0: no operand scaling, 1: dY * scale, 2: X * scale; BIAS: column sums of dY on the side
__global__ __launch_bounds__(256, 2) void wgrad_f16x3_kernel(const WgradArgs a, const float scale, const float out_scale, int* range_flag) {
    constexpr int REG = wreg(128);
    constexpr int STAGE = 2 * WOP;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // wave tile 64 (n) x 64 (c')
    const int l31 = lane & 31, lh = lane >> 5;
    const int mg = wave;                           // staging: this wave's 8-pixel group

    int bid = amp::xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = bid % a.ntc; bid /= a.ntc;
    const int tile_n = bid % a.ntn; bid /= a.ntn;
    const int split = bid;
    const int n0 = tile_n * TN;
    const int kp0 = tile_c * TC;
    const int tap = kp0 / a.Cin;
    const int c0 = kp0 - tap * a.Cin;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BKW - 1) / BKW : 0;

    const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);

    const int pcol = 2 * lane;                     // tile column pair of this lane
    const int pc = n0 + pcol;
    const unsigned int p_voff = (pc < a.N) ? (PS ? (unsigned int)((pc >> 5) * 128 + (pc & 31) * 2) : (unsigned int)(pc * 4)) : OOB;   // N even: a pair is all-in or all-out
    const int qc = c0 + 2 * lane;                  // this lane's channel pair of X
    const unsigned int q_lane = XS ? (unsigned int)((qc >> 5) * 128 + (qc & 31) * 2) : (unsigned int)(qc * 4);
    const unsigned int* tab = a.rowtab + (size_t)tap * a.Mpad + m_begin + 8 * mg;   // wave-uniform: scalar loads

    unsigned int qt[8];                // X-row byte offsets of the step fetched next
    u32x2 rp[8], rq[8];                // fetched, not yet split
    auto load_tab = [&](int step) {
#pragma unroll
        for (int r = 0; r < 8; ++r) qt[r] = tab[step * BKW + r];
    };
    auto fetch = [&](int step) {
        const int row0 = m_begin + step * BKW + 8 * mg;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (PS) {
                rp[r][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_dy, (int)p_voff, (row0 + r) * a.N * 4, 0);
                rp[r][1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_dy, (int)p_voff + 64, (row0 + r) * a.N * 4, 0);
            } else {
                rp[r] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_dy, (int)p_voff, (row0 + r) * a.N * 4, 0);   // past the tensor: zero fill
            }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (XS) {
                rq[r][0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, (int)(qt[r] + q_lane), 0, 0);        // hi halves of the pair
                rq[r][1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, (int)(qt[r] + q_lane + 64), 0, 0);   // lo' halves
            } else {
                rq[r] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_x, (int)(qt[r] + q_lane), 0, 0);         // 0x80000000 + lane part: zero fill
            }
        }
    };
    // BIAS: bias gradient = column sums of dY.  Every lane adds up the dY values it stages anyway (raw, before the 2^16 lift): its two
    // columns over its wave's 8 rows of every step, unconditionally -- a branch here would take the split arithmetic out of the MFMAs'
    // basic block (measured: -7 % on the whole kernel); the registers are zeroed once nothing is fetched any more, so the closing
    // re-commit adds zeros.  Only the workgroups of the first (tap, c) tile write their sums; waves and slices are combined in fixed order.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 bsum = {0.f, 0.f};
    auto commit = [&](int buf) {
        unsigned char* P = lds + buf * STAGE + lane * WPITCH + mg * 16;
        unsigned char* Q = P + WOP;
        if (BIAS) {
#pragma unroll
            for (int r = 0; r < 8; ++r) bsum = bsum + __builtin_bit_cast(f32x2, rp[r]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f16x8 ph, pl;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (PS) {
                    const unsigned int hu = rp[r][0], lu = rp[r][1];
                    ph[r] = __builtin_bit_cast(_Float16, (unsigned short)(q ? (hu >> 16) : (hu & 0xffffu)));
                    pl[r] = __builtin_bit_cast(_Float16, (unsigned short)(q ? (lu >> 16) : (lu & 0xffffu)));
                } else {
                    const unsigned int pu = rp[r][q];   // (bit_cast of a vector ELEMENT lvalue reads element 0)
                    float x = __builtin_bit_cast(float, pu);
                    if (SC == 1) x *= scale;
                    const _Float16 h = (_Float16)x;
                    ph[r] = h;
                    pl[r] = (_Float16)((x - (float)h) * LO_SCALE);
                }
            }
            *reinterpret_cast<f16x8*>(P + q * REG) = ph;
            *reinterpret_cast<f16x8*>(P + q * REG + 64) = pl;
            f16x8 qh, ql;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (XS) {
                    const unsigned int hu = rq[r][0], lu = rq[r][1];
                    qh[r] = __builtin_bit_cast(_Float16, (unsigned short)(q ? (hu >> 16) : (hu & 0xffffu)));
                    ql[r] = __builtin_bit_cast(_Float16, (unsigned short)(q ? (lu >> 16) : (lu & 0xffffu)));
                } else {
                    const unsigned int qu = rq[r][q];
                    float y = __builtin_bit_cast(float, qu);
                    if (SC == 2) y *= scale;
                    const _Float16 h = (_Float16)y;
                    qh[r] = h;
                    ql[r] = (_Float16)((y - (float)h) * LO_SCALE);
                }
            }
            *reinterpret_cast<f16x8*>(Q + q * REG) = qh;
            *reinterpret_cast<f16x8*>(Q + q * REG + 64) = ql;
        }
    };

    f32x16 acc[2][2], acx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }

    if (nsteps > 0) {
        load_tab(0);
        fetch(0);
        if (nsteps > 1) load_tab(1);
        commit(0);
        __syncthreads();
        if (nsteps > 1) {
            fetch(1);
            if (nsteps > 2) load_tab(2);
        } else if (BIAS) {
#pragma unroll
            for (int r = 0; r < 8; ++r) rp[r] = u32x2{0u, 0u};
        }
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            const unsigned char* P = lds + cur * STAGE + (l31 & 1) * REG + (wm * 32 + (l31 >> 1)) * WPITCH + lh * 16;
            const unsigned char* Q = lds + cur * STAGE + WOP + (l31 & 1) * REG + (wn * 32 + (l31 >> 1)) * WPITCH + lh * 16;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8*>(P + i * 16 * WPITCH + kk * 32);
                    al[i] = *reinterpret_cast<const f16x8*>(P + i * 16 * WPITCH + kk * 32 + 64);
                    bh[i] = *reinterpret_cast<const f16x8*>(Q + i * 16 * WPITCH + kk * 32);
                    bl[i] = *reinterpret_cast<const f16x8*>(Q + i * 16 * WPITCH + kk * 32 + 64);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            commit(cur ^ 1);     // operands of step+1 (loads issued a step ago).  Unconditional, so that the split arithmetic shares
                                 // a basic block with the MFMAs and fills their issue gaps; after the last step it rewrites stale
                                 // registers into the buffer nobody reads again
            __syncthreads();                             // LDS[cur] is free, LDS[cur^1] complete
            if (step + 2 < nsteps) {
                fetch(step + 2);
                if (step + 3 < nsteps) load_tab(step + 3);
            } else if (BIAS) {
#pragma unroll
                for (int r = 0; r < 8; ++r) rp[r] = u32x2{0u, 0u};      // nothing fresh any more: the closing re-commit must add zeros
            }
        }
    }
    if (BIAS && tile_c == 0 && a.bias_partial) {     // block-uniform; the operand tiles are dead (the loop ends on a barrier)
        float* red = reinterpret_cast<float*>(lds);
        red[wave * TN + pcol] = bsum[0];
        red[wave * TN + pcol + 1] = bsum[1];
        __syncthreads();
        if (tid < TN && n0 + tid < a.N)
            a.bias_partial[(size_t)split * a.N + n0 + tid] = __fadd_rn(__fadd_rn(__fadd_rn(red[tid], red[TN + tid]), red[2 * TN + tid]), red[3 * TN + tid]);
    }
    float* out = a.partial + (size_t)split * a.N * a.Kp;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kp = kp0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                float v = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                bad |= !(fabsf(v) <= 3.0e38f);
                if (SC != 0) v *= out_scale;
                if (n < a.N && kp < a.Kp) out[(size_t)n * a.Kp + kp] = v;
            }
        }
    if (bad && range_flag) *range_flag = 1;
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad_split_kernel: the same GEMM when BOTH operands are stored in the split hi|lo' row format (the backbone of a training step:
// activations of the native trunk, scaled split gradients).  Nothing is split and nothing is transposed in registers:
//   * a K-step is 32 pixels; the 32 x 128-channel slab of dY rows (512 B per pixel) and the 32 x 256-channel slab of X rows (1 KiB per
//     pixel, through the [tap][pixel] offset table) are copied AS THEY ARE into LDS by LDS-DMA (6 requests per wave and step, zero fill
//     beyond the tensor / outside the image), through a ring of three 48-KiB tiles with counted vmcnt and bare barriers, and the two
//     halves of the workgroup ping-pong between loading and multiplying -- conv_split_kernel's loop (conv.hip);
//   * the MFMA wants 8 consecutive PIXELS per lane and channel, the rows hold consecutive CHANNELS per pixel: gfx950's transposing LDS read
//     does the rest -- ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row x 16-column block of 16-bit elements, so
//     two of them are the f16x8 operand of v_mfma_f32_16x16x32_f16 (probed on the GPU: tools/tr_probe.hip);
//   * 16-B chunk j of row R lives at position j ^ (2 s(R)), s(R) = (R & 3) | ((R >> 3) & 1) << 2 (applied to the DMA's SOURCE chunk):
//     the 8 row pieces (2 groups x 4 rows x 32 B) a half-wave reads at once then cover the 64 banks exactly once.
// X is the MFMA's A operand (rows c), dY its B operand (columns n): a lane ends with 4 consecutive (tap, c) of one n -- 16-B stores
// into the slab [split][n][K'].  128 (n) x 256 (tap, c) tile, 8 waves of 64 x 64; K' tiles of 256 may straddle two taps (Cin = 128).
constexpr int WS_TN = 128, WS_TC = 256;
constexpr int WS_P = 32 * 512, WS_Q = 32 * 1024, WS_TILE = WS_P + WS_Q;       // bytes
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 tr_pair(const unsigned char* p) {          // pixels 8 lq .. 8 lq + 7 of this lane's channel
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 1024));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ f16x8 tr_pair_p(const unsigned char* p) {        // the same on the 512-B rows of the dY tile
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * 512));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

// The fragment reads as inline asm: behind the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of every step's reads (an LDS read
// "may alias" the LDS-DMA still in flight -- it does not for the plain ds_read_b128 of conv_split_kernel), which drains the youngest tile's
// requests and turns the ring's prefetch distance of two tiles into one.  The asm is opaque to that pass, so the data's arrival is
// waited for by hand: tr_wait() -- `s_waitcnt lgkmcnt(0)` that names every fragment register as in/out, so no MFMA can be scheduled above it.
template <int SECOND>
__device__ __forceinline__ f16x8 tr_pair_asm(unsigned int addr) {
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(SECOND) : "memory");
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(f16x8, v);
}

__global__ __launch_bounds__(512, 1) void wgrad_split_kernel(const WgradArgs a, const float out_scale, int* range_flag) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * WS_TILE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;       // wave tile: 64 n (wm) x 64 (tap, c) (wn)
    const int l15 = lane & 15, lq = lane >> 4;

    int bid = amp::xcd_remap(blockIdx.x, gridDim.x);
    const int tile_c = bid % a.ntc; bid /= a.ntc;
    const int tile_n = bid % a.ntn; bid /= a.ntn;
    const int split = bid;
    const int n0 = tile_n * WS_TN;
    const int kp0 = tile_c * WS_TC;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    const int nsteps = (m_end > m_begin) ? (m_end - m_begin + BKW - 1) / BKW : 0;

    const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, a.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
    auto swz = [](int R) { return ((R & 3) | (((R >> 3) & 1) << 2)) << 1; };

    // ---- staging: dY, 2 requests per wave and step, each 2 rows x 512 B; lane -> (row, chunk position), source chunk swizzled ----
    unsigned int p_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = 2 * (2 * wave + i) + (lane >> 5);
        const int src = (lane & 31) ^ swz(R);
        // the tile's 128 channels are 4 groups of 128 B, contiguous in a split row; the row offset (pixel * N * 4) is added per step
        p_voff[i] = (n0 < a.N) ? (unsigned int)((size_t)(m_begin + R) * a.N * 4 + (size_t)(n0 >> 5) * 128 + 16 * src) : OOB;
    }
    const unsigned int p_step = (unsigned int)(BKW * a.N * 4);
    // ---- X: 4 requests per wave and step, one row of 1 KiB each; the 256 (tap, c) of the tile are one run of a tap's channels, or (Cin =
    //      128) two runs of two taps; a lane's half decides which ----
    const int tap0 = kp0 / a.Cin, c0 = kp0 - tap0 * a.Cin;
    const bool two = a.Cin < WS_TC;                                 // Cin == 128: positions 0..31 -> tap0, 32..63 -> tap0 + 1
    const int ntaps = a.KH * a.KW;
    unsigned int q_lane[4];
    bool q_second[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = 4 * wave + i;
        const int src = lane ^ swz(R);
        q_second[i] = two && src >= 32;
        q_lane[i] = two ? (unsigned int)(16 * (src & 31)) : (unsigned int)(c0 * 4 + 16 * src);
    }
    // the row offsets are wave-uniform: read through the scalar unit (constant address space), one K-step ahead of their use
    typedef const __attribute__((address_space(4))) unsigned int* ctab_t;
    ctab_t tab0 = (ctab_t)(a.rowtab + (size_t)min(tap0, ntaps - 1) * a.Mpad + m_begin + 4 * wave);
    ctab_t tab1 = (ctab_t)(a.rowtab + (size_t)min(tap0 + 1, ntaps - 1) * a.Mpad + m_begin + 4 * wave);
    const bool tap0_ok = kp0 < a.Kp, tap1_ok = two && tap0 + 1 < ntaps;
    // Two sets: a step's stage() consumes `qt`, the scalar loads of the NEXT stage's rows go into `qn` and are issued at the START of a step, in
    // front of the fragment reads -- lgkmcnt counts scalar loads and LDS reads alike and scalar loads return out of order, so every wait for
    // the fragments is lgkmcnt(0): issued behind the staging (as they were until the end of round 3) their latency sat between the early
    // half's barrier and its first MFMA in every K-step (SQ_WAIT_ANY 2.8 x conv_split_kernel's per flop, tools/pmc_wgrad_vs_conv.sh).
    unsigned int qt0[4], qt1[4], qn0[4], qn1[4];
    // (one 16-B scalar load per tap: the index is a multiple of 4 -- Mpad, m_begin and BKW are multiples of 32 -- and both pointers stay inside
    //  the table whatever the tile, so the loads are unconditional and the OOB marker is a select: no branch per entry)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) u32x4* ctab4_t;
    auto load_tab = [&](int st) {
        const u32x4 t0 = *(ctab4_t)(tab0 + st * BKW);
        const u32x4 t1 = *(ctab4_t)(tab1 + st * BKW);
#pragma unroll
        for (int i = 0; i < 4; ++i) { qn0[i] = t0[i]; qn1[i] = t1[i]; }      // raw: the first USE of a loaded value is where the wave waits for it
    };
    auto take_tab = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { qt0[i] = tap0_ok ? qn0[i] : OOB; qt1[i] = tap1_ok ? qn1[i] : OOB; }
    };

    int staged = 0;                                                  // next K-step to stage
    auto stage = [&](int buf) {
        unsigned char* P = lds + buf * WS_TILE;
        unsigned char* Q = P + WS_P;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (__attribute__((address_space(3))) void*)(P + (2 * wave + i) * 1024), 16,
                                                     (int)p_voff[i], staged * (int)p_step, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int row = q_second[i] ? qt1[i] : qt0[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(Q + (4 * wave + i) * 1024), 16,
                                                     (int)(row + q_lane[i]), 0, 0, 0);      // 0x80000000 + lane part: zero fill
        }
        ++staged;
    };

    // ---- fragment addresses: row R = 8 lq + q (+ 4 for the second read); lane 4 q + p supplies row q, 8-B piece p of the block's 32 B ----
    const int fq = (lane >> 2) & 3, fp = lane & 3;
    const int fR = 8 * lq + fq;
    const int fs = swz(fR);                                          // (the same for row fR + 4)
    int offP[4][2], offQ[4][2];                                     // [block][hi / lo] byte offsets inside a tile
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int chP = 8 * (2 * wm + (b >> 1)) + 2 * (b & 1) + 4 * h;          // chunk of the block's 32 B in a 512-B dY row
            const int chQ = 8 * (2 * wn + (b >> 1)) + 2 * (b & 1) + 4 * h;          // ... in a 1-KiB X row
            offP[b][h] = fR * 512 + 16 * ((chP + (fp >> 1)) ^ fs) + 8 * (fp & 1);
            offQ[b][h] = WS_P + fR * 1024 + 16 * ((chQ + (fp >> 1)) ^ fs) + 8 * (fp & 1);
        }

    f32x4w acc[4][4], acx[4][4];                                    // [c block][n block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    f16x8 xh[4], xl[4], yh[4], yl[4];

    const unsigned int lds_base = (unsigned int)(size_t)lds;          // LDS byte address of the ring (low half of the generic address)
    auto load = [&](int buf) {
        const unsigned int T = lds_base + (unsigned int)(buf * WS_TILE);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            xh[b] = tr_pair_asm<4 * 1024>(T + (unsigned int)offQ[b][0]);
            xl[b] = tr_pair_asm<4 * 1024>(T + (unsigned int)offQ[b][1]);
            yh[b] = tr_pair_asm<4 * 512>(T + (unsigned int)offP[b][0]);
            yl[b] = tr_pair_asm<4 * 512>(T + (unsigned int)offP[b][1]);
        }
    };
    auto tr_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]),
                       "+v"(yh[0]), "+v"(yh[1]), "+v"(yh[2]), "+v"(yh[3]), "+v"(yl[0]), "+v"(yl[1]), "+v"(yl[2]), "+v"(yl[3])
                     :: "memory");
    };
    auto mfma = [&]() {
        tr_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[i], yh[j], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[i], yl[j], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[i], yh[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        load_tab(0); take_tab();
        stage(0);
        if (nsteps > 1) { load_tab(1); take_tab(); stage(1); }
        if (nsteps > 2) { load_tab(2); take_tab(); }                 // qt: the rows of K-step 2, staged in step 0
        int cur = 0, nxt = 2;
        auto open_step = [&](int step) {
            if (step + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0070 | 6);     // all but the youngest tile's 6 requests; lgkmcnt(0)
            else __builtin_amdgcn_s_waitcnt(0x0070);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto advance = [&]() { cur = (cur == 2) ? 0 : cur + 1; nxt = (nxt == 2) ? 0 : nxt + 1; };
        if (wave < 4) {
            for (int step = 0; step < nsteps; ++step) {
                open_step(step);
                if (step + 3 < nsteps) load_tab(step + 3);           // scalar loads first: their latency passes under the fragment reads
                __builtin_amdgcn_sched_barrier(0);
                load(cur);
                __builtin_amdgcn_sched_barrier(0);
                if (step + 2 < nsteps) stage(nxt);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                mfma();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (step + 3 < nsteps) take_tab();
                advance();
            }
        } else {
            for (int step = 0; step < nsteps; ++step) {
                open_step(step);
                __builtin_amdgcn_s_setprio(1);
                if (step > 0) mfma();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                if (step + 3 < nsteps) load_tab(step + 3);
                __builtin_amdgcn_sched_barrier(0);
                load(cur);
                __builtin_amdgcn_sched_barrier(0);
                if (step + 2 < nsteps) stage(nxt);
                __builtin_amdgcn_sched_barrier(0);
                if (step + 3 < nsteps) take_tab();
                advance();
            }
            mfma();
        }
    }
    // ---- slab [split][n][K']: lane = (n = l15 of block j, 4 consecutive (tap, c) = 4 lq .. of block i) ----
    float* out = a.partial + (size_t)split * a.N * a.Kp;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int kp = kp0 + wn * 64 + i * 16 + 4 * lq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wm * 64 + j * 16 + l15;
            f32x4w v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                bad |= !(fabsf(v[e]) <= 3.0e38f);
                v[e] *= out_scale;
            }
            if (n < a.N && kp < a.Kp) *reinterpret_cast<f32x4w*>(out + (size_t)n * a.Kp + kp) = v;       // K' % 4 == 0
        }
    }
    if (bad && range_flag) *range_flag = 1;
}

template <bool XS, bool BIAS>
void launch_wgrad_f16x3_v(const WgradArgs& a, int blocks, int dy_shift, int x_shift, float sc, float osc, hipStream_t st, int* flag) {
    if (dy_shift) AMP_TIMED_LAUNCH((wgrad_f16x3_kernel<1, XS, BIAS>), dim3(blocks), dim3(256), 0, st, a, sc, osc, flag);
    else if (x_shift && !XS) AMP_TIMED_LAUNCH((wgrad_f16x3_kernel<2, false, BIAS>), dim3(blocks), dim3(256), 0, st, a, sc, osc, flag);
    else AMP_TIMED_LAUNCH((wgrad_f16x3_kernel<0, XS, BIAS>), dim3(blocks), dim3(256), 0, st, a, sc, osc, flag);
}
// fmt: bit 0 = x in the split row format, bit 1 = dy in the split row format AND already multiplied by 2^dy_shift
void launch_wgrad_f16x3(const WgradArgs& a, int blocks, int dy_shift, int x_shift, hipStream_t st, int* flag, int fmt = 0) {
    const int sh = dy_shift ? dy_shift : x_shift;
    const float sc = ldexpf(1.0f, sh), osc = ldexpf(1.0f, -sh);     // exact powers of two
    const bool bias = a.bias_partial != nullptr;
    if (fmt & 2) {          // (no bias sums from a scaled split dy: the FrozenBN convolutions that use it have none)
        if (fmt & 1) AMP_TIMED_LAUNCH((wgrad_f16x3_kernel<1, true, false, true>), dim3(blocks), dim3(256), 0, st, a, sc, osc, flag);
        else AMP_TIMED_LAUNCH((wgrad_f16x3_kernel<1, false, false, true>), dim3(blocks), dim3(256), 0, st, a, sc, osc, flag);
    } else if (fmt & 1) { if (bias) launch_wgrad_f16x3_v<true, true>(a, blocks, dy_shift, x_shift, sc, osc, st, flag); else launch_wgrad_f16x3_v<true, false>(a, blocks, dy_shift, x_shift, sc, osc, st, flag); }
    else { if (bias) launch_wgrad_f16x3_v<false, true>(a, blocks, dy_shift, x_shift, sc, osc, st, flag); else launch_wgrad_f16x3_v<false, false>(a, blocks, dy_shift, x_shift, sc, osc, st, flag); }
}

// rowtab[tap][m] = byte offset of the input pixel that output pixel m sees through tap (ky,kx), or 0x80000000 (outside the image, or
// m >= M): the operand staging of wgrad_mfma_kernel reads it instead of redoing the index arithmetic every step.
__global__ void wgrad_rowtab_kernel(const WgradArgs a, unsigned int* __restrict__ tab) {
    const size_t total = (size_t)a.KH * a.KW * a.Mpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i / a.Mpad);
        const unsigned int m = (unsigned int)(i - (size_t)tap * a.Mpad);
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
        const unsigned int rem = m - b * (unsigned int)(a.Ho * a.Wo);
        const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
        const unsigned int ox = rem - oy * (unsigned int)a.Wo;
        const int iy = (int)oy * a.stride - a.pad + ky, ix = (int)ox * a.stride - a.pad + kx;
        const bool v = (int)m < a.M && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        tab[i] = v ? (unsigned int)(((b * a.H + iy) * a.W + ix) * a.Cin) * 4u : OOB;
    }
}

// grad[n][k'] (= or +=) scale[n] * sum_s partial[s][n][k']   (fixed order -> reproducible)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t nk, int Kp, const float* __restrict__ scale,
                                    float* __restrict__ grad, int accumulate, const float* __restrict__ bias_partial, int N,
                                    float* __restrict__ bias_grad, int bias_accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nk; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s = __fadd_rn(s, partial[(size_t)k * nk + i]);
        if (scale) s = __fmul_rn(s, scale[i / Kp]);
        grad[i] = accumulate ? __fadd_rn(grad[i], s) : s;
    }
    if (bias_partial)      // the bias gradient the MFMA kernel summed on the side (WgradArgs::bias_partial): slices in fixed order
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int k = 0; k < nsplit; ++k) s = __fadd_rn(s, bias_partial[(size_t)k * N + i]);
            bias_grad[i] = bias_accumulate ? __fadd_rn(bias_grad[i], s) : s;
        }
}

// column sums of dY [M][N] (bias gradients): two passes, fixed order.  Pass 1: a workgroup owns COLSUM_ROWS rows; a thread owns one
// 16-byte column group and every (256 / N4)-th row, so a wave reads whole 1 KiB row segments; row lanes are combined through LDS.
constexpr int COLSUM_ROWS = 512;
typedef float cs_f4 __attribute__((ext_vector_type(4)));
// split_out != nullptr: the pass also leaves dY * scale in the split hi|lo' row format (the scaled split gradient wgrad_split_kernel
// stages by LDS-DMA): one read of dY serves the bias gradient and the operand conversion.
typedef _Float16 cs_h4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, int M, int N4, float* __restrict__ partial,
                                                             float* __restrict__ split_out, float scale) {
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
                if (split_out) {
                    cs_h4 hi, lo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = v[q] * scale;
                        const _Float16 h = (_Float16)x;
                        hi[q] = h;
                        lo[q] = (_Float16)((x - (float)h) * LO_SCALE);
                    }
                    const int ch = 4 * c;
                    char* ob = reinterpret_cast<char*>(split_out + (size_t)r * N4 * 4) + (size_t)(ch >> 5) * 128 + (size_t)(ch & 31) * 2;
                    *reinterpret_cast<cs_h4*>(ob) = hi;
                    *reinterpret_cast<cs_h4*>(ob + 64) = lo;
                }
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
// The converting variant (amp_colsum_split) on 8 channels per thread: 32 B of dY in, 16 B of hi halves and 16 B of lo' halves out per row
// and thread (the 4-channel layout above writes 8-B pieces: 0.9 ms for the 2 GB of a p2-level tensor); 32 column groups x 8 row lanes per
// workgroup, the row lanes combined through LDS in lane order.  N % 8 == 0.
typedef _Float16 cs_h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void colsum_split_kernel(const float* __restrict__ dy, int M, int N, float* __restrict__ partial,
                                                           float* __restrict__ split_out, float scale) {
    __shared__ float red[8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
    for (int cbase = 0; cbase < N; cbase += 256) {
        const int ch = cbase + 8 * cg;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ch < N) {
            const size_t col_b = (size_t)(ch >> 5) * 128 + (size_t)(ch & 31) * 2;
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += 8) {
                const cs_f4 v0 = *reinterpret_cast<const cs_f4*>(dy + (size_t)r * N + ch), v1 = *reinterpret_cast<const cs_f4*>(dy + (size_t)r * N + ch + 4);
                cs_h8 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = q < 4 ? v0[q] : v1[q - 4];
                    acc[q] = __fadd_rn(acc[q], v);
                    const float x = v * scale;
                    const _Float16 h = (_Float16)x;
                    hi[q] = h;
                    lo[q] = (_Float16)((x - (float)h) * LO_SCALE);
                }
                char* ob = reinterpret_cast<char*>(split_out + (size_t)r * N) + col_b;
                *reinterpret_cast<cs_h8*>(ob) = hi;
                *reinterpret_cast<cs_h8*>(ob + 64) = lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) red[rl][cg][q] = acc[q];
        __syncthreads();
        if (rl == 0 && ch < N) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = red[0][cg][q];
                for (int k = 1; k < 8; ++k) t = __fadd_rn(t, red[k][cg][q]);
                partial[(size_t)blockIdx.x * N + ch + q] = t;
            }
        }
    }
}

// Column sums of a tensor that is ALREADY split rows of dy * 2^shift (a data-gradient convolution wrote it that way): read-only, half the
// bytes of colsum_split_kernel's pass; partial sums of the decoded values (hi + lo' * 2^-11) * inv_scale, same tree as above.
__global__ __launch_bounds__(256) void colsum_of_split_kernel(const float* __restrict__ dy_split, int M, int N, float* __restrict__ partial, float inv_scale) {
    __shared__ float red[8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
    for (int cbase = 0; cbase < N; cbase += 256) {
        const int ch = cbase + 8 * cg;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ch < N) {
            const size_t col_b = (size_t)(ch >> 5) * 128 + (size_t)(ch & 31) * 2;
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += 8) {
                const char* ib = reinterpret_cast<const char*>(dy_split + (size_t)r * N) + col_b;
                const cs_h8 hi = *reinterpret_cast<const cs_h8*>(ib), lo = *reinterpret_cast<const cs_h8*>(ib + 64);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(__fadd_rn((float)hi[q], __fmul_rn((float)lo[q], 1.0f / LO_SCALE)), inv_scale));
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) red[rl][cg][q] = acc[q];
        __syncthreads();
        if (rl == 0 && ch < N) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = red[0][cg][q];
                for (int k = 1; k < 8; ++k) t = __fadd_rn(t, red[k][cg][q]);
                partial[(size_t)blockIdx.x * N + ch + q] = t;
            }
        }
    }
}

// Pass 2: 32 columns x 8 part-lanes per workgroup; part-lane l adds parts l, l+8, ... in order (4 independent loads in flight),
// the 8 lanes are combined in lane order: a fixed summation tree -> run-to-run reproducible.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nparts, int N, float* __restrict__ out,
                                                           int accumulate) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + col;
    float s = 0.f;
    if (n < N) {
        int p = pl;
        for (; p + 24 < nparts; p += 32) {
            const float v0 = partial[(size_t)p * N + n], v1 = partial[(size_t)(p + 8) * N + n];
            const float v2 = partial[(size_t)(p + 16) * N + n], v3 = partial[(size_t)(p + 24) * N + n];
            s = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, v0), v1), v2), v3);
        }
        for (; p < nparts; p += 8) s = __fadd_rn(s, partial[(size_t)p * N + n]);
    }
    red[pl][col] = s;
    __syncthreads();
    if (pl == 0 && n < N) {
        float t = red[0][col];
#pragma unroll
        for (int k = 1; k < 8; ++k) t = __fadd_rn(t, red[k][col]);
        out[n] = accumulate ? __fadd_rn(out[n], t) : t;
    }
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

namespace {
// Pixel slices (split-K): the grid runs in rounds of 512 workgroups (2 per CU x 256 CUs), and a round costs its slice's steps plus a
// fixed prologue / epilogue (~12 steps: first loads, the 64 KiB partial tile).  Take the slice count with the cheapest
// rounds x (steps + 12): e.g. 36 tiles x 14 slices = 504 workgroups in ONE round instead of 36 x 57 = 2052 in four and a 4-workgroup
// fifth.  Fewer slices also mean fewer slabs for wgrad_reduce_kernel to add.
int pick_nsplit(long long M, int tiles) {
    const int smax = (int)std::max(1LL, std::min(256LL, M / 256));
    long long best = -1;
    int best_s = 1;
    for (int s = 1; s <= smax; ++s) {
        const long long steps = amp::cdiv((int)((M + s - 1) / s), BKW);
        const long long cost = (long long)amp::cdiv(tiles * s, 512) * (steps + 12);
        if (best < 0 || cost < best) { best = cost; best_s = s; }
    }
    return best_s;
}
// the same for wgrad_split_kernel: 128 x 256 tiles, one 512-thread workgroup per CU (rounds of 256), ~10 steps of prologue / epilogue
bool ring_shape(int N, int Cin, int Kp) { return N % WS_TN == 0 && Cin % 128 == 0 && Kp >= WS_TC; }
int pick_nsplit_ring(long long M, int tiles) {
    const int smax = (int)std::max(1LL, std::min(256LL, M / 256));
    long long best = -1;
    int best_s = 1;
    for (int s = 1; s <= smax; ++s) {
        const long long steps = amp::cdiv((int)((M + s - 1) / s), BKW);
        const long long cost = (long long)amp::cdiv(tiles * s, 256) * (steps + 10);
        if (best < 0 || cost < best) { best = cost; best_s = s; }
    }
    return best_s;
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
    int nsplit = pick_nsplit(M, tiles);
    if (ring_shape(d->Cout, d->Cin, Kp)) nsplit = std::max(nsplit, pick_nsplit_ring(M, (d->Cout / WS_TN) * amp::cdiv(Kp, WS_TC)));
    const long long Mpad = (M + BKW - 1) / BKW * BKW;
    return (size_t)nsplit * d->Cout * Kp + (size_t)d->KH * d->KW * Mpad + (size_t)nsplit * d->Cout;   // partial slabs + the row table (4-byte entries) + bias slices
}

int amp_conv2d_wgrad(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                     float* grad, int accumulate) {
    return amp_conv2d_wgrad_scaled(ctx, d, x, dy, scale, scratch, grad, accumulate, 0, 0);
}

int amp_conv2d_wgrad_scaled(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                            float* grad, int accumulate, int dy_shift, int x_shift) {
    return amp_conv2d_wgrad_fmt(ctx, d, x, dy, scale, scratch, grad, accumulate, dy_shift, x_shift, 0, nullptr, 0);
}

int amp_conv2d_wgrad_fmt(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* dy, const float* scale, float* scratch,
                         float* grad, int accumulate, int dy_shift, int x_shift, int x_split, float* bias_grad, int bias_accumulate) {
    AMP_REQUIRE(ctx && d && x && dy && scratch && grad, "amp_conv2d_wgrad: null argument");
    AMP_REQUIRE(!bias_grad || ctx->conv_mode == AMP_CONV_F16X3, "amp_conv2d_wgrad_fmt: the fused bias gradient needs AMP_CONV_F16X3 (use amp_colsum)");
    AMP_REQUIRE(x_split >= 0 && x_split < 4, "amp_conv2d_wgrad_fmt: x_split is a set of bits: 1 = x split, 2 = dy split and pre-scaled by 2^dy_shift");
    AMP_REQUIRE(!(x_split & 1) || (ctx->conv_mode == AMP_CONV_F16X3 && x_shift == 0 && d->Cin % 32 == 0),
                "amp_conv2d_wgrad_fmt: a split-format x needs AMP_CONV_F16X3, no x shift and Cin %% 32 == 0");
    AMP_REQUIRE(!(x_split & 2) || (ctx->conv_mode == AMP_CONV_F16X3 && x_shift == 0 && d->Cout % 32 == 0 && !bias_grad),
                "amp_conv2d_wgrad_fmt: a split-format dy needs AMP_CONV_F16X3, no x shift, Cout %% 32 == 0 and no fused bias gradient");
    AMP_REQUIRE(dy_shift >= 0 && dy_shift <= 24 && x_shift >= 0 && x_shift <= 24 && (dy_shift == 0 || x_shift == 0),
                "amp_conv2d_wgrad: shifts must be in [0, 24] and at most one of them non-zero");
    AMP_REQUIRE(d->Cin % TC == 0, "amp_conv2d_wgrad: Cin=%d must be a multiple of %d", d->Cin, TC);
    AMP_REQUIRE(d->Cout % 4 == 0, "amp_conv2d_wgrad: Cout=%d must be a multiple of 4", d->Cout);
    int par = -1;
    if (ctx->reduce_async) {              // alternate the scratch; the reduction that last read this one must be done before it is overwritten
        par = ctx->side_parity; ctx->side_parity ^= 1;
        if (par == 1) scratch = ctx->side_scratch1;
        if (ctx->side_used[par]) AMP_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->side_ev[par], 0));
    }
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
    a.nsplit = pick_nsplit(M, tiles);
    a.rows_per_split = amp::cdiv(amp::cdiv(a.M, a.nsplit), BKW) * BKW;
    const size_t dyb = (size_t)M * a.N * 4, xb = (size_t)a.B * a.H * a.W * a.Cin * 4;
    AMP_REQUIRE(dyb < (size_t)OOB && xb < (size_t)OOB, "amp_conv2d_wgrad: operand larger than 2 GiB (split the batch)");
    a.dy_bytes = (unsigned int)dyb; a.x_bytes = (unsigned int)xb;
    AMP_REQUIRE(M + 64 < (1ll << 29), "amp_conv2d_wgrad: too many output pixels for the multiply-shift division");
    set_fastdiv((unsigned int)(a.Ho * a.Wo), &a.div_howo_mul, &a.div_howo_shr);
    set_fastdiv((unsigned int)a.Wo, &a.div_wo_mul, &a.div_wo_shr);
    a.Mpad = amp::cdiv(a.M, BKW) * BKW;
    // scratch layout (amp_conv_wgrad_scratch_floats): slabs of the larger of the two kernels' slice counts, the row table, the bias slices
    int nsplit_layout = a.nsplit;
    if (ring_shape(a.N, a.Cin, a.Kp)) nsplit_layout = std::max(nsplit_layout, pick_nsplit_ring(M, (a.N / WS_TN) * amp::cdiv(a.Kp, WS_TC)));
    unsigned int* rowtab = reinterpret_cast<unsigned int*>(scratch + (size_t)nsplit_layout * a.N * a.Kp);
    a.rowtab = rowtab;
    const size_t tab_n = (size_t)a.KH * a.KW * a.Mpad;
    a.bias_partial = bias_grad ? reinterpret_cast<float*>(rowtab + tab_n) : nullptr;
    // The table is a function of the geometry alone: a context keeps the tables of the shapes it has seen (a training loop asks for the same
    // ~30 every step: 69 launches and 0.4 ms per step otherwise) in a two-generation arena (common.h): bounded memory, no allocation inside
    // a step, and a run whose frames change (multi-scale training) keeps the tables of its recent frames instead of the first four it saw.
    // A table larger than a generation, or a failed arena allocation, falls back to the call's scratch as before.
    {
        const int key[9] = {a.B, a.H, a.W, a.Cin, a.KH, a.KW, a.stride, a.pad, a.Mpad};
        static const bool no_cache = getenv("AMP_NO_ROWTAB_CACHE") != nullptr;      // EXPERIMENT switch
        unsigned int* cached = nullptr;
        for (auto& t : ctx->rowtabs) if (memcmp(t.key, key, sizeof(key)) == 0) { cached = t.tab; break; }
        if (cached) ++ctx->rowtab_hits;
        if (!cached && !no_cache) {
            ++ctx->rowtab_misses;
            if (!ctx->rowtab_arena && ctx->rowtab_gen_words == 0) {
                const char* mb = getenv("AMP_ROWTAB_MB");
                const size_t total = (size_t)std::max(64, mb ? atoi(mb) : 512) << 20;
                AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (hipMalloc(&ctx->rowtab_arena, total) == hipSuccess) ctx->rowtab_gen_words = total / 8;      // two generations
                else { (void)hipGetLastError(); ctx->rowtab_arena = nullptr; ctx->rowtab_gen_words = (size_t)-1; }      // do not try again
            }
            const size_t words = (tab_n + 63) & ~(size_t)63;
            if (ctx->rowtab_arena && words <= ctx->rowtab_gen_words) {
                if (ctx->rowtab_used[ctx->rowtab_gen] + words > ctx->rowtab_gen_words) {
                    // the current generation is full: the other one's tables go (none of them was asked for since the last switch -- a hit
                    // re-homes nothing, so a table in steady use is re-made once per two switches at worst) and it becomes the current one
                    const int g = ctx->rowtab_gen ^ 1;
                    ctx->rowtabs.erase(std::remove_if(ctx->rowtabs.begin(), ctx->rowtabs.end(), [g](const amp_ctx::RowTab& t) { return t.gen == g; }), ctx->rowtabs.end());
                    ctx->rowtab_used[g] = 0;
                    ctx->rowtab_gen = g;
                    ++ctx->rowtab_flushes;
                }
                const int g = ctx->rowtab_gen;
                unsigned int* t = ctx->rowtab_arena + (size_t)g * ctx->rowtab_gen_words + ctx->rowtab_used[g];
                ctx->rowtab_used[g] += words;
                WgradArgs ta = a;
                hipLaunchKernelGGL(wgrad_rowtab_kernel, dim3((unsigned)std::min<size_t>((tab_n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, ta, t);
                amp_ctx::RowTab e;
                memcpy(e.key, key, sizeof(key)); e.tab = t; e.n = tab_n; e.gen = g;
                ctx->rowtabs.push_back(e);
                cached = t;
            }
        }
        if (cached) a.rowtab = cached;
        else hipLaunchKernelGGL(wgrad_rowtab_kernel, dim3((unsigned)std::min<size_t>((tab_n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, a, rowtab);
    }
    amp_prof_rec* rec = nullptr;      // live profile (amp_prof_begin): slot 2 = the weight-gradient MFMA kernel alone
    if (ctx->prof_on) {
        if (ctx->prof_used < ctx->prof_pool.size()) {
            rec = &ctx->prof_pool[ctx->prof_used++];
            rec->flops = 2.0 * (double)M * (double)a.N * (double)a.Kp;
            rec->variant = 2;
            // dY + X (a 1x1 layer reads only the pixels its stride samples) + dW once; the split-K slabs are not counted
            rec->bytes = 4.0 * ((double)M * a.N + ((a.KH == 1 && a.KW == 1) ? (double)M : (double)a.B * a.H * a.W) * a.Cin + (double)a.N * a.Kp);
            rec->M = a.N; rec->N = a.Kp; rec->K = M;
        } else {
            ctx->prof_truncated = true;
        }
    }
    {
    amp::ProfLaunchScope timed(rec ? rec->e0 : nullptr, rec ? rec->e1 : nullptr);      // the MFMA kernel below carries the events (not the reduce pass)
    static const bool ring_on = getenv("AMP_NO_WGRAD_RING") == nullptr;      // EXPERIMENT switch
    if (ctx->conv_mode == AMP_CONV_F16X3 && ring_on && (x_split & 3) == 3 && ring_shape(a.N, a.Cin, a.Kp)) {
        // both operands in the split row format: LDS-DMA ring + transposing reads (wgrad_split_kernel), 128 x 256 tiles.  The slab count
        // never exceeds the one the scratch was sized for.
        WgradArgs r = a;
        r.ntn = a.N / WS_TN;
        r.ntc = amp::cdiv(a.Kp, WS_TC);
        const int rtiles = r.ntn * r.ntc;
        r.nsplit = pick_nsplit_ring(M, rtiles);      // (amp_conv_wgrad_scratch_floats reserves the larger of the two slab counts)
        r.rows_per_split = amp::cdiv(amp::cdiv(r.M, r.nsplit), BKW) * BKW;
        AMP_TIMED_LAUNCH(wgrad_split_kernel, dim3(rtiles * r.nsplit), dim3(512), 0, ctx->stream, r, ldexpf(1.0f, -dy_shift), ctx->d_conv_flag);
        a.nsplit = r.nsplit;                       // the reduce below adds this many slabs
    } else if (ctx->conv_mode == AMP_CONV_F16X3) {
        launch_wgrad_f16x3(a, tiles * a.nsplit, dy_shift, x_shift, ctx->stream, ctx->d_conv_flag, x_split);   // the fp32 kernel needs no shift
    } else {
        AMP_TIMED_LAUNCH(wgrad_mfma_kernel, dim3(tiles * a.nsplit), dim3(256), 0, ctx->stream, a);
    }
    }
    const size_t nk = (size_t)a.N * a.Kp;
    hipStream_t rs = ctx->stream;
    if (par >= 0) {
        AMP_HIP_CHECK(hipEventRecord(ctx->wg_ev[par], ctx->stream));
        AMP_HIP_CHECK(hipStreamWaitEvent(ctx->side, ctx->wg_ev[par], 0));
        rs = ctx->side;
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<size_t>((nk + 255) / 256, 4096)), dim3(256), 0, rs, scratch,
                       a.nsplit, nk, a.Kp, scale, grad, accumulate, a.bias_partial, a.N, bias_grad, bias_accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    if (par >= 0) {
        AMP_HIP_CHECK(hipEventRecord(ctx->side_ev[par], ctx->side));
        ctx->side_used[par] = true; ctx->side_last = par;
    }
    return AMP_OK;
}

/* out[n] (= or +=) sum_m dy[m][n]; N % 4 == 0; scratch: >= ceil(M/512)*N floats */
int amp_colsum(amp_ctx* ctx, const float* dy, int M, int N, float* scratch, float* out, int accumulate) {
    AMP_REQUIRE(ctx && dy && scratch && out && M >= 0 && N > 0 && N % 4 == 0, "amp_colsum: bad argument (N %% 4 != 0?)");
    const int parts = std::max(1, amp::cdiv(M, COLSUM_ROWS));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(parts), dim3(256), 0, ctx->stream, dy, M, N / 4, scratch, (float*)nullptr, 1.0f);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(amp::cdiv(N, 32)), dim3(256), 0, ctx->stream, scratch, parts, N, out, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* second pass of the column sums alone (partial [parts][N] from a kernel that summed its slices itself: amp_small_k_dgrad_split) */
int amp_colsum_finish(amp_ctx* ctx, const float* partial, int parts, int N, float* out, int accumulate) {
    AMP_REQUIRE(ctx && partial && out && parts >= 1 && N > 0, "amp_colsum_finish: bad argument");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(amp::cdiv(N, 32)), dim3(256), 0, ctx->stream, partial, parts, N, out, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* amp_colsum that also leaves dy * 2^shift in the split row format (N % 32 == 0): one pass over dy for the bias gradient and for the
 * operand wgrad_split_kernel stages (amp_conv2d_wgrad_fmt x_split & 2). */
int amp_colsum_split(amp_ctx* ctx, const float* dy, int M, int N, float* scratch, float* out, int accumulate, float* dy_split, int shift) {
    AMP_REQUIRE(ctx && dy && scratch && out && dy_split && M >= 0 && N > 0 && N % 32 == 0 && shift >= 0 && shift <= 24, "amp_colsum_split: bad argument (N %% 32 != 0?)");
    const int parts = std::max(1, amp::cdiv(M, COLSUM_ROWS));
    hipLaunchKernelGGL(colsum_split_kernel, dim3(parts), dim3(256), 0, ctx->stream, dy, M, N, scratch, dy_split, ldexpf(1.0f, shift));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(amp::cdiv(N, 32)), dim3(256), 0, ctx->stream, scratch, parts, N, out, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* out[n] (= or +=) sum_m dy[m][n] for dy given as split rows of dy * 2^shift (N % 32 == 0); scratch: >= ceil(M/512)*N floats */
int amp_colsum_of_split(amp_ctx* ctx, const float* dy_split, int M, int N, float* scratch, float* out, int accumulate, int shift) {
    AMP_REQUIRE(ctx && dy_split && scratch && out && M >= 0 && N > 0 && N % 32 == 0 && shift >= 0 && shift <= 24, "amp_colsum_of_split: bad argument (N %% 32 != 0?)");
    const int parts = std::max(1, amp::cdiv(M, COLSUM_ROWS));
    hipLaunchKernelGGL(colsum_of_split_kernel, dim3(parts), dim3(256), 0, ctx->stream, dy_split, M, N, scratch, ldexpf(1.0f, -shift));
    hipLaunchKernelGGL(colsum_final_kernel, dim3(amp::cdiv(N, 32)), dim3(256), 0, ctx->stream, scratch, parts, N, out, accumulate);
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

int amp::wgrad_async_begin(amp_ctx* ctx, float* scratch1) {
    AMP_REQUIRE(ctx && scratch1 && !ctx->reduce_async, "wgrad_async_begin: bad argument or already on");
    if (!ctx->side) {
        AMP_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        for (int p = 0; p < 2; ++p) {
            AMP_HIP_CHECK(hipEventCreateWithFlags(&ctx->wg_ev[p], hipEventDisableTiming));
            AMP_HIP_CHECK(hipEventCreateWithFlags(&ctx->side_ev[p], hipEventDisableTiming));
        }
    }
    ctx->side_scratch1 = scratch1;
    ctx->side_used[0] = ctx->side_used[1] = false;
    ctx->side_parity = 0; ctx->side_last = -1;
    ctx->reduce_async = true;
    return AMP_OK;
}

int amp::wgrad_async_join(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "wgrad_async_join: null context");
    if (!ctx->reduce_async) return AMP_OK;
    // the side stream runs its reductions in order: waiting for both parities' last events covers everything issued
    for (int p = 0; p < 2; ++p)
        if (ctx->side_used[p]) AMP_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->side_ev[p], 0));
    return AMP_OK;
}

int amp::wgrad_async_end(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "wgrad_async_end: null context");
    const int st = amp::wgrad_async_join(ctx);
    ctx->reduce_async = false;
    return st;
}

// hits / misses / generation switches of the weight-gradient row-table cache of a context, and the bytes its tables occupy (tools/bench_trainer.py)
extern "C" int amp_debug_rowtab_stats(amp_ctx* ctx, unsigned long long* out4) {
    if (!ctx || !out4) return AMP_ERR_ARG;
    out4[0] = ctx->rowtab_hits; out4[1] = ctx->rowtab_misses; out4[2] = ctx->rowtab_flushes;
    out4[3] = (unsigned long long)(ctx->rowtab_used[0] + ctx->rowtab_used[1]) * 4ull;
    return AMP_OK;
}
