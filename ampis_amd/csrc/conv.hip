// Implicit-GEMM convolution, fp32 in / fp32 accumulate on the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN convolutions detectron2 runs for AMPIS:
// ResNet-50 stem+res2..res5, FPN lateral/output convs, the RPN head, the box-head FCs (as 1x1 convs over
// "pixels" = RoIs) and the mask head (3x3 convs, ConvTranspose 2x2 s2, 1x1 predictor).
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin, activations NHWC so a K-slice of 32 input
// channels of one tap is 128 contiguous bytes.  Block tile BM x BN x 32, 4 waves (2x2), each wave a
// (BM/2)x(BN/2) tile of 32x32 MFMA accumulators.  Both operand tiles sit in LDS as [row][k] with k contiguous
// and rows padded to 36 floats, so a lane fetches 4 consecutive k with one conflict-free ds_read_b128 and
// feeds 4 MFMAs from it (the k order inside a tile is permuted identically for A and B, which a sum over k
// does not see).  Register-staged double buffering: the global loads of K-step s+1 are issued before the MFMAs
// of step s and written to the other LDS buffer after them; one barrier per step.
//
// Epilogue (fused): y = acc*scale[n] + shift[n] (FrozenBN affine or conv bias), optional residual (same shape,
// or nearest-x2-upsampled for the FPN top-down path), optional ReLU, optional ConvTranspose2x2 scatter.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr float LO_SCALE = 2048.0f;      // AMP_CONV_F16X3: lo' = (x - hi) * 2^11 (see conv_f16x3_kernel)

constexpr int BK = 32;      // K-slice per step (floats)
constexpr int LDS_LD = 36;  // padded row length in LDS (floats): 144 B rows -> conflict-free ds_read_b128

struct ConvArgs {
    const float* x;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    const float* mask;   // optional, indexed like y: y = mask > 0 ? y : 0 (ReLU backward of the tensor the gradient belongs to)
    float* y;
    int B, H, W, Cin;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad;
    int M, K, nsteps;
    int relu, res_mode, out_mode;
    int ntn;  // number of N tiles
    float in_scale, out_scale;   // f16x3 kernels: activations are multiplied by in_scale (a power of two) before the split, the sum by out_scale
    int y_split;            // 1: y is written in the split hi|lo' row format (out_mode 0, Cout % 32 == 0): the next conv's operand
    // out_mode 3 (conv_split_kernel<128,256> only): the ConvTranspose 2x2 of the mask head with its consumers fused into the epilogue --
    // ReLU, the 1x1 predictor row of the detection's class, sigmoid -- so the [N,28,28,256] tensor never exists (see conv_epilogue_predict)
    const float* pred_w;    // [K][C2] predictor weights (fp32), C2 = Cout / 4 = 256
    const float* pred_b;    // [K]
    const int* pred_cls;    // [B] class of each RoI (row b of the input)
    int pred_K;
    // out_mode 4 (conv_split_kernel<128,256> only): the RPN head's tail -- ReLU, then the 16 predictor rows (3 objectness + 12 deltas + pad) as a
    // second f16x3 product in the epilogue (conv_epilogue_rpn); the 256-channel hidden tensor is never written
    const float* rpn_w;     // [16][Cout] split rows
    const float* rpn_b;     // [16]
    float* rpn_pred;        // [M][16]
    float* prob;            // [B][2Ho][2Wo] mask probabilities
    int res_split;          // 1: res is in that format too (decoded in the epilogue: hi + lo' * 2^-11, exact in fp32)
    int mask_split;         // 1: the ReLU mask tensor (training: the forward activation) is in that format (AMP_FMT_MASK_SPLIT)
    int direct_epi;         // conv_split_kernel: split rows written straight from the accumulators (0: staged through LDS; EXPERIMENT switch AMP_DIRECT_EPI)
    int stagger;            // conv_split_kernel: the two halves of the workgroup ping-pong between loading and multiplying (0: lockstep; EXPERIMENT switch AMP_STAGGER)
    int* range_flag;        // f16x3 kernels: set to 1 when an accumulator is not finite (operand beyond fp16 range)
    int cin_win, grouped;   // K runs over KH*KW*cin_win input channels; grouped: the window of N-tile n0 starts at channel n0
    int nblk;
    int korder;             // conv_split_kernel: 0 = K runs tap-major (ky, kx, then the 32-channel slices), 1 = channel-major (slice, then the KH*KW taps:
                            // the nine taps of a slice re-read the same cache lines within nine K-steps, while they are still in the XCD's L2)
    unsigned int div_howo_mul, div_wo_mul;   // x / d == (x * mul) >> shr for every x < 2^29 (Granlund-Montgomery, mul = ceil(2^shr / d))
    int div_howo_shr, div_wo_shr;
};

__device__ __forceinline__ unsigned int fastdiv(unsigned int x, unsigned int mul, int shr) {
    return (unsigned int)(((unsigned long long)x * mul) >> shr);
}

// Epilogue shared by both kernels: accumulators -> LDS (per-wave tile, [m][n]) -> row-wise float4 residual loads + stores.
// The MFMA C layout puts one n per lane and 16 m in registers: stored directly that is 64 dword stores per lane and,
// with a residual, 64 dependent dword loads.  Transposed through LDS every lane owns 4 consecutive n of one row per step:
// 16-B coalesced residual loads (all issued before the first store) and 16-B stores.  Callers pass the wave's tile origin
// (mw0, nw0) and must have synchronised the workgroup after the last operand reads.
template <int WTM, int WTN>
__device__ __forceinline__ void conv_epilogue_generic_rows(const ConvArgs& a, float* stage, int lane, int mw0, int nw0, int HoWo);

template <int WTM, int WTN, int MT, int NT, bool CHECK = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], float* lds, int wave, int lane,
                                              int mw0, int nw0, int HoWo) {
    if (CHECK) {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) bad = bad || !(fabsf(acc[i][j][e]) <= 3.0e38f);
        if (bad) atomicOr(a.range_flag, 1);
    }
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int SLD = WTN + 4;                       // padded row of the staging tile (floats)
    float* stage = lds + wave * (WTM * SLD);
    // (the loop's last __syncthreads() already ordered every wave's operand reads before these writes)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * SLD + j * 32 + l31] = acc[i][j][e];
    conv_epilogue_generic_rows<WTM, WTN>(a, stage, lane, mw0, nw0, HoWo);
}

// the same from MB x NB blocks of 16x16 MFMA accumulators
template <int WTM, int WTN, int MB, int NB, bool CHECK = false>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs& a, f32x4 (&acc)[MB][NB], float* lds, int wave, int lane,
                                                int mw0, int nw0, int HoWo) {
    if (CHECK) {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) bad = bad || !(fabsf(acc[i][j][e]) <= 3.0e38f);
        if (bad) atomicOr(a.range_flag, 1);
    }
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int SLD = WTN + 4;
    float* stage = lds + wave * (WTM * SLD);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                stage[(i * 16 + 4 * lq + e) * SLD + j * 16 + l15] = acc[i][j][e];
    conv_epilogue_generic_rows<WTM, WTN>(a, stage, lane, mw0, nw0, HoWo);
}

template <int WTM, int WTN>
__device__ __forceinline__ void conv_epilogue_generic_rows(const ConvArgs& a, float* stage, int lane, int mw0, int nw0, int HoWo) {
    constexpr int SLD = WTN + 4;                       // padded row of the staging tile (floats)
    constexpr int F4R = WTN / 4;                       // float4 per row
    constexpr int RPI = 64 / F4R;                      // rows per iteration (one wave)
    constexpr int NIT = WTM / RPI;

    const int erow = lane / F4R;                       // row within an iteration
    const int ec4 = lane % F4R;
    const int n = nw0 + ec4 * 4;
    const bool vec = (a.Cout & 3) == 0;                // rows of y / res are 16-B aligned
    const bool nv = n < a.Cout;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (n + q < a.Cout) {
            if (a.scale) sc[q] = a.scale[n + q];
            if (a.shift) sh[q] = a.shift[n + q];
        }
    }
    const int C2 = a.Cout >> 2;                        // deconv scatter only

    size_t yoff[NIT];
    f32x4 rres[NIT];
    bool mv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int m = mw0 + it * RPI + erow;
        mv[it] = nv && m < a.M;
        yoff[it] = (size_t)m * a.Cout + n;
        rres[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!mv[it]) continue;
        if (a.res_mode != 0 || a.out_mode != 0) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            size_t roff = yoff[it];
            if (a.res_mode == 2)
                roff = ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n;
            if (a.res_mode != 0) {
                if (vec) {
                    rres[it] = *reinterpret_cast<const f32x4*>(a.res + roff);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < a.Cout) rres[it][q] = a.res[roff + q];
                }
            }
            if (a.out_mode == 1) {
                const int kk = n / C2;
                const int co = n - kk * C2;
                const int oy2 = 2 * oy + (kk >> 1), ox2 = 2 * ox + (kk & 1);
                yoff[it] = ((size_t)(b * 2 * a.Ho + oy2) * (2 * a.Wo) + ox2) * C2 + co;
            } else if (a.out_mode == 2) {   // stride-2 scatter into a [B,2Ho,2Wo,Cout] tensor (data gradient of a strided 1x1 conv)
                yoff[it] = ((size_t)(b * 2 * a.Ho + 2 * oy) * (2 * a.Wo) + 2 * ox) * a.Cout + n;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                   // staging writes of this wave precede its reads (same-wave LDS order)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (it * RPI + erow) * SLD + ec4 * 4);
        if (!mv[it]) continue;
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t = __fadd_rn(__fmul_rn(v[q], sc[q]), sh[q]);
            if (a.res_mode != 0) t = __fadd_rn(t, rres[it][q]);
            if (a.relu) t = fmaxf(t, 0.f);
            o[q] = t;
        }
        if (a.mask && a.mask_split) {                  // the gating activation in the split row format (Cout % 32 == 0, out_mode 0)
            const char* mb = reinterpret_cast<const char*>(a.mask + (yoff[it] - (size_t)n)) + (n >> 5) * 128 + (n & 31) * 2;
            const f16x4 mh = *reinterpret_cast<const f16x4*>(mb);
            const f16x4 ml = *reinterpret_cast<const f16x4*>(mb + 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = __fadd_rn((float)mh[q], __fmul_rn((float)ml[q], 1.0f / LO_SCALE)) > 0.f ? o[q] : 0.f;
        } else if (a.mask) {
            if (vec) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(a.mask + yoff[it]);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = mk[q] > 0.f ? o[q] : 0.f;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < a.Cout) o[q] = a.mask[yoff[it] + q] > 0.f ? o[q] : 0.f;
            }
        }
        if (vec) {
            *reinterpret_cast<f32x4*>(a.y + yoff[it]) = o;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n + q < a.Cout) a.y[yoff[it] + q] = o[q];
        }
    }
}

// One K-step (32 k) of the AMP_CONV_F16X3 product on a wave tile of MB x NB blocks of 16 x 16, operands in the LDS row format
// (per row 64 B of hi halves, 64 B of lo' halves, 16-B chunks XOR-swizzled by (row >> 1) & 7).  v_mfma_f32_16x16x32_f16: lane =
// (row l15 of a block, k group lq), so the hi halves of k 8lq.. are chunk lq of the row and their lo' halves chunk 4 + lq: one
// ds_read_b128 each, conflict-free over the 16 rows of a lane group.  Every AMP_CONV_F16X3 forward kernel computes through this
// function, whoever staged the tiles: that is what makes "split in the kernel" and "split by the producer" the same arithmetic.
// (Shape: the 16x16x32 MFMA does the flops of the 32x32x16 one in the same cycles with the same LDS bytes, but on random operands
// the chip holds a higher clock under it -- MI355X_MICROARCH.md -- measured here: +2..9 % on the 3x3 / fc layers.)
template <int MB, int NB>
struct F16x3Frags {
    f16x8 ah[MB], al[MB], bh[NB], bl[NB];
};
template <int MB, int NB>
__device__ __forceinline__ void f16x3_load16(const float* As, const float* Bs, int fo_hi, int fo_lo, F16x3Frags<MB, NB>& f) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        f.ah[i] = *reinterpret_cast<const f16x8*>(As + i * 16 * BK + fo_hi);
        f.al[i] = *reinterpret_cast<const f16x8*>(As + i * 16 * BK + fo_lo);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        f.bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo_hi);
        f.bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo_lo);
    }
}
// SWAP = true: the weights are the MFMA's A operand (rows) and the activations its B operand (columns): the same exact products
// summed in the same order into acc[i][j] / acx[i][j] (i: 16 tile rows m, j: 16 channels n), but a lane now holds, per block, FOUR
// CHANNELS (4 lq + e) of ONE tile row (l15) instead of four rows of one channel -- what conv_epilogue_direct needs.
template <int MB, int NB, bool SWAP = false>
__device__ __forceinline__ void f16x3_mfma16(const F16x3Frags<MB, NB>& f, f32x4 (&acc)[MB][NB], f32x4 (&acx)[MB][NB]) {
    // per accumulator the order is lo'*hi, hi*lo' (cross sums), hi*hi; consecutive MFMAs never share an accumulator (NB apart)
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        if (SWAP) {
#pragma unroll
            for (int j = 0; j < NB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bh[j], f.al[i], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bl[j], f.ah[i], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.al[i], f.bh[j], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acx[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[i], f.bl[j], acx[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
        }
    }
}
template <int MB, int NB>
__device__ __forceinline__ void f16x3_step16(const float* As, const float* Bs, int fo_hi, int fo_lo, f32x4 (&acc)[MB][NB], f32x4 (&acx)[MB][NB]) {
    F16x3Frags<MB, NB> f;
    f16x3_load16<MB, NB>(As, Bs, fo_hi, fo_lo, f);
    f16x3_mfma16<MB, NB>(f, acc, acx);
}

// Fast epilogue for the layouts the model actually uses (Cout % 4 == 0): same LDS transposition and the same arithmetic as
// conv_epilogue, but straight-line: the row offsets advance by additions (SPATIAL = false: out_mode 0, res_mode 0/1) or come from
// two multiply-shift divisions per row (SPATIAL = true: upsampled residual, deconv / stride-2 scatter); residual and mask loads of
// all rows are issued before the first store; a block-uniform `full` flag removes the per-row bounds predicates from every tile
// but the last one.  (Measured on gfx950: the generic epilogue's per-row branches cost 13-55 % of the short-K layers.)
template <int WTM, int WTN, bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, float* stage, int lane, int mw0, int nw0);
template <int WTM, int WTN, bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_rows8(const ConvArgs& a, float* stage, int lane, int mw0, int nw0);

template <int WTM, int WTN, int MT, int NT, bool SPATIAL, bool CHECK = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvArgs& a, f32x16 (&acc)[MT][NT], float* lds, int wave, int lane,
                                                   int mw0, int nw0) {
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int SLD = WTN + 4;
    float* stage = lds + wave * (WTM * SLD);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * SLD + j * 32 + l31] = acc[i][j][e];
    if (a.y_split && !a.mask && a.out_mode == 0) conv_epilogue_rows8<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
    else conv_epilogue_rows<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
}

// the same from 16x16 MFMA accumulators (v_mfma_f32_16x16x32_f16: lane = column, 4 consecutive rows per lane group of 16)
template <int WTM, int WTN, int MT, int NT, bool SPATIAL, bool CHECK = false>
__device__ __forceinline__ void conv_epilogue_fast16(const ConvArgs& a, f32x4 (&acc)[MT][NT], float* lds, int wave, int lane,
                                                     int mw0, int nw0) {
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int SLD = WTN + 4;
    float* stage = lds + wave * (WTM * SLD);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                stage[(i * 16 + 4 * lq + e) * SLD + j * 16 + l15] = acc[i][j][e];
    if (a.y_split && !a.mask && a.out_mode == 0) conv_epilogue_rows8<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
    else conv_epilogue_rows<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
}

// Split-format output (the trunk's native activation format), 8 channels per lane: a lane's share of an output row is 16 B of hi
// halves and 16 B of lo' halves (and the same of a split residual), so every global access is a full dwordx4 -- with 4 channels per
// lane (conv_epilogue_rows) they were 8-B accesses and the memory-bound 1x1 layers ran 10 % slower than with fp32 rows.
// out_mode 0, no mask (inference), Cout % 32 == 0; same arithmetic, same order as conv_epilogue_rows.
template <int WTM, int WTN, bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_rows8(const ConvArgs& a, float* stage, int lane, int mw0, int nw0) {
    constexpr int SLD = WTN + 4;
    constexpr int L8R = WTN / 8;                       // lanes per row
    constexpr int RPI = 64 / L8R;                      // rows per iteration
    constexpr int NIT = WTM / RPI;
    const int erow = lane / L8R;
    const int ec8 = lane % L8R;
    const int n = nw0 + ec8 * 8;
    const bool nv = n < a.Cout;                        // Cout % 32 == 0
    float sc[8], sh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sc[q] = (nv && a.scale) ? a.scale[n + q] : 1.f;
        sh[q] = (nv && a.shift) ? a.shift[n + q] : 0.f;
    }
    const bool has_res = a.res_mode != 0;
    const size_t col_b = (size_t)(n >> 5) * 128 + (size_t)(n & 31) * 2;   // byte offset of this lane's hi halves inside a split row
#pragma unroll
    for (int h = 0; h < 2; ++h) {                      // two passes of NIT / 2 rows: bounds the live registers
        constexpr int HN = NIT / 2;
        size_t yrow[HN], rrow[HN];                     // float index of the row start in y / res
        bool mv[HN];
        f32x4 r0[HN], r1[HN];
#pragma unroll
        for (int t = 0; t < HN; ++t) {
            const int it = h * HN + t;
            const int mrow = mw0 + it * RPI + erow;
            mv[t] = nv && mrow < a.M;
            const unsigned int m = min((unsigned int)mrow, (unsigned int)(a.M - 1));
            yrow[t] = (size_t)m * a.Cout;
            rrow[t] = yrow[t];
            if (SPATIAL && a.res_mode == 2) {
                const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
                const unsigned int rem = m - b * (unsigned int)(a.Ho * a.Wo);
                const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
                const unsigned int ox = rem - oy * (unsigned int)a.Wo;
                rrow[t] = ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout;
            }
        }
        if (has_res) {
#pragma unroll
            for (int t = 0; t < HN; ++t) {
                if (!mv[t]) continue;
                if (a.res_split) {
                    const char* rb = reinterpret_cast<const char*>(a.res + rrow[t]) + col_b;
                    r0[t] = *reinterpret_cast<const f32x4*>(rb);          // 8 hi halves
                    r1[t] = *reinterpret_cast<const f32x4*>(rb + 64);     // 8 lo' halves
                } else {
                    r0[t] = *reinterpret_cast<const f32x4*>(a.res + rrow[t] + n);
                    r1[t] = *reinterpret_cast<const f32x4*>(a.res + rrow[t] + n + 4);
                }
            }
        }
        if (h == 0) __builtin_amdgcn_wave_barrier();   // staging writes of this wave precede its reads (same-wave LDS order)
        bool bad = false;
#pragma unroll
        for (int t = 0; t < HN; ++t) {
            const int it = h * HN + t;
            const float* sp = stage + (it * RPI + erow) * SLD + ec8 * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp), v1 = *reinterpret_cast<const f32x4*>(sp + 4);
            f16x8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = q < 4 ? v0[q] : v1[q - 4];
                if (CHECK) bad = bad || !(fabsf(v) <= 3.0e38f);
                float o = __fadd_rn(__fmul_rn(v, sc[q]), sh[q]);
                if (has_res) {
                    float r;
                    if (a.res_split) {
                        const f16x8 rh = __builtin_bit_cast(f16x8, r0[t]), rl = __builtin_bit_cast(f16x8, r1[t]);
                        r = __fadd_rn((float)rh[q], __fmul_rn((float)rl[q], 1.0f / LO_SCALE));
                    } else {
                        r = q < 4 ? r0[t][q] : r1[t][q - 4];
                    }
                    o = __fadd_rn(o, r);
                }
                if (a.relu) o = fmaxf(o, 0.f);
                const _Float16 hh = (_Float16)o;
                hi[q] = hh;
                lo[q] = (_Float16)((o - (float)hh) * LO_SCALE);
            }
            if (mv[t]) {
                char* base = reinterpret_cast<char*>(a.y + yrow[t]) + col_b;
                *reinterpret_cast<f16x8*>(base) = hi;
                *reinterpret_cast<f16x8*>(base + 64) = lo;
            }
        }
        if (CHECK && bad) atomicOr(a.range_flag, 1);
    }
}

// ---- epilogues of conv_split_kernel (role-swapped MFMA, see f16x3_mfma16<.., SWAP = true>) ---------------------------------------
// Channel order inside a wave's 64 channels: LDS weight row w = 16 j + 4 q + e (block j, MFMA row 4 q + e) holds channel
// swap_channel(w) = 32 (j >> 1) + 8 q + 4 (j & 1) + e, so that lane (l15, lq = q) owns, per tile row, channels [8 lq, 8 lq + 8) of
// each of the two 32-channel groups: 16 B of hi halves and 16 B of lo' halves per group in the split row format.
__device__ __forceinline__ int swap_channel(int w) {
    const int j = w >> 4, q = (w >> 2) & 3, e = w & 3;
    return 32 * (j >> 1) + 8 * q + 4 * (j & 1) + e;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Straight from the accumulators to split rows -- no LDS staging (the staged epilogue moves 128 KB per tile through ds_write_b32 at
// 64 B/clk: ~2000 cycles, plus the read-back) and packed arithmetic: per pair of outputs v_pk_mul/add (affine), 2 v_max (ReLU),
// v_cvt_pk_f16_f32 (hi), v_pk_mul (x 2^11), v_pk_fma (hi * -2^11 + o * 2^11: the exact difference, already scaled), v_cvt_pk (lo').
// Same values as conv_epilogue_rows8, bit for bit: o - hi and the scaling by 2^11 are exact in fp32, so the fused form rounds once,
// where the unfused one did not round at all.  Range check: s = fma(v, 0, s) stays 0 unless an accumulator is inf or NaN.
// out_mode 0, split output; residual none or in the split format (res_mode 1: same row; 2: the coarser level's row); ReLU mask
// (training: the data gradient gated by the forward activation) none or in the split format.
// out_mode 1 (SPATIAL instantiation; round 4): the ConvTranspose 2x2 s2 of the mask head as a 1x1 conv to (tap, co) = 4 x C2 channels -- a wave's
// 64 columns lie inside ONE tap (C2 % 64 == 0), tile row m = input pixel (b, oy, ox) goes to output pixel (b, 2 oy + ky, 2 ox + kx), a row of C2
// channels: the same 16-B stores at another row address, so the 1.6-GB tensor of a training step no longer takes the staged epilogue's trip
// through LDS (1048 us at 1.96 TB/s before).
// MBR row blocks of 16 per wave tile (4: the 64 x 64 wave tiles of conv_split_kernel; 2: conv3x3_c64_kernel); mrow_in[i] + (nothing): the GEMM row of
// this lane in row block i (any value >= a.M: not stored) -- consecutive rows for the GEMM tiles, rows of a 2-D pixel patch for the patch kernel
// AFFINE_LDS: scale / shift come from LDS copies indexed by the channel relative to nrel0 (conv1x1_nloop_kernel: a global load between two K-steps
// would wait for the ring's prefetched tiles -- vector-memory operations retire in order)
// PLAIN: the layer has neither residual nor mask, known at compile time -- behind a RUN-TIME "has residual" branch the compiler waits for the branch's
// loads with vmcnt(0) at the join, on both paths, which inside conv1x1_nloop_kernel's loop would drain the ring at every tile
template <bool SPATIAL, bool CHECK, int MBR, bool AFFINE_LDS = false, bool PLAIN = false>
__device__ __forceinline__ void conv_epilogue_direct_rows(const ConvArgs& a, f32x4 (&acc)[MBR][4], int lane, const int (&mrow_in)[MBR], int nw0,
                                                          const float* lsc = nullptr, const float* lsh = nullptr, int nrel0 = 0) {
    const int lq = lane >> 4;
    const bool has_res = !PLAIN && a.res_mode != 0, has_mask = !PLAIN && a.mask != nullptr;
    const bool deconv = SPATIAL && a.out_mode == 1;
    const int C2 = a.Cout >> 2;
    const int tap = deconv ? nw0 / C2 : 0, ncol0 = deconv ? nw0 - tap * C2 : nw0;      // column of this wave inside an output row
    const int yld = deconv ? C2 : a.Cout;                                              // floats per output row
    f32x2 sc[2][4], sh[2][4];
    size_t colb[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int n = nw0 + 32 * g + 8 * lq;
        const int nc = ncol0 + 32 * g + 8 * lq;
        colb[g] = (size_t)(nc >> 5) * 128 + (size_t)(nc & 31) * 2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if constexpr (AFFINE_LDS) {
                const int nr = nrel0 + 32 * g + 8 * lq + 2 * p;
                sc[g][p] = f32x2{lsc[nr], lsc[nr + 1]};
                sh[g][p] = f32x2{lsh[nr], lsh[nr + 1]};
            } else {
                sc[g][p] = a.scale ? f32x2{a.scale[n + 2 * p], a.scale[n + 2 * p + 1]} : f32x2{1.f, 1.f};
                sh[g][p] = a.shift ? f32x2{a.shift[n + 2 * p], a.shift[n + 2 * p + 1]} : f32x2{0.f, 0.f};
            }
        }
    }
    f32x2 chk = {0.f, 0.f};
    // The residual rows of ALL four row groups are requested before the first store: y and res may alias as far as the compiler knows, so
    // it keeps a group's loads behind the previous group's stores -- and loads return in order BEHIND older stores (one vmcnt queue), so
    // every group paid a store round trip plus a load round trip.  The 64 registers are the cross-term accumulators', dead since the fold.
    f16x8 rha[MBR][2], rla[MBR][2];
#ifdef AMP_NO_HOIST
    constexpr bool HOIST = false;
#else
    constexpr bool HOIST = true;
#endif
    auto load_res = [&](int i) {
        {
            const unsigned int m = min((unsigned int)mrow_in[i], (unsigned int)(a.M - 1));
            size_t rrow = (size_t)m * a.Cout;
            if (SPATIAL && a.res_mode == 2) {
                const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
                const unsigned int rem = m - b * (unsigned int)(a.Ho * a.Wo);
                const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
                const unsigned int ox = rem - oy * (unsigned int)a.Wo;
                rrow = ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const char* rb = reinterpret_cast<const char*>(a.res + rrow) + colb[g];
                rha[i][g] = *reinterpret_cast<const f16x8*>(rb);
                rla[i][g] = *reinterpret_cast<const f16x8*>(rb + 64);
            }
        }
    };
    if (has_res && HOIST) {
#pragma unroll
        for (int i = 0; i < MBR; ++i) load_res(i);
    }
    // the same for the gating activation (training: data gradients): all four row groups up front
    f16x8 mha[MBR][2], mla[MBR][2];
    auto load_mask = [&](int i) {
        const unsigned int m = min((unsigned int)mrow_in[i], (unsigned int)(a.M - 1));
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const char* mb = reinterpret_cast<const char*>(a.mask + (size_t)m * a.Cout) + colb[g];
            mha[i][g] = *reinterpret_cast<const f16x8*>(mb);
            mla[i][g] = *reinterpret_cast<const f16x8*>(mb + 64);
        }
    };
    if (has_mask && HOIST) {
#pragma unroll
        for (int i = 0; i < MBR; ++i) load_mask(i);
    }
#pragma unroll
    for (int i = 0; i < MBR; ++i) {
        const int mrow = mrow_in[i];
        const bool mv = mrow < a.M;
        const unsigned int m = min((unsigned int)mrow, (unsigned int)(a.M - 1));
        size_t yrow = (size_t)m * a.Cout;
        if (deconv) {
            const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            yrow = (((size_t)b * 2 * a.Ho + 2 * oy + (tap >> 1)) * (2 * a.Wo) + 2 * ox + (tap & 1)) * (size_t)yld;
        }
        f16x8 rh[2], rl[2], mh[2], ml[2];
        if (has_res) {
            if (!HOIST) load_res(i);
#pragma unroll
            for (int g = 0; g < 2; ++g) { rh[g] = rha[i][g]; rl[g] = rla[i][g]; }
        }
        if (has_mask) {                                  // the gating activation (training), split rows indexed like y
            if (!HOIST) load_mask(i);
#pragma unroll
            for (int g = 0; g < 2; ++g) { mh[g] = mha[i][g]; ml[g] = mla[i][g]; }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f16x8 hi, lo;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4& blk = acc[i][2 * g + (p >> 1)];
                const f32x2 v = {blk[2 * (p & 1)], blk[2 * (p & 1) + 1]};
                if (CHECK) chk = __builtin_elementwise_fma(v, f32x2{0.f, 0.f}, chk);
                f32x2 o = v * sc[g][p] + sh[g][p];
                if (has_res) {
                    const f32x2 r1 = {(float)rh[g][2 * p], (float)rh[g][2 * p + 1]};
                    const f32x2 r2 = {(float)rl[g][2 * p], (float)rl[g][2 * p + 1]};
                    o = o + (r1 + r2 * (1.0f / LO_SCALE));
                }
                if (a.relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
                if (has_mask) {
                    const f32x2 mk = f32x2{(float)mh[g][2 * p], (float)mh[g][2 * p + 1]} + f32x2{(float)ml[g][2 * p], (float)ml[g][2 * p + 1]} * (1.0f / LO_SCALE);
                    o[0] = mk[0] > 0.f ? o[0] : 0.f;
                    o[1] = mk[1] > 0.f ? o[1] : 0.f;
                }
                const f16x2 h = __builtin_convertvector(o, f16x2);
                const f32x2 hf = {(float)h[0], (float)h[1]};
                const f32x2 l = __builtin_elementwise_fma(hf, f32x2{-LO_SCALE, -LO_SCALE}, o * LO_SCALE);
                const f16x2 lh = __builtin_convertvector(l, f16x2);
                hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
                lo[2 * p] = lh[0]; lo[2 * p + 1] = lh[1];
            }
            if (mv) {
                char* base = reinterpret_cast<char*>(a.y + yrow) + colb[g];
                *reinterpret_cast<f16x8*>(base) = hi;
                *reinterpret_cast<f16x8*>(base + 64) = lo;
            }
        }
    }
    if (CHECK && (!(chk[0] == 0.f) || !(chk[1] == 0.f))) atomicOr(a.range_flag, 1);
}

template <bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_direct(const ConvArgs& a, f32x4 (&acc)[4][4], int lane, int mw0, int nw0) {
    const int l15 = lane & 15;
    const int mrows[4] = {mw0 + l15, mw0 + 16 + l15, mw0 + 32 + l15, mw0 + 48 + l15};
    conv_epilogue_direct_rows<SPATIAL, CHECK, 4>(a, acc, lane, mrows, nw0);
}

// The staged epilogues (fp32 output, mask, scatter modes) behind the role-swapped MFMA: only the staging indices differ.
template <int WTM, int WTN, bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_swapped(const ConvArgs& a, f32x4 (&acc)[4][4], float* lds, int wave, int lane, int mw0, int nw0) {
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int SLD = WTN + 4;
    float* stage = lds + wave * (WTM * SLD);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(stage + (i * 16 + l15) * SLD + 32 * (j >> 1) + 8 * lq + 4 * (j & 1)) = acc[i][j];
    if (a.y_split && !a.mask && a.out_mode == 0) conv_epilogue_rows8<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
    else conv_epilogue_rows<WTM, WTN, SPATIAL, CHECK>(a, stage, lane, mw0, nw0);
}

// rows of a wave's staged [WTM][WTN] tile -> affine, residual, ReLU, mask -> y (fp32 or split rows)
template <int WTM, int WTN, bool SPATIAL, bool CHECK>
__device__ __forceinline__ void conv_epilogue_rows(const ConvArgs& a, float* stage, int lane, int mw0, int nw0) {
    constexpr int SLD = WTN + 4;
    constexpr int F4R = WTN / 4;
    constexpr int RPI = 64 / F4R;
    constexpr int NIT = WTM / RPI;

    const int erow = lane / F4R;
    const int ec4 = lane % F4R;
    const int n = nw0 + ec4 * 4;
    const bool nv = n < a.Cout;                        // Cout % 4 == 0: the whole float4 is in range or none of it
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nv) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (a.scale) sc[q] = a.scale[n + q];
            if (a.shift) sh[q] = a.shift[n + q];
        }
    }
    const bool full = __builtin_amdgcn_readfirstlane(mw0) + WTM <= a.M && __builtin_amdgcn_readfirstlane(nw0) + WTN <= a.Cout;
    const bool has_res = a.res_mode != 0, has_mask = a.mask != nullptr;

    size_t yoff[NIT], roff[NIT];
    if (!SPATIAL) {
        const size_t step = (size_t)RPI * a.Cout;
        yoff[0] = (size_t)(mw0 + erow) * a.Cout + n;
#pragma unroll
        for (int it = 1; it < NIT; ++it) yoff[it] = yoff[it - 1] + step;
#pragma unroll
        for (int it = 0; it < NIT; ++it) roff[it] = yoff[it];
    } else {
        const int C2 = a.Cout >> 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned int m = min((unsigned int)(mw0 + it * RPI + erow), (unsigned int)(a.M - 1));   // clamped rows are never stored
            const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            roff[it] = (a.res_mode == 2) ? ((size_t)(b * (a.Ho >> 1) + (oy >> 1)) * (a.Wo >> 1) + (ox >> 1)) * a.Cout + n
                                         : (size_t)m * a.Cout + n;
            if (a.out_mode == 1) {
                const int kk = n / C2;
                const int co = n - kk * C2;
                yoff[it] = ((size_t)(b * 2 * a.Ho + 2 * oy + (kk >> 1)) * (2 * a.Wo) + 2 * ox + (kk & 1)) * C2 + co;
            } else if (a.out_mode == 2) {
                yoff[it] = ((size_t)(b * 2 * a.Ho + 2 * oy) * (2 * a.Wo) + 2 * ox) * a.Cout + n;
            } else {
                yoff[it] = (size_t)m * a.Cout + n;
            }
        }
    }

    auto body = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        bool mv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) mv[it] = FULL || (nv && mw0 + it * RPI + erow < a.M);
        f32x4 rres[NIT], mk[NIT];
        if (has_res && a.res_split) {
            // the residual tensor lives in the split row format: this lane's 4 channels are 8 B of hi halves and 8 B of lo' halves
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (mv[it]) {
                    const char* rb = reinterpret_cast<const char*>(a.res + (roff[it] - (size_t)n)) + (n >> 5) * 128 + (n & 31) * 2;
                    const f16x4 rh = *reinterpret_cast<const f16x4*>(rb);
                    const f16x4 rl = *reinterpret_cast<const f16x4*>(rb + 64);
#pragma unroll
                    for (int q = 0; q < 4; ++q) rres[it][q] = __fadd_rn((float)rh[q], __fmul_rn((float)rl[q], 1.0f / LO_SCALE));
                }
        } else if (has_res) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (mv[it]) rres[it] = *reinterpret_cast<const f32x4*>(a.res + roff[it]);
        }
        if (has_mask && a.mask_split) {
            // the activation whose sign gates the gradient lives in the split row format (training on the native trunk): decoded like a
            // split residual; hi + lo' * 2^-11 > 0 exactly when the stored activation was (down to 2^-35, below which both halves are 0)
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (mv[it]) {
                    const char* mb = reinterpret_cast<const char*>(a.mask + (yoff[it] - (size_t)n)) + (n >> 5) * 128 + (n & 31) * 2;
                    const f16x4 mh = *reinterpret_cast<const f16x4*>(mb);
                    const f16x4 ml = *reinterpret_cast<const f16x4*>(mb + 64);
#pragma unroll
                    for (int q = 0; q < 4; ++q) mk[it][q] = __fadd_rn((float)mh[q], __fmul_rn((float)ml[q], 1.0f / LO_SCALE));
                }
        } else if (has_mask) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (mv[it]) mk[it] = *reinterpret_cast<const f32x4*>(a.mask + yoff[it]);
        }
        __builtin_amdgcn_wave_barrier();               // staging writes of this wave precede its reads (same-wave LDS order)
        bool bad = false;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (it * RPI + erow) * SLD + ec4 * 4);
            if (CHECK) bad = bad || !(fabsf(v[0]) <= 3.0e38f) || !(fabsf(v[1]) <= 3.0e38f) || !(fabsf(v[2]) <= 3.0e38f) || !(fabsf(v[3]) <= 3.0e38f);
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = __fadd_rn(__fmul_rn(v[q], sc[q]), sh[q]);
                if (has_res) t = __fadd_rn(t, rres[it][q]);
                if (a.relu) t = fmaxf(t, 0.f);
                if (has_mask) t = mk[it][q] > 0.f ? t : 0.f;
                o[q] = t;
            }
            if (a.y_split) {
                // row-relative: 32 channels = 64 B of hi halves + 64 B of lo' halves; this lane's 4 channels -> 8 B in each
                if (mv[it]) {
                    f16x4 hi, lo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const _Float16 h = (_Float16)o[q];
                        hi[q] = h;
                        lo[q] = (_Float16)((o[q] - (float)h) * LO_SCALE);
                    }
                    const size_t row = yoff[it] - (size_t)n;                     // float index of the row start
                    char* base = reinterpret_cast<char*>(a.y + row) + (n >> 5) * 128 + (n & 31) * 2;
                    *reinterpret_cast<f16x4*>(base) = hi;
                    *reinterpret_cast<f16x4*>(base + 64) = lo;
                }
            } else if (mv[it]) {
                *reinterpret_cast<f32x4*>(a.y + yoff[it]) = o;
            }
        }
        if (CHECK && bad) atomicOr(a.range_flag, 1);
    };
    if (full) body(std::true_type{});
    else body(std::false_type{});
}

// ABL (ablation, tools/bench_conv_ablate.py only; results wrong for ABL != 0): 1 = no global loads in the K loop,
// 2 = also no LDS writes / barriers, 3 = MFMA only (fragments read once).
template <int BM, int BN, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;  // float4 loads per thread per step
    constexpr int TILE_FLOATS = (BM + BN) * LDS_LD;

    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tile = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = tile % a.ntn;
    const int tile_m = tile / a.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread staging geometry: thread loads float4 #col4 of rows lrow + 32*r ----
    const int col4 = tid & 7;
    const int lrow = tid >> 3;

    int a_iy0[RA], a_ix0[RA], a_pb[RA];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int m = m0 + lrow + 32 * r;
        if (m < a.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            a_iy0[r] = oy * a.stride - a.pad;
            a_ix0[r] = ox * a.stride - a.pad;
            a_pb[r] = b * a.H * a.W;
        } else {
            a_iy0[r] = -(1 << 28);
            a_ix0[r] = 0;
            a_pb[r] = 0;
        }
    }

    f32x4 ra[RA], rb[RB];

    auto load_tiles = [&](int step) {
        const int kidx = step * BK + col4 * 4;
        const int t = kidx / a.Cin;
        const int c = kidx - t * a.Cin;
        const int ky = t / a.KW;
        const int kx = t - ky * a.KW;
        const bool kv = kidx < a.K;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int iy = a_iy0[r] + ky;
            const int ix = a_ix0[r] + kx;
            const bool v = kv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (v) {
                const size_t off = (size_t)(a_pb[r] + iy * a.W + ix) * a.Cin + c;
                val = *reinterpret_cast<const f32x4*>(a.x + off);
            }
            ra[r] = val;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int n = n0 + lrow + 32 * r;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (kv && n < a.Cout) {
                val = *reinterpret_cast<const f32x4*>(a.w + (size_t)n * a.K + kidx);
            }
            rb[r] = val;
        }
    };

    auto store_tiles = [&](int buf) {
        float* As = lds + buf * TILE_FLOATS;
        float* Bs = As + BM * LDS_LD;
#pragma unroll
        for (int r = 0; r < RA; ++r)
            *reinterpret_cast<f32x4*>(As + (lrow + 32 * r) * LDS_LD + col4 * 4) = ra[r];
#pragma unroll
        for (int r = 0; r < RB; ++r)
            *reinterpret_cast<f32x4*>(Bs + (lrow + 32 * r) * LDS_LD + col4 * 4) = rb[r];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    for (int step = 0; step < a.nsteps; ++step) {
        const int cur = step & 1;
        const bool more = step + 1 < a.nsteps;
        if (more && ABL == 0) load_tiles(step + 1);

        const float* As = lds + cur * TILE_FLOATS + (wm * WTM + l31) * LDS_LD + 4 * lh;
        const float* Bs = lds + cur * TILE_FLOATS + BM * LDS_LD + (wn * WTN + l31) * LDS_LD + 4 * lh;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[MT], bf[NT];
            if (ABL < 3 || step == 0) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDS_LD + 8 * q);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDS_LD + 8 * q);
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = f32x4{acc[i][0][0], 1.f, 2.f, 3.f};
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = f32x4{acc[0][j][1], 1.f, 2.f, 3.f};
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
        }

        if (ABL < 2) {
            if (more) store_tiles(cur ^ 1);
            __syncthreads();
        }
    }
    if (ABL >= 2) __syncthreads();

    static_assert(4 * (BM / 2) * (BN / 2 + 4) <= 2 * TILE_FLOATS, "staging tile must fit in the operand buffers");
    conv_epilogue<WTM, WTN, MT, NT>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN, HoWo);
}

// ------------------------------------------------------------------------------------------------------------------
// conv_glds_kernel: same tiling and epilogue, operand staging by LDS-DMA (buffer_load_dwordx4 ... lds), for Cin % 32 == 0
// (every layer but the stem).  What the experiments on this kernel family showed (tools/bench_conv_ablate.py and the
// timing variants recorded in DESIGN.md §4, 3x3 256->256 @256^2): MFMA-only floor 151.7 TF; LDS fragment reads and the
// per-step barrier are free (148-150 TF); register-staged operands with predicated global loads 120 TF; LDS-DMA with
// per-lane 64-bit pointers + zero-page selects 134 TF, of which ~3 % is the DMA instructions and ~9 % their address VALU
// (each VALU instruction takes an issue slot the MFMA stream wants).  Hence: buffer addressing.  A row's byte offset
// (voffset, 32-bit) is recomputed only when the tap (ky,kx) changes; the channel offset inside the tap and the K offset
// of the weights are block-uniform and ride in the scalar soffset; out-of-image taps and rows beyond M / Cout get an
// out-of-range voffset, for which the hardware bounds check writes zeros into LDS (probed on gfx950) -- no predication,
// no zero page, no per-step VALU.  One wave-instruction writes 8 rows x 128 B linearly, so the bank-conflict fix is an
// XOR swizzle applied to the per-lane SOURCE chunk and again on the fragment reads (chunk ^ ((row >> 1) & 7):
// conflict-free for the 16-lane groups of ds_read_b128).
// ------------------------------------------------------------------------------------------------------------------
constexpr unsigned int OOB_VOFF = 0x80000000u;   // >= num_records of every buffer this kernel accepts (< 2 GiB)

// STEM = true: the 7x7 stride-2 stem on the [B,H,W,4] input with weights [64][7][8][4]: a K-step is one kernel row ky, whose
// 8 taps x 4 channels are 128 contiguous bytes; the 16-B chunk index IS the tap kx, so validity is per chunk.
// EPI: 0 = generic epilogue (any Cout), 1 = fast (Cout % 4 == 0, out_mode 0, res_mode 0/1), 2 = fast with per-row (b,oy,ox).
// F16 = true: AMP_CONV_F16X3 with BOTH operands already in the split hi|lo' row format -- the input tensor was written that way by
// its producer (conv epilogue / RoIAlign with split output; byte offsets equal the fp32 tensor's) -- so activations and weights
// are staged by LDS-DMA with no per-step VALU, and the compute loop is conv_f16x3_kernel's.  BN = 256 runs 8 waves (2 x 4).
// G32 = true (grouped conv with <= 32 channels per group, BN = 64, F16): a tap is ONE K-step.  The 64-channel window of the N tile is
// staged as two planes of 32 channels; wave column wn multiplies plane wn only -- in the block-diagonal window layout the weights of
// outputs [32 wn, 32 wn + 32) are zero outside that plane, so the two-steps-per-tap form spends half its MFMAs on zeros.  The weight
// rows are read from the same split copy (group (n >> 5) & 1 of each tap's 64 channels).  Same sums in the same order, bit for bit.
template <int BN, bool STEM = false, int EPI = 0, bool F16 = false, bool G32 = false>
__global__ __launch_bounds__(BN == 256 ? 512 : 256, BN == 256 ? 1 : 2) void conv_glds_kernel(const ConvArgs a, const unsigned int x_bytes,
                                                                                               const unsigned int w_bytes) {
    static_assert(!G32 || (BN == 64 && F16 && !STEM), "G32 is the grouped form of the 64-wide f16x3 tile");
    constexpr int BM = 128;
    constexpr int WTM = BM / 2, WTN = (BN == 64) ? 32 : 64;
    constexpr int NWN = BN / WTN, NW = 2 * NWN;   // waves across N, waves per workgroup
    constexpr int MT = WTM / 32, NT = WTN / 32;
    constexpr int NPL = G32 ? 2 : 1;      // A planes per tile
    constexpr int GA = BM / NW / 8;       // DMA instructions per wave, step and plane for A: BM/NW rows per wave, 8 rows each
    constexpr int GB = BN / NW / 8;       // ... for B: BN/NW rows per wave
    constexpr int TILE_FLOATS = (NPL * BM + BN) * BK;
    constexpr int SLD = WTN + 4;
    constexpr int STAGE_FLOATS = NW * WTM * SLD;
    // RING (the 64-wide f16x3 tiles: res2's and res5's 3x3 layers, the grouped layers of ResNeXt): three LDS buffers, the DMA of tile s+2
    // issued when tile s starts computing, counted vmcnt and a bare barrier -- conv_split_kernel's ring without the ping-pong.  A K-step of
    // this tile is 384 cycles of MFMA per wave and an LDS-DMA request needs ~2300 from issue to landing: with one tile in flight per
    // workgroup (two workgroups per CU) the matrix pipe waited two thirds of the time (PMC MFMA busy 0.26); 3 x 24.5 KB still fits twice.
    constexpr bool RING = F16 && BN == 64 && !STEM && !G32;      // (G32: two A planes, 3 x 41 KB would leave one workgroup per CU)
    constexpr int NBUF = RING ? 3 : 2;
    constexpr int LDS_FLOATS = (NBUF * TILE_FLOATS > STAGE_FLOATS) ? NBUF * TILE_FLOATS : STAGE_FLOATS;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tile = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = tile % a.ntn;
    const int tile_m = tile / a.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int HoWo = a.Ho * a.Wo;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);

    // ---- staging geometry: instruction g of this wave fills rows wave*R + 8g + (lane>>3), 16-B position lane&7 ----
    const int srow = lane >> 3, spos = lane & 7;
    int a_iy0[GA], a_ix0[GA], a_pb[GA], a_chunk[GA];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        const int r = wave * (BM / NW) + 8 * g + srow;
        a_chunk[g] = 4 * (spos ^ ((r >> 1) & 7));      // source chunk (floats) = 4 * (pos ^ swz(row))
        const int m = m0 + r;
        if (m < a.M) {
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            a_iy0[g] = oy * a.stride - a.pad;
            a_ix0[g] = ox * a.stride - a.pad;
            a_pb[g] = b * a.H * a.W;
        } else {
            a_iy0[g] = -(1 << 28);
            a_ix0[g] = 0;
            a_pb[g] = 0;
        }
    }
    unsigned int b_voff[GB];              // byte offset of the weight row (+ swizzled chunk); the K offset rides in soffset
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * (BN / NW) + 8 * g + srow;
        const int n = n0 + r;
        b_voff[g] = (n < a.Cout) ? (unsigned int)(((size_t)n * a.K + (G32 ? 32 * ((n >> 5) & 1) : 0) + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
    unsigned int a_voff[GA];              // byte offset of the input pixel of the current tap (+ swizzled chunk), or OOB

    const int csteps = (STEM || G32) ? 1 : a.cin_win / BK;   // K-steps per tap
    const int nsteps = G32 ? a.nsteps / 2 : a.nsteps;        // (G32: both 32-channel halves of a tap's window in one step)
    const int a_win = a.grouped ? n0 * 4 : 0;   // bytes: first input channel of this N-tile's window (grouped conv)
    const int kw_taps = STEM ? 1 : a.KW;        // STEM: the 8 taps of a row travel inside one K-step
    int ky = 0, kx = 0, cs = 0;           // block-uniform tap state of the tile being STAGED
    int kstep = 0;                        // index of the tile being staged

    auto stage = [&](int buf) {
        if (cs == 0) {                    // new tap: recompute the row offsets (uniform branch, once per Cin/32 steps)
#pragma unroll
            for (int g = 0; g < GA; ++g) {
                const int iy = a_iy0[g] + ky;
                const int ix = a_ix0[g] + (STEM ? (a_chunk[g] >> 2) : kx);
                const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int off = STEM ? (a_pb[g] + iy * a.W + ix) * 4 : (a_pb[g] + iy * a.W + ix) * a.Cin + a_chunk[g];
                a_voff[g] = v ? (unsigned int)(off * 4) : OOB_VOFF;
            }
        }
        float* As = lds + buf * TILE_FLOATS;
        float* Bs = As + NPL * BM * BK;
        const int a_soff = cs * (BK * 4) + a_win;   // bytes, block-uniform
        const int b_soff = kstep * (NPL * BK * 4);  // (G32: a tap's weights are two groups of 32 channels, this row's one picked in b_voff)
#pragma unroll
        for (int h = 0; h < NPL; ++h)       // (G32: plane h = channels [32 h, 32 h + 32) of the window: the same rows, 128 B further)
#pragma unroll
            for (int g = 0; g < GA; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + h * BM * BK + (wave * (BM / NW) + 8 * g) * BK),
                                                         16, (int)a_voff[g], a_soff + h * (BK * 4), 0, 0);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * (BN / NW) + 8 * g) * BK),
                                                     16, (int)b_voff[g], b_soff, 0, 0);
        ++kstep;
        if (++cs == csteps) {
            cs = 0;
            if (++kx == kw_taps) { kx = 0; ++ky; }
        }
    };

    f32x16 acc[MT][NT];                          // fp32 MFMA (F16 = false)
    constexpr int MB = WTM / 16, NB = WTN / 16;  // F16: blocks of 16 x 16 (f16x3_step16)
    f32x4 acc16[MB][NB], acx16[MB][NB];          // hi*hi sums; cross-term sums (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc16[i][j][e] = 0.f; acx16[i][j][e] = 0.f; }
    const int l15 = lane & 15, lq = lane >> 4;
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));

    // fragment read offsets (floats): row*32 + 4*((2q+lh) ^ swz(row)), swz(row) = (l31>>1)&7 for every 32-row fragment
    const int fswz = (l31 >> 1) & 7;
    int foff[BK / 8];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) foff[q] = 4 * ((2 * q + lh) ^ fswz);

    if constexpr (RING) {
        constexpr int NDMA = NPL * GA + GB;     // LDS-DMA requests of this wave per tile
        static_assert(NDMA < 16, "vmcnt immediate");
        stage(0);
        if (nsteps > 1) stage(1);
        int cur = 0, nxt = 2;
        for (int step = 0; step < nsteps; ++step) {
            // tile `step` has landed once all but the youngest tile's requests of this wave are done (in-order vmcnt); bare s_barrier: the
            // fence of __syncthreads() would wait for vmcnt(0), i.e. for the tile the ring keeps in flight (see conv_split_kernel)
            if (step + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0070 | NDMA);
            else __builtin_amdgcn_s_waitcnt(0x0070);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();       // every wave's share of tile `step` is in LDS; everyone is done with tile step-1
            asm volatile("" ::: "memory");
            if (step + 2 < nsteps) stage(nxt);  // into the buffer tile step-1 occupied
            f16x3_step16<MB, NB>(lds + cur * TILE_FLOATS + (G32 ? wn * BM * BK : 0) + (wm * WTM + l15) * BK,
                                 lds + cur * TILE_FLOATS + NPL * BM * BK + (wn * WTN + l15) * BK, fo16_hi, fo16_lo, acc16, acx16);
            cur = (cur == 2) ? 0 : cur + 1;
            nxt = (nxt == 2) ? 0 : nxt + 1;
        }
        __syncthreads();                        // all operand reads done: the epilogue re-uses the buffers as its staging tile
    } else {
    stage(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the LDS-DMA has landed (explicit: do not rely on the fence lowering)
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < nsteps) stage(cur ^ 1);

        const float* As = lds + cur * TILE_FLOATS + (wm * WTM + l31) * BK;
        const float* Bs = lds + cur * TILE_FLOATS + NPL * BM * BK + (wn * WTN + l31) * BK;
        if (F16) {
            f16x3_step16<MB, NB>(lds + cur * TILE_FLOATS + (G32 ? wn * BM * BK : 0) + (wm * WTM + l15) * BK,
                                 lds + cur * TILE_FLOATS + NPL * BM * BK + (wn * WTN + l15) * BK, fo16_hi, fo16_lo, acc16, acx16);
        } else {
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * BK + foff[q]);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * BK + foff[q]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][t], bf[j][t], acc[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the LDS-DMA of the next tile has landed
        __syncthreads();   // everyone is done with `cur`
    }
    }

    if (F16) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc16[i][j][e] = __fadd_rn(acc16[i][j][e], __fmul_rn(acx16[i][j][e], 1.0f / LO_SCALE));
        if (EPI == 0) conv_epilogue16<WTM, WTN, MB, NB, true>(a, acc16, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN, HoWo);
        else conv_epilogue_fast16<WTM, WTN, MB, NB, EPI == 2, true>(a, acc16, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
    } else {
        if (EPI == 0) conv_epilogue<WTM, WTN, MT, NT, false>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN, HoWo);
        else conv_epilogue_fast<WTM, WTN, MT, NT, EPI == 2, false>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
    }
}


// The predictor rows a wave tile can need, fetched ahead of the epilogue (conv_split_kernel issues this BEFORE its K loop): a wave tile is 64
// consecutive GEMM rows = pixels of at most TWO RoIs (196 pixels each), so two class ids, two weight quads per lane and two biases cover it.
// Before round 4 every one of the 16 row groups of the epilogue walked pred_cls[b] -> pred_w[cls] again -- two dependent global round trips
// per group, and two more in the final reduction: ~8 us of a 19-us tile (K = 256: 8 K-steps) that nothing hid.
struct PredictPrefetch {
    unsigned int b0;
    f32x4 wp0, wp1;
    float pb0, pb1;
};
__device__ __forceinline__ PredictPrefetch predict_prefetch(const ConvArgs& a, int wave, int lane, int m0) {
    const int wm = wave >> 2, wn = wave & 3, l15 = lane & 15;
    const int mw0 = m0 + wm * 64;
    const unsigned int mf = min((unsigned int)mw0, (unsigned int)(a.M - 1)), ml = min((unsigned int)(mw0 + 63), (unsigned int)(a.M - 1));
    PredictPrefetch p;
    p.b0 = fastdiv(mf, a.div_howo_mul, a.div_howo_shr);
    const unsigned int b1 = fastdiv(ml, a.div_howo_mul, a.div_howo_shr);
    int c0 = a.pred_cls[p.b0], c1 = a.pred_cls[b1];
    c0 = (c0 >= 0 && c0 < a.pred_K) ? c0 : 0;
    c1 = (c1 >= 0 && c1 < a.pred_K) ? c1 : 0;
    const int co = wn * 64 + l15 * 4;
    p.wp0 = *reinterpret_cast<const f32x4*>(a.pred_w + (size_t)c0 * 256 + co);
    p.wp1 = *reinterpret_cast<const f32x4*>(a.pred_w + (size_t)c1 * 256 + co);
    p.pb0 = a.pred_b[c0];
    p.pb1 = a.pred_b[c1];
    return p;
}

// Epilogue of the fused mask-head tail (out_mode 3).  The GEMM is the ConvTranspose2d 2x2 s2 as a 1x1 convolution to
// Cout = 4 * 256 channels ordered (tap, co); a 128 x 256 tile is therefore 128 input pixels x ONE tap x all 256 output channels.
// Per tile row (RoI b, input pixel (oy, ox), tap (ky, kx)): v[co] = relu(acc[co] + bias[co]) is the deconv output at output pixel
// (2oy + ky, 2ox + kx); the predictor's logit for the RoI's class c is  sum_co v[co] * wp[c][co] + bp[c];  the mask probability its
// sigmoid.  fp32 products and sums in a fixed order (4 columns per lane, xor tree over the 16 lanes of a row, the 4 N-waves in order):
// deterministic.  Replaces a 1.28 GB write, its read-back, the 1x1 predictor GEMM and mask_prob_kernel (B = 8, 200 detections).
__device__ __forceinline__ void conv_epilogue_predict(const ConvArgs& a, f32x4 (&acc)[4][4], float* lds, int wave, int lane, int m0, int n0, const PredictPrefetch& pf) {
    constexpr int WTM = 64, WTN = 64, SLD = WTN + 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    float* stage = lds + wave * (WTM * SLD);
    float* red = lds + 8 * (WTM * SLD);                 // [4 N-waves][128 rows] partial sums (2 KiB behind the staging tiles)
#pragma unroll
    for (int i = 0; i < 4; ++i)                          // role-swapped accumulators: lane = (tile row l15, channels swap_channel(16 j + 4 lq + e))
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(stage + (i * 16 + l15) * SLD + 32 * (j >> 1) + 8 * lq + 4 * (j & 1)) = acc[i][j];
    __builtin_amdgcn_wave_barrier();
    const int tap = n0 >> 8;                            // Cout = 4 * 256: the tile's N range is one tap
    const int co = wn * WTN + l15 * 4;                  // this lane's 4 output channels
    const int mw0 = m0 + wm * WTM;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (a.shift) bias = *reinterpret_cast<const f32x4*>(a.shift + n0 + co);
    bool bad = false;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int rl = it * 4 + lq;                     // row of the wave tile
        const unsigned int m = min((unsigned int)(mw0 + rl), (unsigned int)(a.M - 1));
        const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
        const f32x4 wp = (b == pf.b0) ? pf.wp0 : pf.wp1;         // the RoI's class row, fetched ahead (same values as pred_w[pred_cls[b]])
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + rl * SLD + l15 * 4);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bad = bad || !(fabsf(v[q]) <= 3.0e38f);
            const float o = fmaxf(__fadd_rn(v[q], bias[q]), 0.f);
            s = __fadd_rn(s, __fmul_rn(o, wp[q]));
        }
        s = __fadd_rn(s, __shfl_xor(s, 8, 16));
        s = __fadd_rn(s, __shfl_xor(s, 4, 16));
        s = __fadd_rn(s, __shfl_xor(s, 2, 16));
        s = __fadd_rn(s, __shfl_xor(s, 1, 16));
        if (l15 == 0) red[wn * 128 + wm * WTM + rl] = s;
    }
    if (bad) atomicOr(a.range_flag, 1);
    __syncthreads();
    if (wn == 0) {                                      // 2 waves x 64 rows: one row per lane
        const int rl = wm * WTM + lane;
        const int m = m0 + rl;
        if (m < a.M) {
            const unsigned int b = fastdiv((unsigned int)m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = (unsigned int)m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            float x = __fadd_rn(__fadd_rn(__fadd_rn(red[rl], red[128 + rl]), red[256 + rl]), red[384 + rl]);
            x = __fadd_rn(x, (b == pf.b0) ? pf.pb0 : pf.pb1);
            const float p = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x)));
            a.prob[((size_t)b * 2 * a.Ho + 2 * oy + (tap >> 1)) * (2 * a.Wo) + 2 * ox + (tap & 1)] = p;
        }
    }
}

// Epilogue of the fused RPN tail (out_mode 4, 128 x 256 tile = 128 pixels x all 256 hidden channels).  With the role-swapped MFMA a lane
// holds, per tile row, channels [8 lq, 8 lq + 8) of each 32-channel group of its wave's 64 -- after affine, ReLU and the split, exactly the
// A fragment (row = pixel l15, k group lq) of a v_mfma_f32_16x16x32_f16 over that group.  So the 1x1 predictors are a second f16x3
// product right here: B = the predictor rows (16 outputs x 32 channels, split rows, straight from memory), 2 groups per wave, the four
// N-waves' partial sums added through LDS in wave order, + bias -> pred [M][16].  Same operands as the separate 1x1 launch read from the
// stored hidden tensor (the halves are identical), another summation order.  Replaces a 537 MB write + read at p2 and a launch per level.
// row_of(i, r): the GEMM row (pixel) of row r (0..15) of the wave tile's 16-row block i, or a.M (or more) for a row outside the image
template <class RowOf>
__device__ __forceinline__ void conv_epilogue_rpn_rows(const ConvArgs& a, f32x4 (&acc)[4][4], float* lds, int wave, int lane, int n0, RowOf row_of) {
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nw0 = n0 + wn * 64;
    f32x2 sc[2][4], sh[2][4];
    f16x8 wh[2], wl[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int n = nw0 + 32 * g + 8 * lq;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            sc[g][p] = a.scale ? f32x2{a.scale[n + 2 * p], a.scale[n + 2 * p + 1]} : f32x2{1.f, 1.f};
            sh[g][p] = a.shift ? f32x2{a.shift[n + 2 * p], a.shift[n + 2 * p + 1]} : f32x2{0.f, 0.f};
        }
        const char* wb = reinterpret_cast<const char*>(a.rpn_w + (size_t)l15 * a.Cout) + (size_t)(n >> 5) * 128 + (size_t)(n & 31) * 2;
        wh[g] = *reinterpret_cast<const f16x8*>(wb);
        wl[g] = *reinterpret_cast<const f16x8*>(wb + 64);
    }
    f32x4* red = reinterpret_cast<f32x4*>(lds);                     // [wm][wn][i][lane]
    f32x2 chk = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 d = {0.f, 0.f, 0.f, 0.f}, dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f16x8 hi, lo;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4& blk = acc[i][2 * g + (p >> 1)];
                const f32x2 v = {blk[2 * (p & 1)], blk[2 * (p & 1) + 1]};
                chk = __builtin_elementwise_fma(v, f32x2{0.f, 0.f}, chk);
                f32x2 o = v * sc[g][p] + sh[g][p];
                o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f);
                const f16x2 h = __builtin_convertvector(o, f16x2);
                const f32x2 hf = {(float)h[0], (float)h[1]};
                const f32x2 l = __builtin_elementwise_fma(hf, f32x2{-LO_SCALE, -LO_SCALE}, o * LO_SCALE);
                const f16x2 lh = __builtin_convertvector(l, f16x2);
                hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
                lo[2 * p] = lh[0]; lo[2 * p + 1] = lh[1];
            }
            dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, wh[g], dx, 0, 0, 0);
            dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wl[g], dx, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wh[g], d, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __fadd_rn(d[e], __fmul_rn(dx[e], 1.0f / LO_SCALE));
        red[((wm * 4 + wn) * 4 + i) * 64 + lane] = d;
    }
    if (!(chk[0] == 0.f) || !(chk[1] == 0.f)) atomicOr(a.range_flag, 1);
    __syncthreads();
    if (wn == 0) {                                                  // lane = (output l15, pixels 4 lq .. 4 lq + 3 of block i)
        const float bias = a.rpn_b[l15];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 s = red[((wm * 4 + 0) * 4 + i) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 t = red[((wm * 4 + w) * 4 + i) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], t[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = row_of(i, 4 * lq + e);
                if (m < a.M) a.rpn_pred[(size_t)m * 16 + l15] = __fadd_rn(s[e], bias);
            }
        }
    }
}
__device__ __forceinline__ void conv_epilogue_rpn(const ConvArgs& a, f32x4 (&acc)[4][4], float* lds, int wave, int lane, int m0, int n0) {
    const int mw0 = m0 + (wave >> 2) * 64;
    conv_epilogue_rpn_rows(a, acc, lds, wave, lane, n0, [mw0](int i, int r) { return mw0 + i * 16 + r; });
}

// ------------------------------------------------------------------------------------------------------------------
// conv_split_kernel: AMP_CONV_F16X3 with BOTH operands already in the split hi|lo' row format (the trunk's native activation
// format in inference, and the pre-split weights), 8 waves on a 128x256 or 256x128 tile, one workgroup per CU.
// Compared with conv_glds_kernel<.., F16 = true> (two LDS buffers, vmcnt(0) + barrier at the end of every K-step) the operand
// tiles travel through a ring of THREE LDS buffers: the LDS-DMA of tile s+2 is issued when tile s starts computing, and the wait
// in front of tile s is vmcnt(NDMA) -- everything but the youngest tile's requests.  Why: a K-step is 24 MFMAs per wave
// (1536 cycles per SIMD at two waves per SIMD, ~0.8 us) but an LDS-DMA request needs ~1.1 us from issue to landing when every CU
// streams (MI355X_MICROARCH.md, cost cell "ldsdma-fill"), so with one tile in flight every step ended in a stall on its own
// prefetch; with two in flight the request has two steps to land.  3 x 48 KB = 144 KB of the CU's 160 KB.
// ------------------------------------------------------------------------------------------------------------------
#ifdef AMP_STAMP
// Lab build only (make EXTRA=-DAMP_STAMP): in-kernel phase timing of conv_split_kernel -- cycles per wave slot and phase, summed over
// all workgroups: [wave][0] wait+barrier, [1] DMA issue, [2] fragment reads (issue + return), [3] MFMA issue, [4] whole loop,
// [5] kernel entry -> loop, [6] loop end -> epilogue stores issued, [7] loop end -> past the final barrier.
__device__ unsigned long long g_stamp[8 * 8];
// [0] shader cycles (s_memtime) and [1] 100-MHz ticks (s_memrealtime) around the K loop of wave 0, summed over workgroups: the clock the chip held
// in the loop = [0] / [1] * 100 MHz (MI355X_MICROARCH.md, "DVFS give-back" item 6)
__device__ unsigned long long g_stamp_clk[2];
#define STAMP_T(var) const long long var = clock64()
#define STAMP_ADD(slot, t0, t1) st_acc[slot] += (t1) - (t0)
#else
#define STAMP_T(var)
#define STAMP_ADD(slot, t0, t1)
#endif
template <int BM, int BN, int EPI, int NSTAGE = 3, bool CHAN = false>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, NSTAGE == 2 ? 2 : 1) void conv_split_kernel(const ConvArgs a, const unsigned int x_bytes,
                                                                                                     const unsigned int w_bytes) {
    constexpr int WTM = 64, WTN = 64;
    constexpr int NWM = BM / WTM, NWN = BN / WTN, NW = NWM * NWN;
    constexpr int MB = WTM / 16, NB = WTN / 16;
    constexpr int GA = BM / NW / 8;       // DMA instructions per wave per K-step for A (8 rows x 128 B each)
    constexpr int GB = BN / NW / 8;       // ... for B
    constexpr int NDMA = GA + GB;
    // NSTAGE = 2 (128 x 128 tiles, byte-bound short-K layers): 68 KB of LDS, 4 waves -> TWO workgroups per CU
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    constexpr int SLD = WTN + 4;
    constexpr int STAGE_FLOATS = NW * WTM * SLD;
    constexpr int LDS_FLOATS = (NSTAGE * TILE_FLOATS > STAGE_FLOATS) ? NSTAGE * TILE_FLOATS : STAGE_FLOATS;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(GA >= 1 && GB >= 1 && NDMA < 16, "tile / wave split");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
#ifdef AMP_STAMP
    const long long st_entry = clock64();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    const int tile = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = tile % a.ntn;
    const int tile_m = tile / a.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);

    // ---- staging geometry (as conv_glds_kernel): instruction g of this wave fills rows wave*R + 8g + (lane>>3), 16-B position lane&7 ----
    const int srow = lane >> 3, spos = lane & 7;
    int a_iy0[GA], a_ix0[GA], a_pb[GA], a_chunk[GA];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        const int r = wave * (BM / NW) + 8 * g + srow;
        a_chunk[g] = 4 * (spos ^ ((r >> 1) & 7));
        const int m = m0 + r;
        if (m < a.M) {
            const unsigned int b = fastdiv((unsigned int)m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = (unsigned int)m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            a_iy0[g] = (int)oy * a.stride - a.pad;
            a_ix0[g] = (int)ox * a.stride - a.pad;
            a_pb[g] = (int)b * a.H * a.W;
        } else {
            a_iy0[g] = -(1 << 28);
            a_ix0[g] = 0;
            a_pb[g] = 0;
        }
    }
    unsigned int b_voff[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * (BN / NW) + 8 * g + srow;               // LDS row; its weight row follows swap_channel (see conv_epilogue_direct)
        const int n = n0 + (r & ~63) + swap_channel(r & 63);
        b_voff[g] = (n < a.Cout) ? (unsigned int)(((size_t)n * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
    unsigned int a_voff[GA];

    const int csteps = a.Cin / BK;
    int ky = 0, kx = 0, cs = 0, kstep = 0;     // block-uniform state of the tile being STAGED

    constexpr bool chan_major = CHAN;      // compile-time: a run-time flag here cost 12 B of scratch and a vmcnt(0) per step (250 VGPRs, 96 SGPRs in use)
    auto stage = [&](int buf) {
        if (cs == 0 || chan_major) {           // the tap changed: new pixel offsets (channel-major: every step)
#pragma unroll
            for (int g = 0; g < GA; ++g) {
                const int iy = a_iy0[g] + ky, ix = a_ix0[g] + kx;
                const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                a_voff[g] = v ? (unsigned int)(((a_pb[g] + iy * a.W + ix) * a.Cin + a_chunk[g]) * 4) : OOB_VOFF;
            }
        }
        float* As = lds + buf * TILE_FLOATS;
        float* Bs = As + BM * BK;
        const int a_soff = cs * (BK * 4);
        int b_soff;
        if constexpr (chan_major) b_soff = ((ky * a.KW + kx) * csteps + cs) * (BK * 4);
        else b_soff = kstep * (BK * 4);
#pragma unroll
        for (int g = 0; g < GA; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + (wave * (BM / NW) + 8 * g) * BK),
                                                     16, (int)a_voff[g], a_soff, 0, 0);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * (BN / NW) + 8 * g) * BK),
                                                     16, (int)b_voff[g], b_soff, 0, 0);
        if constexpr (chan_major) {
            if (++kx == a.KW) {
                kx = 0;
                if (++ky == a.KH) { ky = 0; ++cs; }
            }
        } else {
            ++kstep;
            if (++cs == csteps) {
                cs = 0;
                if (++kx == a.KW) { kx = 0; ++ky; }
            }
        }
    };

    f32x4 acc[MB][NB], acx[MB][NB];            // hi*hi sums; cross-term sums (scaled by 2^11): 128 registers
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    const int l15 = lane & 15, lq = lane >> 4;
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));

    stage(0);
    if (a.nsteps > 1) stage(1);
    // fused mask-head tail only: its predictor rows, requested here -- behind the first two tiles' DMA (the class-id load the row addresses
    // wait for must not delay the staging; younger loads in the queue only make the ring's counted vmcnt waits stricter, never looser)
    PredictPrefetch ppf;
    if constexpr (BM == 128 && BN == 256 && EPI == 3) {
        if (a.out_mode == 3) ppf = predict_prefetch(a, wave, lane, m0);
    }
    int cur = 0, nxt = 2;                      // ring positions of the tile being computed / staged
    auto open_step = [&](int step) {
        // tile `step` has landed once all but the youngest tile's requests of this wave are done (LDS-DMA counts in vmcnt, in order);
        // lgkmcnt(0): this wave's fragment reads of tile step-1 have returned (the staggered half issues them last in its step)
        if (step + 1 < a.nsteps) __builtin_amdgcn_s_waitcnt(0x0070 | NDMA);
        else __builtin_amdgcn_s_waitcnt(0x0070);
        // Bare s_barrier: __syncthreads() is fence + barrier, and the workgroup-release fence waits for vmcnt(0) -- it would drain the
        // very requests this ring keeps in flight.  Nothing a fence orders is needed here: the tiles are written by LDS-DMA (covered by
        // the vmcnt above) and read by ds_read whose data has returned (lgkmcnt above).  The empty asm statements keep the compiler
        // from moving LDS accesses across the barrier.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's share of tile `step` is in LDS; everyone is done with tile step-1
        asm volatile("" ::: "memory");
    };
    auto advance = [&]() {
        cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
        nxt = (nxt == NSTAGE - 1) ? 0 : nxt + 1;
    };
    // Ping-pong between the two waves of a SIMD (MI355X_MICROARCH.md, "Two waves per SIMD"): waves w and w + NW/2 share a SIMD, and run
    // in lockstep they first both issue LDS-DMA and fragment reads (matrix pipe idle) and then both issue MFMAs (pipe contended) -- the
    // in-kernel stamps (tools/stamp_conv.py) gave 2400-2540 cycles per K-step against 1536 of MFMA work.  So the second half of the
    // workgroup runs its MFMAs one step late: a K-step has two halves separated by a second barrier; in the first the early half
    // stages tile s+2 and reads the fragments of tile s while the late half multiplies the fragments of tile s-1 it holds in registers;
    // in the second the early half multiplies tile s while the late half stages and reads.  The multiplying half runs at raised priority.
    // Same tiles, same sums per accumulator in the same order -- bit-identical results (1930-2010 cycles per K-step; +5..14 % per layer).
    F16x3Frags<MB, NB> fr;
#ifdef AMP_STAMP
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_mt0 = __builtin_amdgcn_s_memtime();
    const long long st_begin = clock64();
#endif
    const float* As0 = lds + (wm * WTM + l15) * BK;
    const float* Bs0 = lds + BM * BK + (wn * WTN + l15) * BK;
    const bool pp = a.stagger != 0;
    if (!pp || wave < NW / 2) {
        for (int step = 0; step < a.nsteps; ++step) {
            STAMP_T(t0);
            open_step(step);
            STAMP_T(t1);
            // fragment reads first: they return while the DMA requests are being issued (~75 cycles each), not in front of the MFMAs
            f16x3_load16<MB, NB>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t2);
            if constexpr (NSTAGE == 2) {
                // two buffers: tile step+2 goes into the buffer of tile `step` itself, once every wave holds its fragments in registers
                if (step + 2 < a.nsteps) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's fragment reads have returned
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    stage(cur);
                }
            } else
            if (step + 2 < a.nsteps) stage(nxt);   // into the buffer tile step-1 occupied
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t3);
            if (pp) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            f16x3_mfma16<MB, NB, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t4);
            STAMP_ADD(0, t0, t1); STAMP_ADD(2, t1, t2); STAMP_ADD(1, t2, t3); STAMP_ADD(3, t3, t4);
            advance();
        }
    } else {
        for (int step = 0; step < a.nsteps; ++step) {
            STAMP_T(t0);
            open_step(step);
            STAMP_T(t1);
            __builtin_amdgcn_s_setprio(1);
            if (step > 0) f16x3_mfma16<MB, NB, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            STAMP_T(t2);
            f16x3_load16<MB, NB>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t3);
            if (step + 2 < a.nsteps) stage(nxt);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t4);
            STAMP_ADD(0, t0, t1); STAMP_ADD(3, t1, t2); STAMP_ADD(2, t2, t3); STAMP_ADD(1, t3, t4);
            advance();
        }
        f16x3_mfma16<MB, NB, true>(fr, acc, acx);
    }
#ifdef AMP_STAMP
    if (wave == 0 && lane == 0) {
        atomicAdd(&g_stamp_clk[0], __builtin_amdgcn_s_memtime() - st_mt0);
        atomicAdd(&g_stamp_clk[1], __builtin_amdgcn_s_memrealtime() - st_rt0);
    }
    st_acc[4] = clock64() - st_begin;
    st_acc[5] = st_begin - st_entry;
    const long long st_loop_end = clock64();
#endif
    __syncthreads();                           // all operand reads done: the epilogue re-uses the buffers as its staging tile
#ifdef AMP_STAMP
    st_acc[7] = clock64() - st_loop_end;
#endif
    const bool scaled_in = a.out_scale != 1.0f;

#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[i][j][e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                if (scaled_in) acc[i][j][e] *= a.out_scale;      // x arrived as split rows of x * 2^shift (a scaled loss gradient): exact power of two
            }
    if constexpr (BM == 128 && BN == 256 && EPI == 3) {          // the fused tails have an instantiation of their own: no row epilogue, no spills
        if (a.out_mode == 3) conv_epilogue_predict(a, acc, lds, wave, lane, m0, n0, ppf);
        else conv_epilogue_rpn(a, acc, lds, wave, lane, m0, n0);
        return;
    }
    if (a.y_split && (!a.mask || a.mask_split) && (a.out_mode == 0 || (EPI == 2 && a.out_mode == 1 && !a.mask && a.res_mode == 0)) && (a.res_mode == 0 || a.res_split) && a.direct_epi)
        conv_epilogue_direct<EPI == 2, true>(a, acc, lane, m0 + wm * WTM, n0 + wn * WTN);
    else
        conv_epilogue_swapped<WTM, WTN, EPI == 2, true>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
#ifdef AMP_STAMP
    st_acc[6] = clock64() - st_loop_end;
    if (lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_stamp[wave * 8 + q], (unsigned long long)st_acc[q]);
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// mask_tail_kernel (round 4): the mask head's tail -- ConvTranspose 2x2 s2 as a GEMM to (tap, co) = 4 x 256 columns, ReLU, the predictor row of the
// detection's class, sigmoid -- with the FOUR taps of a 128-pixel block in ONE workgroup.  conv_split_kernel<128, 256, 3> ran one (pixel block, tap) tile
// per workgroup: K = 256 is eight K-steps, so every workgroup paid a cold ring (3300 cycles per step against 1950 in steady state: each step waits for
// its DMA), a prologue and an epilogue as long as the ideal loop for 5 us of MFMA work -- 17 us per tile (tools/stamp_conv.py deconv.gemm).  Here the ring
// runs through all 32 K-steps: tap t + 1's first tiles are requested during tap t's last steps (the activation rows come out of the L2 the second time),
// and a tap's epilogue needs no LDS staging: with the role-swapped MFMA a lane holds 16 channels of ONE pixel per row block -- the four 4-column groups
// 8 (j >> 1) + 2 lq + (j & 1) of conv_epilogue_predict's lanes -- so its sums are that epilogue's sums in that epilogue's order (4 columns in turn, then the
// xor tree 8, 4, 2, 1 over the groups: in-lane, lanes ^ 32, lanes ^ 16, in-lane; then the four N-waves in wave order through 2 KB of LDS behind the ring):
// BIT-IDENTICAL probabilities.  The two halves of the workgroup (tile rows 0-63 / 64-127) exchange only within themselves, at the loop's own barriers.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void mask_tail_kernel(const ConvArgs a, const unsigned int x_bytes, const unsigned int w_bytes) {
    constexpr int BM = 128, BN = 256, NW = 8, GA = 2, GB = 4, NDMA = GA + GB, NTAP = 4;
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    // behind the ring: the N-waves' partial sums, the predictor rows of the (at most two) RoIs of the 128 pixels, their biases, the deconv's bias
    __shared__ __attribute__((aligned(16))) float lds[3 * TILE_FLOATS + 4 * BM + 2 * 256 + 4 + NTAP * BN];
    float* red = lds + 3 * TILE_FLOATS;                                   // [N-wave][128 rows]
    float* wps = red + 4 * BM;                                            // [RoI slot][256 columns]
    float* pbs = wps + 2 * 256;                                           // [RoI slot]
    float* dbias = pbs + 4;                                               // [tap][256 columns]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int m0 = amp::xcd_remap(blockIdx.x, a.nblk) * BM;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);
    const int srow = lane >> 3, spos = lane & 7;
    unsigned int a_voff[GA], b_voff[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        const int r = wave * (BM / NW) + 8 * g + srow;
        const int m = m0 + r;
        a_voff[g] = (m < a.M) ? (unsigned int)(((size_t)m * a.Cin + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * (BN / NW) + 8 * g + srow;
        const int n = (r & ~63) + swap_channel(r & 63);
        b_voff[g] = (unsigned int)(((size_t)n * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4);
    }
    const int csteps = a.Cin / BK;                 // K-steps per tap (8)
    const int total = NTAP * csteps;
    const int tap_bytes = BN * a.K * 4;            // weight rows of one tap
    int s_cs = 0, s_tap = 0;                       // the step being STAGED
    auto stage = [&](int buf) {
        float* As = lds + buf * TILE_FLOATS;
        float* Bs = As + BM * BK;
        const int a_soff = s_cs * (BK * 4);
        const int b_soff = s_tap * tap_bytes + s_cs * (BK * 4);
#pragma unroll
        for (int g = 0; g < GA; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + (wave * (BM / NW) + 8 * g) * BK), 16, (int)a_voff[g], a_soff, 0, 0);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * (BN / NW) + 8 * g) * BK), 16, (int)b_voff[g], b_soff, 0, 0);
        if (++s_cs == csteps) { s_cs = 0; ++s_tap; }
    };
    f32x4 acc[4][4], acx[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    };
    zero_acc();
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));
    stage(0);
    stage(1);
    // the predictor rows of the block's RoIs (128 consecutive pixels of 196-pixel RoIs: at most two) and the deconv's bias go into LDS once, requested
    // behind the first two tiles' DMA (the class-id load the row addresses wait for must not delay the staging)
    const int mw0 = m0 + wm * 64;
    const unsigned int b_first = fastdiv(min((unsigned int)m0, (unsigned int)(a.M - 1)), a.div_howo_mul, a.div_howo_shr);
    {
        const unsigned int b_last = fastdiv(min((unsigned int)(m0 + BM - 1), (unsigned int)(a.M - 1)), a.div_howo_mul, a.div_howo_shr);
        const int slot = tid >> 8, col = tid & 255;
        int c = a.pred_cls[slot ? b_last : b_first];
        c = (c >= 0 && c < a.pred_K) ? c : 0;
        wps[tid] = a.pred_w[(size_t)c * 256 + col];
        if (col == 0) pbs[slot] = a.pred_b[c];
        dbias[tid] = a.shift ? a.shift[tid] : 0.f;
        dbias[512 + tid] = a.shift ? a.shift[512 + tid] : 0.f;
    }
    bool bad = false;
    // a tap's sums of this wave: per tile row, the 64 columns of the wave against the RoI's predictor row -> red[wn][row]
    auto partial_sums = [&](int tap) {
        f32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = *reinterpret_cast<const f32x4*>(dbias + tap * BN + wn * 64 + 32 * (j >> 1) + 8 * lq + 4 * (j & 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned int m = min((unsigned int)(mw0 + i * 16 + l15), (unsigned int)(a.M - 1));
            const unsigned int b = fastdiv(m, a.div_howo_mul, a.div_howo_shr);
            const float* wrow = wps + ((b == b_first) ? 0 : 256) + wn * 64 + 8 * lq;
            float sj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 wp = *reinterpret_cast<const f32x4*>(wrow + 32 * (j >> 1) + 4 * (j & 1));
                float sg = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                    bad = bad || !(fabsf(v) <= 3.0e38f);
                    const float o = fmaxf(__fadd_rn(v, bias[j][e]), 0.f);
                    sg = __fadd_rn(sg, __fmul_rn(o, wp[e]));
                }
                sj[j] = sg;
            }
            float u0 = __fadd_rn(sj[0], sj[2]), u1 = __fadd_rn(sj[1], sj[3]);      // xor 8 of the group index: j ^ 2, in the lane
            u0 = __fadd_rn(u0, __shfl_xor(u0, 32)); u1 = __fadd_rn(u1, __shfl_xor(u1, 32));      // xor 4: lq ^ 2
            u0 = __fadd_rn(u0, __shfl_xor(u0, 16)); u1 = __fadd_rn(u1, __shfl_xor(u1, 16));      // xor 2: lq ^ 1
            const float sres = __fadd_rn(u0, u1);                                   // xor 1: j ^ 1, in the lane
            if (lq == 0) red[wn * BM + wm * 64 + i * 16 + l15] = sres;
        }
    };
    // ... and the end of a tap for the 64 rows of this half: one row per lane of its N-wave 0
    auto finish_rows = [&](int tap) {
        const int rl = wm * 64 + lane;
        const int m = m0 + rl;
        if (m < a.M) {
            const unsigned int b = fastdiv((unsigned int)m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = (unsigned int)m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            float x = __fadd_rn(__fadd_rn(__fadd_rn(red[rl], red[BM + rl]), red[2 * BM + rl]), red[3 * BM + rl]);
            x = __fadd_rn(x, pbs[(b == b_first) ? 0 : 1]);
            const float p = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x)));
            a.prob[((size_t)b * 2 * a.Ho + 2 * oy + (tap >> 1)) * (2 * a.Wo) + 2 * ox + (tap & 1)] = p;
        }
    };
    int cur = 0, nxt = 2, c_cs = 0, c_tap = 0;      // ring positions; the step being COMPUTED
    auto open_step = [&](int step) {
        if (step + 1 < total) __builtin_amdgcn_s_waitcnt(0x0070 | NDMA);
        else __builtin_amdgcn_s_waitcnt(0x0070);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto advance = [&]() {
        cur = (cur == 2) ? 0 : cur + 1;
        nxt = (nxt == 2) ? 0 : nxt + 1;
        if (++c_cs == csteps) { c_cs = 0; ++c_tap; }
    };
    F16x3Frags<4, 4> fr;
    const float* As0 = lds + (wm * 64 + l15) * BK;
    const float* Bs0 = lds + BM * BK + (wn * 64 + l15) * BK;
    if (wave < NW / 2) {
        for (int step = 0; step < total; ++step) {
            open_step(step);
            if (c_cs == 0 && c_tap > 0 && wn == 0) finish_rows(c_tap - 1);       // the previous tap's sums of this half: written before this barrier
            f16x3_load16<4, 4>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < total) stage(nxt);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c_cs == csteps - 1) { partial_sums(c_tap); zero_acc(); }
            advance();
        }
    } else {
        for (int step = 0; step < total; ++step) {
            open_step(step);
            __builtin_amdgcn_s_setprio(1);
            if (step > 0) f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            const bool tap_done = c_cs == 0 && c_tap > 0;                         // the MFMAs above finished tap c_tap - 1
            if (tap_done) { partial_sums(c_tap - 1); zero_acc(); __builtin_amdgcn_s_waitcnt(0xc07f); }
            __builtin_amdgcn_s_barrier();
            if (tap_done && wn == 0) finish_rows(c_tap - 1);
            f16x3_load16<4, 4>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < total) stage(nxt);
            __builtin_amdgcn_sched_barrier(0);
            advance();
        }
        f16x3_mfma16<4, 4, true>(fr, acc, acx);
        partial_sums(NTAP - 1);
    }
    __syncthreads();
    if (wn == 0) finish_rows(NTAP - 1);
    if (bad) atomicOr(a.range_flag, 1);
}

// ------------------------------------------------------------------------------------------------------------------
// conv1x1_nloop_kernel (round 4): the short-K 1x1 layers that run one 128 x 256 tile per CU (stride-2 shortcuts, conv1 of a stage's first block, their
// data gradients, the mask head's deconv in a training step: K = 64 ... 512, two to sixteen K-steps) with SEVERAL N tiles of a 128-pixel block in one
// workgroup -- mask_tail_kernel's loop with conv_epilogue_direct_rows as the tile epilogue.  A workgroup of conv_split_kernel pays a cold ring, a prologue and
// an epilogue per tile for 5 us of MFMA work (K = 256); here the ring runs on across the tiles (tile t + 1's first weight tiles and the pixel block's rows,
// now in the L2, are requested during tile t's last steps) and a half's epilogue -- loads of residual / mask, 16 stores per wave -- sits between two steps.
// Those stores are YOUNGER than the requests already in flight, so the counted wait in front of a step allows for them exactly where they are in the queue:
//   early half (epilogue after the last MFMAs of step e):   step e + 1 needs D(e+1); younger: D(e+2), E      -> vmcnt(6 + 16)
//                                                            step e + 2 needs D(e+2); younger: E, D(e+3)      -> vmcnt(6 + 16)
//   late half  (epilogue in step e + 1, before its requests): step e + 1 needs D(e+1); younger: D(e+2)         -> vmcnt(6)
//                                                            step e + 2 needs D(e+2); younger: E, D(e+3)      -> vmcnt(6 + 16)
// and vmcnt(6) from then on (which also retires E).  16 is a LOWER bound on a full tile's stores (more younger operations only make the wait stricter);
// a partial tile (rows beyond M: stores skipped) waits with vmcnt(6).  Same products, same order, same epilogue: bit-identical to conv_split_kernel.
// ------------------------------------------------------------------------------------------------------------------
template <bool SPATIAL>
__global__ __launch_bounds__(512, 1) void conv1x1_nloop_kernel(const ConvArgs a, const unsigned int x_bytes, const unsigned int w_bytes, const int NT) {
    constexpr int BM = 128, BN = 256, NW = 8, GA = 2, GB = 4, NDMA = GA + GB, NST = 16;
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    constexpr int NT_MAX = 8;                      // scale + shift of NT_MAX * 256 channels: the 16 KB behind the ring
    __shared__ __attribute__((aligned(16))) float lds[3 * TILE_FLOATS + 2 * NT_MAX * BN];
    float* lsc = lds + 3 * TILE_FLOATS;
    float* lsh = lsc + NT_MAX * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ngrp = a.ntn / NT;                   // groups of NT consecutive N tiles
    const int wg = amp::xcd_remap(blockIdx.x, a.nblk);
    const int grp = wg % ngrp;
    const int m0 = (wg / ngrp) * BM;
    const int n_first = grp * NT * BN;
    const bool full = m0 + BM <= a.M;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);
    const int srow = lane >> 3, spos = lane & 7;
    unsigned int a_voff[GA], b_voff[GB];
#pragma unroll
    for (int g = 0; g < GA; ++g) {
        const int r = wave * (BM / NW) + 8 * g + srow;
        const int m = m0 + r;
        a_voff[g] = OOB_VOFF;
        if (m < a.M) {                             // 1x1, pad 0: output pixel (b, oy, ox) reads input pixel (oy, ox) * stride
            const unsigned int b = fastdiv((unsigned int)m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = (unsigned int)m - b * (unsigned int)(a.Ho * a.Wo);
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            a_voff[g] = (unsigned int)((((size_t)(b * a.H + oy * a.stride) * a.W + ox * a.stride) * a.Cin + 4 * (spos ^ ((r >> 1) & 7))) * 4);
        }
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * (BN / NW) + 8 * g + srow;
        const int n = n_first + (r & ~63) + swap_channel(r & 63);
        b_voff[g] = (unsigned int)(((size_t)n * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4);
    }
    const int csteps = a.Cin / BK;                 // K-steps per tile
    const int total = NT * csteps;
    const int tile_bytes = BN * a.K * 4;           // weight rows of one N tile
    int s_cs = 0, s_tile = 0;                      // the step being STAGED
    auto stage = [&](int buf) {
        float* As = lds + buf * TILE_FLOATS;
        float* Bs = As + BM * BK;
        const int a_soff = s_cs * (BK * 4);
        const int b_soff = s_tile * tile_bytes + s_cs * (BK * 4);
#pragma unroll
        for (int g = 0; g < GA; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(As + (wave * (BM / NW) + 8 * g) * BK), 16, (int)a_voff[g], a_soff, 0, 0);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * (BN / NW) + 8 * g) * BK), 16, (int)b_voff[g], b_soff, 0, 0);
        if (++s_cs == csteps) { s_cs = 0; ++s_tile; }
    };
    f32x4 acc[4][4], acx[4][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    };
    zero_acc();
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));
    stage(0);
    stage(1);
    // FrozenBN scale / shift (or 1 / bias) of the group's channels: once, into LDS (visible behind the first step's barrier)
    for (int c = tid; c < NT * BN; c += 512) {
        lsc[c] = a.scale ? a.scale[n_first + c] : 1.0f;
        lsh[c] = a.shift ? a.shift[n_first + c] : 0.0f;
    }
    const int mw0 = m0 + wm * 64;
    const bool scaled_in = a.out_scale != 1.0f;
    auto tile_epilogue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[i][j][e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                    if (scaled_in) acc[i][j][e] *= a.out_scale;
                }
        // one 16-row block at a time: the loop's state stays live across this epilogue, and two blocks' hoisted residual / mask rows beside it spill
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mrows[1] = {mw0 + 16 * i + l15};
            conv_epilogue_direct_rows<SPATIAL, true, 1, true, true>(a, reinterpret_cast<f32x4 (&)[1][4]>(acc[i]), lane, mrows, n_first + tile * BN + wn * 64, lsc, lsh, tile * BN + wn * 64);
        }
    };
    int cur = 0, nxt = 2, c_cs = 0, c_tile = 0;    // ring positions; the step being COMPUTED
    const bool early = wave < NW / 2;
    auto open_step = [&](int step) {
        // the stores of the previous tile's epilogue in the queue (see the table above); exact for full tiles only
        const bool behind_epilogue = full && c_tile > 0 && (early ? c_cs <= 1 : c_cs == 1);
        if (step + 1 >= total) __builtin_amdgcn_s_waitcnt(0x0070);
        else if (behind_epilogue) __builtin_amdgcn_s_waitcnt(0x0070 | ((NDMA + NST) & 15) | (((NDMA + NST) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0070 | NDMA);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto advance = [&]() {
        cur = (cur == 2) ? 0 : cur + 1;
        nxt = (nxt == 2) ? 0 : nxt + 1;
        if (++c_cs == csteps) { c_cs = 0; ++c_tile; }
    };
    F16x3Frags<4, 4> fr;
    const float* As0 = lds + (wm * 64 + l15) * BK;
    const float* Bs0 = lds + BM * BK + (wn * 64 + l15) * BK;
    if (early) {
        for (int step = 0; step < total; ++step) {
            open_step(step);
            f16x3_load16<4, 4>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < total) stage(nxt);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c_cs == csteps - 1) { tile_epilogue(c_tile); zero_acc(); __builtin_amdgcn_sched_barrier(0); }
            advance();
        }
    } else {
        for (int step = 0; step < total; ++step) {
            open_step(step);
            __builtin_amdgcn_s_setprio(1);
            if (step > 0) f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (c_cs == 0 && c_tile > 0) { tile_epilogue(c_tile - 1); zero_acc(); __builtin_amdgcn_sched_barrier(0); }      // the MFMAs above finished tile c_tile - 1
            __builtin_amdgcn_s_barrier();
            f16x3_load16<4, 4>(As0 + cur * TILE_FLOATS, Bs0 + cur * TILE_FLOATS, fo16_hi, fo16_lo, fr);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < total) stage(nxt);
            __builtin_amdgcn_sched_barrier(0);
            advance();
        }
        f16x3_mfma16<4, 4, true>(fr, acc, acx);
        tile_epilogue(NT - 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3x3_patch_kernel (round 4): the 3x3 stride-1 pad-1 convolutions of the FPN / RPN / res4 / res5 (Cin % 32 == 0, Cout % 256 == 0) with both
// operands in the split row format -- conv_split_kernel<128, 256>'s loop (ring of three weight tiles filled by LDS-DMA two steps ahead, counted
// vmcnt, bare barriers, the two halves of the workgroup one half-step apart), but the 128 GEMM rows of a workgroup are an 8 x 16 PIXEL tile and
// the activation operand is not staged per K-step: the K order is channel-major (32-channel chunk c outside, tap inside) and the (8 + 2) x (16 + 2)
// pixel patch of chunk c (180 pixels x 128 B, out-of-image pixels zero-filled by the buffer load) is staged ONCE for its nine taps -- tap (ky, kx)
// of tile row y is the same fragment read at patch pixel (y + ky) * 18 + kx + column.  Per K-step a workgroup now moves 32 KB of weights plus a
// ninth of 23 KB instead of 48 KB through the L2 -> LDS path, and an input pixel crosses it 1.4 times per N tile instead of nine: the implicit-GEMM
// kernel's counter traffic was 2.6 x its algorithmic bytes on these layers (profiles/r04/traffic_per_launch_tap_major.txt), on a chip that holds
// 1.84 GHz under this loop (tools/stamp_conv.py) because of the energy it draws.  LDS: 3 x 32 KB weight tiles + 2 x 23 KB patch chunks = 142 KB.
// The patch of chunk c + 1 goes into the other patch buffer during the first three steps of chunk c (one 1-KB piece per wave and step, issued
// behind the step's weight requests; the counted vmcnt in front of a step allows for them).
// Same exact products as every AMP_CONV_F16X3 kernel, summed in conv_split_kernel<.., CHAN = true>'s order (bit-identical to it: AMP_KORDER=1).
// ------------------------------------------------------------------------------------------------------------------
template <int EPI>      // 1: split rows through conv_epilogue_direct_rows; 3: the fused RPN tail (conv_epilogue_rpn_rows)
__global__ __launch_bounds__(512, 1) void conv3x3_patch_kernel(const ConvArgs a, const unsigned int x_bytes, const unsigned int w_bytes, const int tiles_x, const int tiles_y) {
    constexpr int TH = 8, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW;            // 180 patch pixels
    constexpr int NPIECE = 24;                                                      // DMA pieces of 8 rows x 128 B: 3 per wave (the last 12 rows are zero-filled padding: every wave issues the same number of requests)
    constexpr int BT = 256 * BK;                                                    // floats per weight tile (32 KB)
    constexpr int PT = NPIECE * 8 * BK;                                             // floats per patch chunk (23 KB)
    constexpr int GB = 4;                                                           // weight DMA instructions per wave and step
    static_assert(BK == 32, "128-B rows");
    __shared__ __attribute__((aligned(16))) float lds[3 * BT + 2 * PT];
    float* patch = lds + 3 * BT;
#ifdef AMP_STAMP
    const long long st_entry = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    int t = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = t % a.ntn; t /= a.ntn;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * 256;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);
    const int srow = lane >> 3, spos = lane & 7;      // weight DMA: lane = (row of 8, 16-B position)
    // ---- patch layout in LDS: CHUNK-MAJOR -- the 16-B piece p (0..7: hi 0-7, .., hi 24-31, lo' 0-7, ..) of patch pixel q at p * 3072 + q * 16 bytes
    // (192 pixel slots per plane).  The 16 consecutive pixels a lane group reads for one fragment are then 256 contiguous bytes -- every bank once,
    // whatever the alignment of the first pixel -- and a lane's address is (its own base) + (tap, tile row) * constant: the nine taps and four tile
    // rows are immediate offsets of ds_read_b128, where a row-major swizzled patch cost ~8 VALU instructions per fragment, issued in the half-step
    // in which the SIMD's other wave owns the vector issue with its MFMAs (stamps: 540 cycles of fragment reads per step against 251).
    // DMA piece (wave w, j): plane p = w, pixels 64 j .. 64 j + 63 -- a lane fetches 16 B of its own pixel's row (the 8 waves read the same 64 rows).
    unsigned int p_voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = 64 * j + lane;
        const int py = q / PW, px = q - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool v = q < NPIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        p_voff[j] = v ? (unsigned int)(((size_t)(b * a.H + iy) * a.W + ix) * a.Cin * 4 + (size_t)(wave * 16)) : OOB_VOFF;
    }
    auto stage_patch = [&](int j, int c) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(patch + (c & 1) * PT + wave * 768 + j * 256), 16, (int)p_voff[j], c * (BK * 4), 0, 0);
    };
    unsigned int b_voff[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * 32 + 8 * g + srow;
        const int n = n0 + (r & ~63) + swap_channel(r & 63);
        b_voff[g] = (n < a.Cout) ? (unsigned int)(((size_t)n * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
    const int csteps = a.Cin / BK;
    int s_c = 0, s_tap = 0;                  // the step being STAGED
    auto stage_w = [&](int buf) {
        const int soff = (s_tap * csteps + s_c) * (BK * 4);      // (reading the weights as [Cout][chunk][tap][32], consecutive lines step after step: no difference, measured)
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(lds + buf * BT + (wave * 32 + 8 * g) * BK), 16, (int)b_voff[g], soff, 0, 0);
        if (++s_tap == 9) { s_tap = 0; ++s_c; }
    };
    f32x4 acc[4][4], acx[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));
    const int qb = (4 * wm) * PW + l15;      // patch row of (tile row 4 wm, column l15) at tap (0, 0)

#pragma unroll
    for (int j = 0; j < 3; ++j) stage_patch(j, 0);
    stage_w(0);
    stage_w(1);
    int cur = 0, nxt = 2;
    int c_c = 0, c_tap = 0, c_toff = 0;      // the step being COMPUTED: chunk, tap, patch-row offset of the tap (ky * 18 + kx)
#ifdef AMP_STAMP
    long long st_vm = 0;
#endif
    auto open_step = [&](int step) {
        // weight tile `step` (requested two steps ago) has landed once everything younger is all that is outstanding: the patch piece of step - 2
        // (issued behind that tile's requests), the 4 weight requests of step - 1 and its patch piece -- counted exactly, an LDS-DMA request needs
        // more than one step to land (the patch itself is requested six steps or more before its chunk begins)
        if (step + 1 < a.nsteps) {
            const int extra = (c_c + 1 < csteps) ? ((c_tap >= 1 && c_tap <= 3) ? 1 : 0) + ((c_tap >= 2 && c_tap <= 4) ? 1 : 0) : 0;
            if (extra == 0) __builtin_amdgcn_s_waitcnt(0x0070 | GB);
            else if (extra == 1) __builtin_amdgcn_s_waitcnt(0x0070 | (GB + 1));
            else __builtin_amdgcn_s_waitcnt(0x0070 | (GB + 2));
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);
        }
#ifdef AMP_STAMP
        st_vm = clock64();
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    F16x3Frags<4, 4> fr;
    const float* Pl = patch + qb * 4 + lq * 768;
    auto load_frags = [&]() {
        const float* P = Pl + ((c_c & 1) * PT + c_toff * 4);
        const float* Bs = lds + cur * BT + (wn * 64 + l15) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fr.ah[i] = *reinterpret_cast<const f16x8*>(P + i * (PW * 4));
            fr.al[i] = *reinterpret_cast<const f16x8*>(P + i * (PW * 4) + 4 * 768);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fr.bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo16_hi);
            fr.bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo16_lo);
        }
    };
    auto stage_next = [&](int step) {        // in compute step `step`: weight tile step + 2, then one piece of the next chunk's patch
        if (step + 2 < a.nsteps) stage_w(nxt);
        if (c_tap < 3 && c_c + 1 < csteps) stage_patch(c_tap, c_c + 1);
    };
    auto advance = [&]() {
        cur = (cur == 2) ? 0 : cur + 1;
        nxt = (nxt == 2) ? 0 : nxt + 1;
        ++c_tap; ++c_toff;
        if (c_tap == 3 || c_tap == 6) c_toff += PW - 3;
        if (c_tap == 9) { c_tap = 0; c_toff = 0; ++c_c; }
    };
#ifdef AMP_STAMP
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_mt0 = __builtin_amdgcn_s_memtime();
    const long long st_begin = clock64();
#endif
    if (wave < 4) {
        for (int step = 0; step < a.nsteps; ++step) {
            STAMP_T(t0);
            open_step(step);
            STAMP_T(t1);
            // the DMA requests in front of the fragment reads (the other way round in conv_split_kernel): this half reaches the second barrier ~200
            // cycles before the multiplying half does, and every request issued earlier is a shorter vmcnt wait two steps on (-2 % on the layer)
            stage_next(step);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t2);
            load_frags();
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t3);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t4);
            STAMP_ADD(0, t0, t1); STAMP_ADD(1, t1, t2); STAMP_ADD(2, t2, t3); STAMP_ADD(3, t3, t4);
#ifdef AMP_STAMP
            st_acc[7] += st_vm - t0;
            if (c_tap >= 3 && c_tap <= 5) st_acc[6] += t1 - t0;
#endif
            advance();
        }
    } else {
        for (int step = 0; step < a.nsteps; ++step) {
            STAMP_T(t0);
            open_step(step);
            STAMP_T(t1);
            __builtin_amdgcn_s_setprio(1);
            if (step > 0) f16x3_mfma16<4, 4, true>(fr, acc, acx);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            STAMP_T(t2);
            stage_next(step);
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t3);
            load_frags();
            __builtin_amdgcn_sched_barrier(0);
            STAMP_T(t4);
            STAMP_ADD(0, t0, t1); STAMP_ADD(3, t1, t2); STAMP_ADD(1, t2, t3); STAMP_ADD(2, t3, t4);
#ifdef AMP_STAMP
            st_acc[7] += st_vm - t0;
            if (c_tap >= 3 && c_tap <= 5) st_acc[6] += t1 - t0;
#endif
            advance();
        }
        f16x3_mfma16<4, 4, true>(fr, acc, acx);
    }
#ifdef AMP_STAMP
    if (wave == 0 && lane == 0) {
        atomicAdd(&g_stamp_clk[0], __builtin_amdgcn_s_memtime() - st_mt0);
        atomicAdd(&g_stamp_clk[1], __builtin_amdgcn_s_memrealtime() - st_rt0);
    }
    st_acc[4] = clock64() - st_begin;
    st_acc[5] = st_begin - st_entry;
    const long long st_loop_end = clock64();
#endif
    __syncthreads();

    const bool scaled_in = a.out_scale != 1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[i][j][e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                if (scaled_in) acc[i][j][e] *= a.out_scale;
            }
    const int oyw = oy0 + 4 * wm;
    if constexpr (EPI == 3) {
        const int Ho = a.Ho, Wo = a.Wo, M = a.M;
        conv_epilogue_rpn_rows(a, acc, lds, wave, lane, n0, [=](int i, int r) {
            const int oy = oyw + i, ox = ox0 + r;
            return (oy < Ho && ox < Wo) ? (b * Ho + oy) * Wo + ox : M;
        });
    } else {
        // two tile rows at a time: the rows of an 8 x 16 tile are not consecutive GEMM rows, and four rows' hoisted residual + mask loads with
        // their own addresses do not fit the 256 registers of a wave (312 B of scratch in the epilogue)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int mrows[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = oyw + 2 * h + i, ox = ox0 + l15;
                mrows[i] = (oy < a.Ho && ox < a.Wo) ? (b * a.Ho + oy) * a.Wo + ox : a.M;
            }
            conv_epilogue_direct_rows<false, true, 2>(a, reinterpret_cast<f32x4 (&)[2][4]>(acc[2 * h]), lane, mrows, n0 + wn * 64);
        }
    }
#ifdef AMP_STAMP
    if (lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(&g_stamp[wave * 8 + q], (unsigned long long)st_acc[q]);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// conv3x3_c64_kernel (round 4): the 3x3 stride-1 convolution over a 64-channel WINDOW -- res2's dense 64 -> 64 layers and the grouped conv2
// of a ResNeXt (every 64-wide N tile reads its own 64 input channels) -- with both operands in the split row format.
// The implicit-GEMM kernels stage, per tap and K-step, the tile's (shifted) input pixels again: 18 K-steps re-read a 256-pixel tile's
// 64 channels nine times over, 1.2 GB of L2 -> LDS traffic per res2 layer for 134 MB of input, and that DMA stream, not the matrix pipe
// (0.31 of its peak), is their bound.  Here a workgroup stages the (8 + 2) x (16 + 2) pixel PATCH under its 8 x 16 output pixels once (180 pixels
// x 256 B = 46 KB, out-of-image pixels zero-filled by the buffer load) and all nine taps read their MFMA operands out of it: a 16 x 16
// MFMA block is one tile row of 16 pixels, so tap (ky, kx) is the same fragment read at patch pixel (row + ky) * 18 + kx + column.  Only the
// weights travel per tap (64 rows x 256 B, two buffers).  4 waves x (2 tile rows x 64 channels), two workgroups per CU (78 KB of LDS each).
// LDS rows are 256 B = 16 chunks of 16 B ([hi 0-31 | lo' 0-31 | hi 32-63 | lo' 32-63]); chunk c of row r sits at position c ^ (r & 15), applied
// to the SOURCE chunk of the LDS-DMA, so the 16 consecutive pixels (or weight rows) a lane group reads hit 16 different positions: conflict-free.
// Same products as every other AMP_CONV_F16X3 kernel (f16x3_mfma16), summed tap by tap (ky, kx, then the two 32-channel halves: the K order of
// the implicit-GEMM kernels), epilogue = conv_epilogue_direct_rows (FrozenBN / bias, ReLU, split ReLU mask for data gradients, split rows out).
// ------------------------------------------------------------------------------------------------------------------
// DIAG (groups of at most 32 channels): the 64 x 64 weight window is two 32 x 32 diagonal blocks -- LDS row blocks 0, 1 (channels 0-31 under swap_channel)
// see only the first 32-channel half of the window, blocks 2, 3 only the second: the other half of the products is exactly zero and is not computed
// (half the MFMAs and weight fragment reads; adding exact zeros changes no sum, so the result is the full product's bit for bit).
// FUSE3 (dense 64 -> 64 only: res2's conv2): the block's conv3 (1x1, 64 -> C3, FrozenBN, + residual, ReLU) runs in the same workgroup.  With the role-swapped MFMA
// a lane holds, per tile row, channels [8 lq, 8 lq + 8) of each 32-channel half of conv2's output -- after FrozenBN, ReLU and the split exactly the B fragment
// (pixel l15, k group lq) of a v_mfma_f32_16x16x32_f16 over that half: conv3's activation operand never leaves the registers, the 64-channel tensor t2 is
// neither written nor read back (2 x 134 MB per res2 block at B = 8) and a launch disappears.  conv3's weights (C3 rows of 256 B) are staged once into the LDS the
// patch and the tap buffers no longer need; every 64-channel chunk of the output goes through conv_epilogue_direct_rows (split residual, ReLU, split rows).
// The halves are the ones the two-launch chain stores and reloads, the products and their order conv_split_kernel's: bit-identical to the chain.
struct Fuse3Args {
    const float* w3;        // [C3][64] split rows
    const float* scale3;
    const float* shift3;
    const float* res;       // [M][C3] split rows
    float* y;               // [M][C3] split rows
    int C3;
    unsigned int w3_bytes;
};
template <bool DIAG, bool FUSE3 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(const ConvArgs a, const unsigned int x_bytes, const unsigned int w_bytes, const int tiles_x, const int tiles_y, const Fuse3Args f3) {
    constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;      // 180 patch pixels
    constexpr int ROWF = 64;                                                        // floats per LDS row (256 B)
    constexpr int NINST = NPIX / 4;                                                 // 45 DMA instructions of 4 pixels x 16 chunks
    static_assert(NPIX % 4 == 0, "patch pixels per DMA instruction");
    __shared__ __attribute__((aligned(16))) float lds[NPIX * ROWF + 2 * 64 * ROWF];
    float* patch = lds;
    float* wbuf = lds + NPIX * ROWF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    int t = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = t % a.ntn; t /= a.ntn;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * 64;
    const int cwin = a.grouped ? n0 : 0;                                            // first input channel of the window
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);
    const int sub = lane >> 4, chunk = lane & 15;                                   // DMA: lane = (row sub of 4, 16-B chunk)
    // ---- the patch, once ----
    for (int it = wave; it < NINST; it += 4) {
        const int q = it * 4 + sub;
        const int py = q / PW, px = q - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned int voff = v ? (unsigned int)((((size_t)(b * a.H + iy) * a.W + ix) * a.Cin + cwin) * 4 + (size_t)((chunk ^ (q & 15)) * 16)) : OOB_VOFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(patch + it * 4 * ROWF), 16, (int)voff, 0, 0, 0);
    }
    // ---- weights of one tap: 64 rows (LDS row r = output channel n0 + swap_channel(r)) x 256 B ----
    auto stage_w = [&](int tap, int buf) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = (wave * 4 + g) * 4 + sub;
            const int n = n0 + swap_channel(r);
            const unsigned int voff = (n < a.Cout) ? (unsigned int)(((size_t)n * a.K + (size_t)tap * 64) * 4 + (size_t)((chunk ^ (r & 15)) * 16)) : OOB_VOFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(wbuf + buf * 64 * ROWF + (wave * 4 + g) * 4 * ROWF), 16, (int)voff, 0, 0, 0);
        }
    };
    stage_w(0, 0);
    f32x4 acc[2][4], acx[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();                        // (vmcnt(0) + barrier) the patch and this tap's weights have landed; everyone is done with the other weight buffer
        if (tap + 1 < 9) stage_w(tap + 1, (tap + 1) & 1);
        const int ky = tap / 3, kx = tap - 3 * ky;
        const float* wb = wbuf + (tap & 1) * 64 * ROWF;
#pragma unroll
        for (int g = 0; g < 2; ++g) {           // the two 32-channel halves of the window
            F16x3Frags<2, 4> f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = (2 * wave + i + ky) * PW + kx + l15;
                const float* row = patch + q * ROWF;
                f.ah[i] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + lq) ^ (q & 15)));
                f.al[i] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + 4 + lq) ^ (q & 15)));
            }
            if constexpr (DIAG) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float* row = wb + ((2 * g + jj) * 16 + l15) * ROWF;
                    f.bh[jj] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + lq) ^ l15));
                    f.bl[jj] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + 4 + lq) ^ l15));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {     // f16x3_mfma16<.., SWAP>'s order per accumulator: lo'*hi, hi*lo', hi*hi
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) acx[i][2 * g + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bh[jj], f.al[i], acx[i][2 * g + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) acx[i][2 * g + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bl[jj], f.ah[i], acx[i][2 * g + jj], 0, 0, 0);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) acc[i][2 * g + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.bh[jj], f.ah[i], acc[i][2 * g + jj], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* row = wb + (j * 16 + l15) * ROWF;
                    f.bh[j] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + lq) ^ l15));
                    f.bl[j] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + 4 + lq) ^ l15));
                }
                f16x3_mfma16<2, 4, true>(f, acc, acx);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
    int mrows[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + l15;
        mrows[i] = (oy < a.Ho && ox < a.Wo) ? (b * a.Ho + oy) * a.Wo + ox : a.M;
    }
    if constexpr (!FUSE3) {
        conv_epilogue_direct_rows<false, true, 2>(a, acc, lane, mrows, n0);
    } else {
        // ---- conv2's epilogue in registers: FrozenBN, ReLU, split -> the B fragments of conv3 (what conv_epilogue_direct would have stored) ----
        f16x8 t2h[2][2], t2l[2][2];
        f32x2 chk = {0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x2 sc[4], sh[4];
            const int n = 32 * g + 8 * lq;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                sc[p] = a.scale ? f32x2{a.scale[n + 2 * p], a.scale[n + 2 * p + 1]} : f32x2{1.f, 1.f};
                sh[p] = a.shift ? f32x2{a.shift[n + 2 * p], a.shift[n + 2 * p + 1]} : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4& blk = acc[i][2 * g + (p >> 1)];
                    const f32x2 v = {blk[2 * (p & 1)], blk[2 * (p & 1) + 1]};
                    chk = __builtin_elementwise_fma(v, f32x2{0.f, 0.f}, chk);
                    f32x2 o = v * sc[p] + sh[p];
                    if (a.relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); }
                    const f16x2 h = __builtin_convertvector(o, f16x2);
                    const f32x2 hf = {(float)h[0], (float)h[1]};
                    const f32x2 l = __builtin_elementwise_fma(hf, f32x2{-LO_SCALE, -LO_SCALE}, o * LO_SCALE);
                    const f16x2 lh = __builtin_convertvector(l, f16x2);
                    t2h[i][g][2 * p] = h[0]; t2h[i][g][2 * p + 1] = h[1];
                    t2l[i][g][2 * p] = lh[0]; t2l[i][g][2 * p + 1] = lh[1];
                }
            }
        }
        if (!(chk[0] == 0.f) || !(chk[1] == 0.f)) atomicOr(a.range_flag, 1);
        // ---- conv3's weights: C3 rows x 256 B into the LDS of the patch + tap buffers (everyone is past its last tap) ----
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rsrc_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f3.w3), 0, f3.w3_bytes, 0x00020000);
        float* w3s = lds;
        const int ninst3 = f3.C3 / 4;                    // 4 rows per DMA instruction
        for (int it = wave; it < ninst3; it += 4) {
            const int r = it * 4 + sub;
            const int n = (r & ~63) + swap_channel(r & 63);
            const unsigned int voff = (unsigned int)((size_t)n * 256 + (size_t)((chunk ^ (r & 15)) * 16));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w3, (__attribute__((address_space(3))) void*)(w3s + it * 4 * ROWF), 16, (int)voff, 0, 0, 0);
        }
        __syncthreads();
        ConvArgs a3 = a;
        a3.scale = f3.scale3; a3.shift = f3.shift3; a3.res = f3.res; a3.mask = nullptr; a3.y = f3.y;
        a3.Cout = f3.C3; a3.relu = 1; a3.res_mode = 1; a3.res_split = 1; a3.y_split = 1; a3.out_mode = 0; a3.mask_split = 0;
        for (int c = 0; c < f3.C3 / 64; ++c) {
            f32x4 acc3[2][4], acx3[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc3[i][j][e] = 0.f; acx3[i][j][e] = 0.f; }
            const float* wc = w3s + c * 64 * ROWF;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                F16x3Frags<2, 4> f;
#pragma unroll
                for (int i = 0; i < 2; ++i) { f.ah[i] = t2h[i][g]; f.al[i] = t2l[i][g]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* row = wc + (j * 16 + l15) * ROWF;
                    f.bh[j] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + lq) ^ l15));
                    f.bl[j] = *reinterpret_cast<const f16x8*>(row + 4 * ((8 * g + 4 + lq) ^ l15));
                }
                f16x3_mfma16<2, 4, true>(f, acc3, acx3);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc3[i][j][e] = __fadd_rn(acc3[i][j][e], __fmul_rn(acx3[i][j][e], 1.0f / LO_SCALE));
            conv_epilogue_direct_rows<false, true, 2>(a3, acc3, lane, mrows, 64 * c);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv_f16x3_kernel (amp_set_conv_mode(ctx, AMP_CONV_F16X3)): the same implicit GEMM on the f16 matrix pipe, fp32 in / fp32 out.
// Every operand x is split as x = hi + lo with hi = f16(x) and lo = x - hi (exact in fp32); the low half is stored as
// lo' = f16(lo * 2^11), which has the magnitude of hi, so it stays a normal f16 number however small x is (22 significant bits
// down to |x| ~ 2^-25; a plain f16(lo) would go subnormal below |x| = 0.125).  a*b ~= a_hi*b_hi + (a_hi*b_lo' + a_lo'*b_hi) * 2^-11,
// every product an exact-product v_mfma_f32_32x32x16_f16 accumulated in fp32, the cross terms in their own accumulators that
// the epilogue folds in with one exact scaling (the dropped a_lo*b_lo term is < 2^-22 |a*b|): 3 MFMAs of 32 cycles do the work
// of 8 fp32 MFMAs of 64 cycles (5.3x the fp32 matrix rate).
// Weights are split once (split_weights_kernel: per K-step of 32 a row holds 64 B of hi halves then 64 B of lo' halves) and
// staged by LDS-DMA exactly like the fp32 kernel; activations stay fp32 in HBM, are fetched with bounds-checked buffer loads
// (zero fill outside the image) one K-step ahead into registers, split on the VALU and written to LDS in the same hi|lo' row
// format.  Range: |operand| must stay below 65504 (fp16 max) -- checked: a violation makes an accumulator non-finite, the
// epilogue raises ctx->d_conv_flag and the caller re-runs in fp32; the fp32 kernel has no such limit.
// ------------------------------------------------------------------------------------------------------------------

template <bool SCALED>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, f16x8& hi, f16x8& lo, float in_scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = j < 4 ? v0[j] : v1[j - 4];
        if (SCALED) x *= in_scale;
        const _Float16 h = (_Float16)x;
        hi[j] = h;
        lo[j] = (_Float16)((x - (float)h) * LO_SCALE);
    }
}

// STEM = true: the 7x7 stride-2 stem on the [B,H,W,4] input with weights [64][7][8][4] (see conv_glds_kernel): a K-step is one
// kernel row, a lane's 8 consecutive k are two taps x 4 channels, each tap bounds-checked on its own.
// SCALED = true: activations are multiplied by a.in_scale before the split and the result by a.out_scale (powers of two, exact).
// The data-gradient convolutions use it: loss gradients of 1e-9..1e-4 would sit in the f16 subnormals (the hi half keeps 11 bits
// only above 6.1e-5); scaled by 2^16 they split like activations.
// BN = 256 runs 8 waves (2 x 4, one workgroup per CU): the activation tile is fetched and split once for 256 output channels, so a
// wave carries half the split work and half the activation loads per MFMA of the 4-wave tiles.
template <int BN, int EPI, bool STEM = false, bool SCALED = false>
__global__ __launch_bounds__(BN == 256 ? 512 : 256, BN == 256 ? 1 : 2) void conv_f16x3_kernel(const ConvArgs a, const unsigned int x_bytes,
                                                                                                const unsigned int w_bytes) {
    constexpr int BM = 128;
    constexpr int WTM = BM / 2, WTN = (BN == 64) ? 32 : 64;
    constexpr int NWN = BN / WTN, NW = 2 * NWN;   // waves across N, waves per workgroup
    constexpr int RPT = 512 / (64 * NW);  // activation rows per lane (128 rows x 4 lanes per row over the workgroup)
    constexpr int GB = BN / NW / 8;       // DMA instructions per wave per step for B: BN/NW rows per wave, 8 rows each
    constexpr int TILE_FLOATS = (BM + BN) * BK;   // a row = 32 k = 64 B hi halves + 64 B lo halves = 32 dwords, as in the fp32 kernel
    constexpr int SLD = WTN + 4;
    constexpr int STAGE_FLOATS = NW * WTM * SLD;
    constexpr int LDS_FLOATS = (2 * TILE_FLOATS > STAGE_FLOATS) ? 2 * TILE_FLOATS : STAGE_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    const int tile = amp::xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = tile % a.ntn;
    const int tile_m = tile / a.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int HoWo = a.Ho * a.Wo;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);

    // ---- A (activations): lane = 4 * (row in a 16-row group) + kg; a lane owns 8 consecutive k (32 B) of rows ra[0], ra[1] ----
    const int arow = lane >> 2, akg = lane & 3;
    int a_iy0[RPT], a_ix0[RPT], a_pb[RPT], a_row[RPT];
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
        const int r = p * 64 + wave * 16 + arow;
        a_row[p] = r;
        const int m = m0 + r;
        if (m < a.M) {
            const unsigned int b = fastdiv((unsigned int)m, a.div_howo_mul, a.div_howo_shr);
            const unsigned int rem = (unsigned int)m - b * (unsigned int)HoWo;
            const unsigned int oy = fastdiv(rem, a.div_wo_mul, a.div_wo_shr);
            const unsigned int ox = rem - oy * (unsigned int)a.Wo;
            a_iy0[p] = (int)oy * a.stride - a.pad;
            a_ix0[p] = (int)ox * a.stride - a.pad;
            a_pb[p] = (int)b * a.H * a.W;
        } else {
            a_iy0[p] = -(1 << 28);
            a_ix0[p] = 0;
            a_pb[p] = 0;
        }
    }
    // ---- B (split weights): LDS-DMA, 8 rows x 128 B per wave-instruction, source chunk XOR-swizzled ----
    const int srow = lane >> 3, spos = lane & 7;
    unsigned int b_voff[GB];
#pragma unroll
    for (int g = 0; g < GB; ++g) {
        const int r = wave * (BN / NW) + 8 * g + srow;
        const int n = n0 + r;
        b_voff[g] = (n < a.Cout) ? (unsigned int)(((size_t)n * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }

    const int csteps = STEM ? 1 : a.cin_win / BK;
    const int a_win = a.grouped ? n0 * 4 : 0;   // bytes: first input channel of this N tile's window (grouped conv, BN = 64 only)
    int ky = 0, kx = 0, cs = 0, kstep = 0;     // block-uniform state of the K-step being FETCHED
    unsigned int a_voff[RPT];
    u32x4 ra[RPT][2];                             // fetched, not yet split: [row][half of the 32 B]

    auto fetch = [&](int buf) {                 // A(kstep) -> registers, B(kstep) -> LDS[buf]
        if (STEM) {
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                const int iy = a_iy0[p] + kstep;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ix = a_ix0[p] + 2 * akg + h;
                    const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                    const unsigned int vo = v ? (unsigned int)((a_pb[p] + iy * a.W + ix) * 16) : OOB_VOFF;
                    ra[p][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)vo, 0, 0);
                }
            }
        } else if (cs == 0) {
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                const int iy = a_iy0[p] + ky, ix = a_ix0[p] + kx;
                const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                a_voff[p] = v ? (unsigned int)(((a_pb[p] + iy * a.W + ix) * a.Cin + 8 * akg) * 4) : OOB_VOFF;
            }
        }
        const int a_soff = cs * (BK * 4) + a_win;
        const int b_soff = kstep * (BK * 4);
        if (!STEM) {
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                ra[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)a_voff[p], a_soff, 0);
                ra[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)a_voff[p] + 16, a_soff, 0);
            }
        }
        float* Bs = lds + buf * TILE_FLOATS + BM * BK;
#pragma unroll
        for (int g = 0; g < GB; ++g)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * (BN / NW) + 8 * g) * BK),
                                                     16, (int)b_voff[g], b_soff, 0, 0);
        ++kstep;
        if (++cs == csteps) {
            cs = 0;
            if (++kx == a.KW) { kx = 0; ++ky; }
        }
    };
    auto commit = [&](int buf) {                // split the fetched A registers into LDS[buf]
        float* As = lds + buf * TILE_FLOATS;
#pragma unroll
        for (int p = 0; p < RPT; ++p) {
            f16x8 hi, lo;
            split8<SCALED>(__builtin_bit_cast(f32x4, ra[p][0]), __builtin_bit_cast(f32x4, ra[p][1]), hi, lo, a.in_scale);
            const int r = a_row[p], sw = (r >> 1) & 7;
            *reinterpret_cast<f16x8*>(As + r * BK + 4 * (akg ^ sw)) = hi;
            *reinterpret_cast<f16x8*>(As + r * BK + 4 * ((4 + akg) ^ sw)) = lo;
        }
    };

    constexpr int MB = WTM / 16, NB = WTN / 16;
    f32x4 acc[MB][NB], acx[MB][NB];        // hi*hi sums; cross-term sums (scaled by 2^11): blocks of 16 x 16 (f16x3_step16)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }

    const int l15 = lane & 15, lq = lane >> 4;
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));

    fetch(0);
    commit(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // B(0) has landed (see the loop)
    __syncthreads();
    if (a.nsteps > 1) fetch(1);

    for (int step = 0; step < a.nsteps; ++step) {
        const int cur = step & 1;
        f16x3_step16<MB, NB>(lds + cur * TILE_FLOATS + (wm * WTM + l15) * BK, lds + cur * TILE_FLOATS + BM * BK + (wn * WTN + l15) * BK,
                             fo16_hi, fo16_lo, acc, acx);
        // A(step+1) is in registers (loads issued a step ago, behind the previous barrier), B(step+1) is landing in LDS[cur^1]
        if (step + 1 < a.nsteps) commit(cur ^ 1);
        // The weight DMA into LDS[cur^1] must have LANDED before anyone reads it.  The compiler only counts the register loads of
        // fetch() (it emitted vmcnt(2..3) here and no vmcnt(0) at the barrier: a race that showed up as soon as the weights were
        // cold in L2), so the wait is explicit: vmcnt(0), other counters untouched.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();   // LDS[cur] is free, LDS[cur^1] complete
        if (step + 2 < a.nsteps) fetch(cur);    // A(step+2) -> registers, B(step+2) -> LDS[cur]
    }

    // fold the cross terms in (their 2^-11 is exact), then the shared epilogue
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[i][j][e] = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                if (SCALED) acc[i][j][e] *= a.out_scale;
            }
    if (EPI == 0) conv_epilogue16<WTM, WTN, MB, NB, true>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN, HoWo);
    else conv_epilogue_fast16<WTM, WTN, MB, NB, EPI == 2, true>(a, acc, lds, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
}

// stem_pool_f16x3_kernel: the 7x7 stride-2 stem (conv_f16x3_kernel<64, ., STEM>) and the 3x3 stride-2 max-pool behind it in ONE kernel.
// Separately the stem writes its [B, H/2, W/2, 64] fp32 output (537 MB for a batch of 8 at 1024^2) and the pool reads it back: 1.07 GB
// through HBM and a launch for a tensor nobody else reads.  Here a workgroup owns a tile of SP_PH x SP_PW pooled pixels: its M tile is
// the (2 SP_PH + 1) x (2 SP_PW + 1) = 17 x 15 = 255 convolution outputs under that tile (256 rows of A, 8 waves as 4 x 2 wave tiles of
// 64 x 32), the accumulators take scale / shift / ReLU as in conv_epilogue_rows, land in LDS as the patch, and 448 threads take the
// maximum of the nine patch pixels of a pooled pixel, 8 channels each, and write it -- in the split hi | lo' row format when the trunk
// runs on it.  Every convolution output is the same sum in the same order as in the unfused kernel (7 K-steps of one kernel row each), so the
// pooled tensor is bit-identical; the halo costs 256 / 224 = 1.14 x the MFMA work of the stem.
constexpr int SP_PH = 8, SP_PW = 7, SP_CH = 2 * SP_PH + 1, SP_CW = 2 * SP_PW + 1;      // pooled tile, conv patch
struct StemPoolArgs {
    float* pool;            // [B][Hq][Wq][64] (fp32 rows or split rows)
    int Hq, Wq;             // pooled size
    int tiles_x, tiles_y;
    int pool_split;
};

template <bool X_SPLIT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem_pool_f16x3_kernel(const ConvArgs a, const StemPoolArgs sp, const unsigned int x_bytes,
                                                                 const unsigned int w_bytes) {
    constexpr int BM = 256, BN = 64;
    constexpr int WTM = 64, WTN = 32, NWN = 2;      // 8 waves: 4 x 2 wave tiles
    constexpr int RPT = 2;                // activation rows per lane: 256 rows x 4 lanes per row over 512 threads
    constexpr int TILE_FLOATS = (BM + BN) * BK;
    constexpr int SLD = BN + 4;           // patch row in LDS: 64 channels + pad
    static_assert(2 * TILE_FLOATS >= BM * SLD, "the patch reuses the operand buffers");
    static_assert(SP_CH * SP_CW <= BM, "the patch is the M tile");
    __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    const int tx = blockIdx.x % sp.tiles_x;
    const int tyb = blockIdx.x / sp.tiles_x;
    const int ty = tyb % sp.tiles_y, b = tyb / sp.tiles_y;
    const int oy_first = 2 * ty * SP_PH - 1, ox_first = 2 * tx * SP_PW - 1;      // conv pixel of patch row 0 / column 0

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);

    // ---- A: lane = 4 * (row in a 16-row group) + kg; a lane owns 8 consecutive k (two taps x 4 channels) of rows ra[0], ra[1] ----
    const int arow = lane >> 2, akg = lane & 3;
    int a_iy0[RPT], a_ix0[RPT], a_row[RPT];
    const int a_pb = b * a.H * a.W;
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
        const int r = p * 128 + wave * 16 + arow;
        a_row[p] = r;
        const int pr = r / SP_CW, pc = r - pr * SP_CW;
        const int oy = oy_first + pr, ox = ox_first + pc;
        const bool v = pr < SP_CH && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
        a_iy0[p] = v ? oy * a.stride - a.pad : -(1 << 28);
        a_ix0[p] = v ? ox * a.stride - a.pad : 0;
    }
    // ---- B (split weights): LDS-DMA, 8 rows x 128 B per wave-instruction, source chunk XOR-swizzled ----
    const int srow = lane >> 3, spos = lane & 7;
    unsigned int b_voff;
    {
        const int r = wave * 8 + srow;
        b_voff = (r < a.Cout) ? (unsigned int)(((size_t)r * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
    int kstep = 0;
    u32x4 ra[RPT][2];
    auto fetch = [&](int buf) {                 // A(kstep) -> registers, B(kstep) -> LDS[buf]
#pragma unroll
        for (int p = 0; p < RPT; ++p) {
            const int iy = a_iy0[p] + kstep;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ix = a_ix0[p] + 2 * akg + h;
                const bool v = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned int vo = v ? (unsigned int)((a_pb + iy * a.W + ix) * 16) : OOB_VOFF;
                ra[p][h] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)vo, 0, 0);
            }
        }
        float* Bs = lds + buf * TILE_FLOATS + BM * BK;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bs + (wave * 8) * BK), 16, (int)b_voff,
                                                 kstep * (BK * 4), 0, 0);
        ++kstep;
    };
    auto commit = [&](int buf) {                // split the fetched A registers into LDS[buf]
        float* As = lds + buf * TILE_FLOATS;
#pragma unroll
        for (int p = 0; p < RPT; ++p) {
            const int r = a_row[p], sw = (r >> 1) & 7;
            if (X_SPLIT) {      // the pixels arrive split (preprocess_run): a lane's two taps are {hi4 | lo4} twice -- regroup, no arithmetic
                const u32x4 hi = {ra[p][0][0], ra[p][0][1], ra[p][1][0], ra[p][1][1]};
                const u32x4 lo = {ra[p][0][2], ra[p][0][3], ra[p][1][2], ra[p][1][3]};
                *reinterpret_cast<u32x4*>(As + r * BK + 4 * (akg ^ sw)) = hi;
                *reinterpret_cast<u32x4*>(As + r * BK + 4 * ((4 + akg) ^ sw)) = lo;
            } else {
                f16x8 hi, lo;
                split8<false>(__builtin_bit_cast(f32x4, ra[p][0]), __builtin_bit_cast(f32x4, ra[p][1]), hi, lo, 1.0f);
                *reinterpret_cast<f16x8*>(As + r * BK + 4 * (akg ^ sw)) = hi;
                *reinterpret_cast<f16x8*>(As + r * BK + 4 * ((4 + akg) ^ sw)) = lo;
            }
        }
    };

    constexpr int MB = WTM / 16, NB = WTN / 16;
    f32x4 acc[MB][NB], acx[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }

    const int l15 = lane & 15, lq = lane >> 4;
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));

    fetch(0);
    commit(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // B(0) has landed
    __syncthreads();
    if (a.nsteps > 1) fetch(1);
    for (int step = 0; step < a.nsteps; ++step) {
        const int cur = step & 1;
        f16x3_step16<MB, NB>(lds + cur * TILE_FLOATS + (wm * WTM + l15) * BK, lds + cur * TILE_FLOATS + BM * BK + (wn * WTN + l15) * BK,
                             fo16_hi, fo16_lo, acc, acx);
        if (step + 1 < a.nsteps) commit(cur ^ 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);     // the weight DMA into LDS[cur^1] has landed (see conv_f16x3_kernel)
        __syncthreads();
        if (step + 2 < a.nsteps) fetch(cur);
    }

    // ---- the patch: fold the cross terms, scale / shift / ReLU (conv_epilogue_rows' arithmetic), -1 for pixels outside the image ----
    float* patch = lds;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = wn * WTN + j * 16 + l15;
        const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f, sh = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = wm * WTM + i * 16 + 4 * lq + e;
                const int pr = r / SP_CW, pc = r - pr * SP_CW;
                const bool v = pr < SP_CH && (unsigned)(oy_first + pr) < (unsigned)a.Ho && (unsigned)(ox_first + pc) < (unsigned)a.Wo;
                const float s = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                bad = bad || !(fabsf(s) <= 3.0e38f);
                float t = __fadd_rn(__fmul_rn(s, sc), sh);
                if (a.relu) t = fmaxf(t, 0.f);
                patch[r * SLD + n] = v ? t : -1.0f;
            }
    }
    if (bad) atomicOr(a.range_flag, 1);
    __syncthreads();

    // ---- pool: thread = (pooled pixel q, 8 channels) ----
    if (tid < SP_PH * SP_PW * 8) {
        const int q = tid >> 3, c8 = tid & 7;
        const int ppy = q / SP_PW, ppx = q - ppy * SP_PW;
        const int py = ty * SP_PH + ppy, px = tx * SP_PW + ppx;
        if (py < sp.Hq && px < sp.Wq) {
            float m[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) m[c] = -1.0f;         // (the centre of a window is always inside the image and ReLU output is >= 0)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* src = patch + ((2 * ppy + dy) * SP_CW + 2 * ppx + dx) * SLD + 8 * c8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { m[c] = fmaxf(m[c], v0[c]); m[4 + c] = fmaxf(m[4 + c], v1[c]); }
                }
            float* orow = sp.pool + ((size_t)(b * sp.Hq + py) * sp.Wq + px) * 64;
            if (sp.pool_split) {
                f16x8 hi, lo;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const _Float16 h = (_Float16)m[c];
                    hi[c] = h;
                    lo[c] = (_Float16)((m[c] - (float)h) * LO_SCALE);
                }
                const int ch = 8 * c8;
                char* base = reinterpret_cast<char*>(orow) + (ch >> 5) * 128 + (ch & 31) * 2;
                *reinterpret_cast<f16x8*>(base) = hi;
                *reinterpret_cast<f16x8*>(base + 64) = lo;
            } else {
                reinterpret_cast<f32x4*>(orow)[2 * c8] = f32x4{m[0], m[1], m[2], m[3]};
                reinterpret_cast<f32x4*>(orow)[2 * c8 + 1] = f32x4{m[4], m[5], m[6], m[7]};
            }
        }
    }
}

// w [rows][K] fp32 -> split rows: per K-step of 32, 32 f16 hi halves (64 B) then 32 f16 lo' halves (64 B)
__global__ void split_weights_kernel(const float* __restrict__ w, size_t rows, int K, unsigned int* __restrict__ out) {
    const size_t total = rows * (size_t)(K / 2);     // one thread per pair of consecutive k
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (K / 2);
        const int kp = (int)(i - r * (K / 2));       // pair index in the row
        const int step = kp / 16, j = kp % 16;       // 16 pairs per K-step
        unsigned int hw = 0, lw = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x = w[r * K + 2 * kp + q];
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)((x - (float)h) * LO_SCALE);
            hw |= (unsigned int)__builtin_bit_cast(unsigned short, h) << (16 * q);
            lw |= (unsigned int)__builtin_bit_cast(unsigned short, l) << (16 * q);
        }
        unsigned int* o = out + r * K + step * 32;
        o[j] = hw;
        o[16 + j] = lw;
    }
}

// The split copies of MANY weight tensors in one launch (a training step re-makes them all after the SGD update: one launch instead of
// one per layer, twice).  A job is one tensor; a chunk is (job, first pair) and covers up to 8192 pairs of consecutive K elements.
//   transpose = 0: out = split rows of w [N][KH*KW*C] (what split_weights_kernel writes)
//   transpose = 1: out = split rows of the data-gradient form Wt[c][KH-1-ky][KW-1-kx][n] = w[n][ky][kx][c] * scale[n] (dgrad_weight_kernel
//                  followed by split_weights_kernel, same roundings), N % 32 == 0
//   transpose = 2: the same tensor as 1 for N % 64 == 0 and C % 64 == 0; a chunk is one 64 (n) x 64 (c) tile of one tap, transposed through
//                  LDS so that both the reads (along c) and the writes (along n) are whole cache lines (the pairwise form reads one float
//                  per 4 * KH * KW * C bytes: 0.75 ms for the R50-FPN heads + trunk against 0.1 ms of traffic)
__global__ __launch_bounds__(256) void weight_jobs_kernel(const amp::WeightJob* __restrict__ jobs, const uint2* __restrict__ chunks) {
    const uint2 ch = chunks[blockIdx.x];
    const amp::WeightJob L = jobs[ch.x];
    if (L.transpose == 2) {
        __shared__ float tile[64][65];
        const int ntc = L.C / 64, ntn = L.N / 64;
        int id = (int)ch.y;
        const int tc = id % ntc; id /= ntc;
        const int tn = id % ntn;
        const int tap = id / ntn;
        const int ky = tap / L.KW, kx = tap - ky * L.KW;
        const int n0 = tn * 64, c0 = tc * 64;
        const int lc = threadIdx.x & 63, lr = threadIdx.x >> 6;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int nn = it * 4 + lr;
            float v = L.w[(((size_t)(n0 + nn) * L.KH + ky) * L.KW + kx) * L.C + c0 + lc];
            if (L.scale) v = __fmul_rn(v, L.scale[n0 + nn]);
            tile[nn][lc] = v;
        }
        __syncthreads();
        const int kyp = L.KH - 1 - ky, kxp = L.KW - 1 - kx;
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
            const int p = it * 256 + (int)threadIdx.x;
            const int cc = p >> 5, nn = (p & 31) * 2;
            unsigned int hw = 0, lw = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float x = tile[nn + q][cc];
                const _Float16 h = (_Float16)x;
                const _Float16 l = (_Float16)((x - (float)h) * LO_SCALE);
                hw |= (unsigned int)__builtin_bit_cast(unsigned short, h) << (16 * q);
                lw |= (unsigned int)__builtin_bit_cast(unsigned short, l) << (16 * q);
            }
            const size_t i = (((size_t)(c0 + cc) * L.KH + kyp) * L.KW + kxp) * L.N + n0 + nn;
            unsigned int* o = L.out + (i & ~(size_t)31);
            const int j = (int)(i & 31) >> 1;
            o[j] = hw;
            o[16 + j] = lw;
        }
        return;
    }
    const size_t npairs = (size_t)L.N * L.KH * L.KW * L.C / 2;
    for (int t = threadIdx.x; t < 8192; t += 256) {
        const size_t pi = (size_t)ch.y + t;
        if (pi >= npairs) break;
        const size_t i = 2 * pi;                       // index of the pair's first element in the output tensor
        float x[2];
        if (L.transpose) {
            const int n = (int)(i % L.N);
            size_t r = i / L.N;
            const int kxp = (int)(r % L.KW); r /= L.KW;
            const int kyp = (int)(r % L.KH);
            const int c = (int)(r / L.KH);
            const int ky = L.KH - 1 - kyp, kx = L.KW - 1 - kxp;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                x[q] = L.w[(((size_t)(n + q) * L.KH + ky) * L.KW + kx) * L.C + c];
                if (L.scale) x[q] = __fmul_rn(x[q], L.scale[n + q]);
            }
        } else {
            x[0] = L.w[i]; x[1] = L.w[i + 1];
        }
        unsigned int hw = 0, lw = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const _Float16 h = (_Float16)x[q];
            const _Float16 l = (_Float16)((x[q] - (float)h) * LO_SCALE);
            hw |= (unsigned int)__builtin_bit_cast(unsigned short, h) << (16 * q);
            lw |= (unsigned int)__builtin_bit_cast(unsigned short, l) << (16 * q);
        }
        unsigned int* o = L.out + (i & ~(size_t)31);   // rows are multiples of 32: a K-step's 32 values are 32 consecutive elements
        const int j = (int)(i & 31) >> 1;
        o[j] = hw;
        o[16 + j] = lw;
    }
}

template <int BN, bool STEM>
void launch_glds(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    switch (epi) {
        case 1: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, STEM, 1>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb); break;
        case 2: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, STEM, 2>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb); break;
        default: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, STEM, 0>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
    }
}

template <int BN>
void launch_f16x3s(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    constexpr int NT_ = (BN == 256) ? 512 : 256;
    switch (epi) {
        case 1: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, false, 1, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        case 2: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, false, 2, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        default: AMP_TIMED_LAUNCH((conv_glds_kernel<BN, false, 0, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
    }
}

void launch_f16x3s_g32(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {     // grouped, <= 32 channels per group: one K-step per tap
    switch (epi) {
        case 1: AMP_TIMED_LAUNCH((conv_glds_kernel<64, false, 1, true, true>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb); break;
        case 2: AMP_TIMED_LAUNCH((conv_glds_kernel<64, false, 2, true, true>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb); break;
        default: AMP_TIMED_LAUNCH((conv_glds_kernel<64, false, 0, true, true>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
    }
}

template <int BM, int BN>
void launch_split(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    constexpr int NT_ = (BM / 64) * (BN / 64) * 64;
    if constexpr (BM == 128 && BN == 256) {
        if (a.korder) {      // EXPERIMENT (AMP_KORDER=1): channel-major K order
            if (epi == 3) AMP_TIMED_LAUNCH((conv_split_kernel<128, 256, 3, 3, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
            else if (epi == 2) AMP_TIMED_LAUNCH((conv_split_kernel<128, 256, 2, 3, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
            else AMP_TIMED_LAUNCH((conv_split_kernel<128, 256, 1, 3, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
            return;
        }
        if (epi == 3) { AMP_TIMED_LAUNCH((conv_split_kernel<128, 256, 3>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); return; }
    }
    if (epi == 2) AMP_TIMED_LAUNCH((conv_split_kernel<BM, BN, 2>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
    else AMP_TIMED_LAUNCH((conv_split_kernel<BM, BN, 1>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
}
// the two-workgroups-per-CU form (two buffers, no ping-pong: the SIMD partner is the other workgroup's wave)
static void launch_split_short(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    if (epi == 2) AMP_TIMED_LAUNCH((conv_split_kernel<128, 128, 2, 2>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
    else AMP_TIMED_LAUNCH((conv_split_kernel<128, 128, 1, 2>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
}

// Cout = 64 (res2's 3x3 and 1x1 layers): 256 x 64 tiles, four waves of 64 x 64 (one wave spans all 64 output channels: 2/3 of the LDS bytes per
// MFMA of the 128 x 64 tiles' 64 x 32 wave tiles), two buffers, two workgroups per CU
static void launch_split_tall64(const ConvArgs& a, hipStream_t st, unsigned int xb, unsigned int wb) {
    AMP_TIMED_LAUNCH((conv_split_kernel<256, 64, 1, 2>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
}

void launch_f16x3_stem(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    if (epi == 1) AMP_TIMED_LAUNCH((conv_f16x3_kernel<64, 1, true>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
    else AMP_TIMED_LAUNCH((conv_f16x3_kernel<64, 0, true>), dim3(a.nblk), dim3(256), 0, st, a, xb, wb);
}

template <int BN>
void launch_f16x3_scaled(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    constexpr int NT_ = (BN == 256) ? 512 : 256;
    switch (epi) {
        case 1: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 1, false, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        case 2: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 2, false, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        default: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 0, false, true>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
    }
}

template <int BN>
void launch_f16x3(const ConvArgs& a, int epi, hipStream_t st, unsigned int xb, unsigned int wb) {
    if (a.in_scale != 1.0f) { launch_f16x3_scaled<BN>(a, epi, st, xb, wb); return; }
    constexpr int NT_ = (BN == 256) ? 512 : 256;
    switch (epi) {
        case 1: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 1>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        case 2: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 2>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb); break;
        default: AMP_TIMED_LAUNCH((conv_f16x3_kernel<BN, 0>), dim3(a.nblk), dim3(NT_), 0, st, a, xb, wb);
    }
}

void set_fastdiv(unsigned int d, unsigned int* mul, int* shr) {
    int l = 0;
    while ((1ull << l) < d) ++l;
    *shr = 29 + l;
    *mul = (unsigned int)(((1ull << *shr) + d - 1) / d);
}

}  // namespace

static int g_f16x3_bn256 = 1;   // EXPERIMENT switch: 256-wide 8-wave tiles where Cout % 256 == 0
extern "C" void amp_debug_set_f16x3_bn256(int v) { g_f16x3_bn256 = v; }
// one round of 128 x 256 tiles (192 ... 511 of them) is taken from this many K-steps on.  Round 4 measured the alternative's bound -- two 128 x 128 workgroups
// per CU move 64 KB per 1536 MFMA cycles through an L2 -> LDS path that gives a CU ~33 B/clk -- and AMP_WIDE_NSTEPS=16: res4's conv1 (M = 32768, K = 1024)
// 54 -> 49 us, fc2 55 -> 47 us, res4.0 conv1 32 -> 29 us in the per-launch table (bit-identical), 506.3 / 507.7 against 509.0 / 506.0 images/s for the step:
// nothing outside the noise, so the rule stays at 64
static int g_wide_nsteps = getenv("AMP_WIDE_NSTEPS") ? atoi(getenv("AMP_WIDE_NSTEPS")) : 64;
static int g_short_k_steps = getenv("AMP_SHORT_K_STEPS") ? atoi(getenv("AMP_SHORT_K_STEPS")) : 16;     // K <= 512 (2: K <= 64 only)
static int g_short_k = getenv("AMP_NO_SHORT_K") ? 0 : 1;      // EXPERIMENT switch: K <= 64 layers on 128 x 128 tiles, two workgroups per CU (0: the 128 x 256 ring tiles)
extern "C" void amp_debug_set_short_k(int v) { g_short_k = v; }
static int g_tall64 = getenv("AMP_TALL64") ? atoi(getenv("AMP_TALL64")) : 1;      // Cout = 64 layers on 256 x 64 tiles of conv_split_kernel, two workgroups per CU (0: the 128 x 64 ring tiles of conv_glds_kernel): res2 3x3 160 -> 145 us, 1x1 256 -> 64 148 -> 140 us, bit-identical
extern "C" void amp_debug_set_tall64(int v) { g_tall64 = v; }
static int g_split_ring = getenv("AMP_SPLIT_RING") ? atoi(getenv("AMP_SPLIT_RING")) : 1;    // EXPERIMENT switch: the 3-buffer conv_split_kernel for pre-split inputs (0: the 2-buffer conv_glds_kernel<.., F16>)
extern "C" void amp_debug_set_split_ring(int v) { g_split_ring = v; }
#ifdef AMP_STAMP
extern "C" int amp_debug_read_stamps(unsigned long long* out) {     // 64 values; zeroes the device counters
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    unsigned long long z[64] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
extern "C" int amp_debug_read_stamp_clock(unsigned long long* out) {     // 2 values (g_stamp_clk); zeroes the device counters
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp_clk), sizeof(unsigned long long) * 2) != hipSuccess) return -1;
    unsigned long long z[2] = {0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_clk), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
static int g_direct_epi = getenv("AMP_DIRECT_EPI") ? atoi(getenv("AMP_DIRECT_EPI")) : 1;
extern "C" void amp_debug_set_direct_epi(int v) { g_direct_epi = v; }
static int g_korder = getenv("AMP_KORDER") ? atoi(getenv("AMP_KORDER")) : 0;     // EXPERIMENT switch: 1 = channel-major K order in conv_split_kernel (ConvArgs::korder)
extern "C" void amp_debug_set_korder(int v) { g_korder = v; }
// EXPERIMENT switch, default OFF: 1 = conv3x3_patch_kernel for the wide 3x3 layers (0: conv_split_kernel<128, 256>); 2 = whatever the grid size (tests).
// Measured (round 4, tools/lab/time_conv.py, tools/lab/pmc_p256.sh): FETCH_SIZE per launch of the FPN output conv at p2 1.97 -> 0.44 M KiB, L2 misses
// 35.7 -> 11.2 M -- and the SAME wall time on random operands (1410-1420 us either way; the chip holds 1.85 GHz under the ring kernel's loop and
// 1.97-2.16 GHz under this one, which needs 2230 cycles per K-step against 1970): the layer is limited by the power the MFMAs draw, not by its
// traffic; on all-zero operands (2.4 GHz either way) the ring kernel wins by the cycle ratio, 1049 against 1124 us.  And the channel-major sums move
// one box of the full-size gate from 0.9e-3 to 1.01e-3 px off the fp32 oracle (bare tolerance 1e-3): not adopted.
static int g_patch256 = getenv("AMP_PATCH256") ? atoi(getenv("AMP_PATCH256")) : 0;
extern "C" void amp_debug_set_patch256(int v) { g_patch256 = v; }
// EXPERIMENT switch, default OFF (AMP_NLOOP=1): several N tiles per workgroup for the short-K 1x1 layers (conv1x1_nloop_kernel).  Measured (round 4,
// tools/lab/time_nloop.py, A/B in one process, bit-identical): the training step's deconv 930-970 -> 856 us, res3.0's stride-2 shortcut 315 -> 291 us at B = 16
// and 154 -> 151 at B = 8, res4.0's shortcut 223 -> 240 us (WORSE), an inference step 16.16 against 16.12 ms: what bounds these layers is not the ring's
// cold start (that was the mask tail's case: four taps over the SAME pixels and a store-free epilogue) but their bytes; 0.1 ms of a 74-ms training step
// does not pay for a second loop structure on the training path.
static int g_nloop = getenv("AMP_NLOOP") ? atoi(getenv("AMP_NLOOP")) : 0;
extern "C" void amp_debug_set_nloop(int v) { g_nloop = v; }
static int g_mask_tail_loop = getenv("AMP_NO_MASK_TAIL_LOOP") ? 0 : 1;      // EXPERIMENT switch: 0 = the fused mask-head tail on conv_split_kernel<128, 256, 3> (one tap per workgroup)
extern "C" void amp_debug_set_mask_tail_loop(int v) { g_mask_tail_loop = v; }
static int g_patch_conv = getenv("AMP_NO_PATCH_CONV") ? 0 : 1;      // EXPERIMENT switch: 0 = the implicit-GEMM kernels for the 64-channel-window 3x3 layers; 2 = conv3x3_c64_kernel whatever the grid size (tests)
extern "C" void amp_debug_set_patch_conv(int v) { g_patch_conv = v; }
static int g_stagger = getenv("AMP_STAGGER") ? atoi(getenv("AMP_STAGGER")) : 1;
extern "C" void amp_debug_set_stagger(int v) { g_stagger = v; }
static int g_conv_generic_epi = 0;   // tests: force the generic epilogue
extern "C" void amp_debug_set_conv_generic_epilogue(int on) { g_conv_generic_epi = on; }
static int g_conv_ablate = 0;   // tools/bench_conv_ablate.py: timing variants of the register-staged kernel
extern "C" void amp_debug_set_conv_ablate(int mode) { g_conv_ablate = mode; }

static int conv_impl(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w, const float* scale,
                     const float* shift, const float* res, const float* mask, float* y) {
    return amp::conv_run(ctx, d, groups, x, w, nullptr, 0, scale, shift, res, mask, y);
}

// w [rows][K] fp32 (K % 32 == 0) -> the operand layout of the AMP_CONV_F16X3 kernels, same byte size (rows * K * 4)
int amp::weight_jobs_run(amp_ctx* ctx, const amp::WeightJob* jobs_dev, const void* chunks_dev, int nchunks) {
    if (nchunks <= 0) return AMP_OK;
    hipLaunchKernelGGL(weight_jobs_kernel, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, jobs_dev, reinterpret_cast<const uint2*>(chunks_dev));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

/* The split data-gradient form of one weight tensor: wt_split = split rows (amp_split_weights) of Wt[c][KH-1-ky][KW-1-kx][n] =
 * w[n][ky][kx][c] * scale[n] (amp_dgrad_weights), bit for bit, in one pass.  Cout % 32 == 0.  A training step of the model makes these
 * for all its layers in one launch (weight_jobs_kernel); this entry runs the same kernel on one tensor (it allocates its two small
 * tables per call: for tests and bindings, not for a hot loop). */
extern "C" int amp_dgrad_weights_split(amp_ctx* ctx, const float* w, const float* scale, int Cout, int KH, int KW, int Cin, float* wt_split) {
    AMP_REQUIRE(ctx && w && wt_split && Cout > 0 && KH > 0 && KW > 0 && Cin > 0 && Cout % 32 == 0, "amp_dgrad_weights_split: bad argument (Cout %% 32 != 0?)");
    amp::WeightJob job{w, scale, reinterpret_cast<unsigned int*>(wt_split), Cout, KH, KW, Cin, (Cout % 64 == 0 && Cin % 64 == 0) ? 2 : 1};
    std::vector<unsigned int> ch;
    if (job.transpose == 2) {
        const size_t ntiles = (size_t)KH * KW * (Cout / 64) * (Cin / 64);
        for (size_t t = 0; t < ntiles; ++t) { ch.push_back(0u); ch.push_back((unsigned int)t); }
    } else {
        const size_t npairs = (size_t)Cout * KH * KW * Cin / 2;
        for (size_t p0 = 0; p0 < npairs; p0 += 8192) { ch.push_back(0u); ch.push_back((unsigned int)p0); }
    }
    amp::WeightJob* d_job = nullptr;
    unsigned int* d_ch = nullptr;
    AMP_HIP_CHECK(hipMalloc(&d_job, sizeof(job)));
    if (hipMalloc(&d_ch, ch.size() * sizeof(unsigned int)) != hipSuccess) { (void)hipFree(d_job); AMP_REQUIRE(false, "amp_dgrad_weights_split: out of device memory"); }
    int st = AMP_OK;
    if (hipMemcpy(d_job, &job, sizeof(job), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_ch, ch.data(), ch.size() * sizeof(unsigned int), hipMemcpyHostToDevice) != hipSuccess) st = AMP_ERR_HIP;
    if (st == AMP_OK) st = amp::weight_jobs_run(ctx, d_job, d_ch, (int)(ch.size() / 2));
    if (st == AMP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = AMP_ERR_HIP;     // the tables are freed below
    (void)hipFree(d_job);
    (void)hipFree(d_ch);
    return st;
}

extern "C" int amp_split_weights(amp_ctx* ctx, const float* w, long long rows, int K, float* w_split) {
    AMP_REQUIRE(ctx && w && w_split && rows > 0 && K > 0 && K % 32 == 0, "amp_split_weights: bad argument (K %% 32 != 0?)");
    hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, ctx->stream, w, (size_t)rows, K, reinterpret_cast<unsigned int*>(w_split));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

namespace {
__global__ void unsplit_rows_kernel(const unsigned int* __restrict__ in, size_t rows, int C, float* __restrict__ out) {
    const size_t total = rows * (size_t)(C / 2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / (C / 2);
        const int kp = (int)(i - r * (C / 2));
        const int step = kp / 16, j = kp % 16;
        const unsigned int hw = in[r * C + step * 32 + j], lw = in[r * C + step * 32 + 16 + j];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const _Float16 h = __builtin_bit_cast(_Float16, (unsigned short)(hw >> (16 * q)));
            const _Float16 l = __builtin_bit_cast(_Float16, (unsigned short)(lw >> (16 * q)));
            out[r * C + 2 * kp + q] = __fadd_rn((float)h, __fmul_rn((float)l, 1.0f / LO_SCALE));
        }
    }
}
}  // namespace

// the inverse of amp_split_weights / of a producer's split output: rows of hi|lo' halves -> fp32 (hi + lo' * 2^-11, exact)
extern "C" int amp_unsplit_rows(amp_ctx* ctx, const float* x_split, long long rows, int C, float* out) {
    AMP_REQUIRE(ctx && x_split && out && rows > 0 && C > 0 && C % 32 == 0, "amp_unsplit_rows: bad argument (C %% 32 != 0?)");
    hipLaunchKernelGGL(unsplit_rows_kernel, dim3(2048), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned int*>(x_split), (size_t)rows, C, out);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

extern "C" int amp_conv2d_nhwc_fmt(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w, const float* scale, const float* shift,
                                   const float* res, float* y, int fmt) {
    AMP_REQUIRE(fmt >= 0 && fmt < 8, "amp_conv2d_nhwc_fmt: bad fmt");
    AMP_REQUIRE(fmt == 0 || (ctx && ctx->conv_mode == AMP_CONV_F16X3), "amp_conv2d_nhwc_fmt: split formats exist in AMP_CONV_F16X3 only");
    return amp::conv_run(ctx, d, 1, x, w, nullptr, 0, scale, shift, res, nullptr, y, 0, fmt);
}

// Stage a11, fused: x [B,H,W,256] (a pyramid level, split rows) -> 3x3 conv 256 -> 256 + bias + ReLU -> the 16 predictor rows (3 objectness
// logits, 12 anchor deltas, one zero row; w_pred [16][256], b_pred [16]) -> pred [B*H*W][16]; the hidden tensor is never written.
// The predictor rows are split here per call (the model keeps its own split copy); B*H*W >= 24576 so that the 128 x 256 tiles fill the chip.
extern "C" int amp_rpn_head_fused(amp_ctx* ctx, const float* x_split, int B, int H, int W, const float* w_conv, const float* b_conv, const float* w_pred,
                                  const float* b_pred, float* pred) {
    AMP_REQUIRE(ctx && x_split && w_conv && b_conv && w_pred && b_pred && pred && B > 0 && H > 0 && W > 0, "amp_rpn_head_fused: bad argument");
    AMP_REQUIRE(ctx->conv_mode == AMP_CONV_F16X3, "amp_rpn_head_fused: AMP_CONV_F16X3 only (the fp32 path runs the two convolutions)");
    AMP_REQUIRE((long long)B * H * W >= 24576, "amp_rpn_head_fused: fewer than 24576 pixels: run the two convolutions (amp_conv2d_nhwc_fmt)");
    float* wps = nullptr;
    AMP_HIP_CHECK(hipMalloc(&wps, 16 * 256 * sizeof(float)));
    hipLaunchKernelGGL(split_weights_kernel, dim3(8), dim3(256), 0, ctx->stream, w_pred, (size_t)16, 256, reinterpret_cast<unsigned int*>(wps));
    amp_conv_desc d;
    d.B = B; d.H = H; d.W = W; d.Cin = 256; d.Cout = 256; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1; d.relu = 1; d.res_mode = 0; d.out_mode = 0;
    amp::RpnFuse rf{wps, b_pred, pred};
    const int st = amp::conv_run(ctx, &d, 1, x_split, w_conv, nullptr, 0, nullptr, b_conv, nullptr, nullptr, pred, 0, 1, nullptr, &rf);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(wps);
    return st;
}

// Stage a16 tail, fused: x [N,14,14,256] in the split row format -> ConvTranspose2d 2x2 s2 (w_deconv [(ky,kx,co)][256], bias [1024]: the
// bias of co repeated per tap) -> ReLU -> 1x1 predictor row of classes[n] (pred_w [K][256], pred_b [K]) -> sigmoid -> prob [N,28,28]
extern "C" int amp_mask_deconv_predict(amp_ctx* ctx, const float* x_split, int N, const float* w_deconv, const float* bias, const float* pred_w,
                                       const float* pred_b, const int* classes, int K, float* prob) {
    AMP_REQUIRE(ctx && x_split && w_deconv && bias && pred_w && pred_b && classes && prob && N >= 0 && K >= 1, "amp_mask_deconv_predict: bad argument");
    AMP_REQUIRE(ctx->conv_mode == AMP_CONV_F16X3, "amp_mask_deconv_predict: AMP_CONV_F16X3 only (the fp32 path runs the three stages)");
    if (N == 0) return AMP_OK;
    amp_conv_desc d;
    d.B = N; d.H = 14; d.W = 14; d.Cin = 256; d.Cout = 1024; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0; d.relu = 1; d.res_mode = 0; d.out_mode = 1;
    amp::PredictFuse f{pred_w, pred_b, classes, K, prob};
    return amp::conv_run(ctx, &d, 1, x_split, w_deconv, nullptr, 0, nullptr, bias, nullptr, nullptr, prob, 0, 1, &f);
}

extern "C" int amp_conv2d_nhwc_ex(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w, const float* scale,
                                  const float* shift, const float* res, const float* mask, float* y) {
    return conv_impl(ctx, d, 1, x, w, scale, shift, res, mask, y);
}

// Grouped convolution (ResNeXt conv2: Cin == Cout, groups | Cin, channels per group in {8,16,32,64}).  The kernel is the dense
// LDS-DMA kernel with 64-wide N tiles whose K range is restricted to the tile's own 64 input channels; `w_win` is the
// block-diagonal window layout [Cout][KH][KW][64] produced by amp_group_expand_weights (zeros outside the channel's group).
extern "C" int amp_conv2d_grouped_nhwc(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w_win,
                                       const float* scale, const float* shift, const float* res, float* y) {
    AMP_REQUIRE(d && groups >= 1, "amp_conv2d_grouped_nhwc: bad argument");
    return conv_impl(ctx, d, groups, x, w_win, scale, shift, res, nullptr, y);
}

// the same with split-format tensors (AMP_CONV_F16X3; fmt as in amp_conv2d_nhwc_fmt: bit 0 x, bit 1 y, bit 2 res)
extern "C" int amp_conv2d_grouped_nhwc_fmt(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w_win,
                                           const float* scale, const float* shift, const float* res, float* y, int fmt) {
    AMP_REQUIRE(d && groups >= 1 && fmt >= 0 && fmt < 8, "amp_conv2d_grouped_nhwc_fmt: bad argument");
    AMP_REQUIRE(fmt == 0 || (ctx && ctx->conv_mode == AMP_CONV_F16X3), "amp_conv2d_grouped_nhwc_fmt: split formats exist in AMP_CONV_F16X3 only");
    return amp::conv_run(ctx, d, groups, x, w_win, nullptr, 0, scale, shift, res, nullptr, y, 0, fmt);
}

namespace {
__global__ void group_expand_kernel(const float* __restrict__ w, int Cout, int taps, int cpg, float* __restrict__ out) {
    const size_t total = (size_t)Cout * taps * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 63);
        const size_t nt = i >> 6;
        const int t = (int)(nt % taps), n = (int)(nt / taps);
        const int c = (n & ~63) + j;                  // input channel of window slot j
        const int g = n / cpg;
        out[i] = (c / cpg == g) ? w[((size_t)n * taps + t) * cpg + (c - g * cpg)] : 0.f;
    }
}
}  // namespace

// w [Cout][KH][KW][cpg] (grouped OHWI) -> w_win [Cout][KH][KW][64]
extern "C" int amp_group_expand_weights(amp_ctx* ctx, const float* w, int Cout, int KH, int KW, int cpg, float* w_win) {
    AMP_REQUIRE(ctx && w && w_win && Cout > 0 && KH > 0 && KW > 0, "amp_group_expand_weights: bad argument");
    AMP_REQUIRE((cpg == 8 || cpg == 16 || cpg == 32 || cpg == 64) && Cout % 64 == 0,
                "amp_group_expand_weights: channels per group must be 8/16/32/64 and Cout a multiple of 64 (got %d, %d)", cpg, Cout);
    const size_t total = (size_t)Cout * KH * KW * 64;
    hipLaunchKernelGGL(group_expand_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, ctx->stream, w, Cout,
                       KH * KW, cpg, w_win);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

// stem_pool_u8_kernel: the fused stem + pool straight from the uint8 image.  stem_pool_f16x3_kernel fetches its A tile from the
// preprocessed [B,H,W,4] tensor once per kernel row: 2.1 GB of L2 -> LDS traffic per batch for 134 MB of pixels (every input pixel is
// the tap of ~12 convolution outputs of a tile, seven K-steps over), after a kernel that wrote those 134 MB.  Here a workgroup reads the
// 39 x 36 input pixels under its 17 x 15 convolution patch ONCE -- three bytes each, normalised and split exactly as preprocess_kernel<true>
// does -- into two LDS planes (hi halves, lo' halves: 8 B per pixel each), and the MFMA A fragments are read straight out of the planes:
// a lane's 8 consecutive k of kernel row ky are taps 2 lq, 2 lq + 1 x 4 channels = the 16 bytes at pixel (2 pr + ky, 2 pc + 2 lq) of a
// plane (stride-2 pixels: 16 lanes cover 256 contiguous bytes, conflict-free).  No A staging, no preprocess pass, only the 8-KB weight
// tile per K-step by LDS-DMA.  Same products, same 7 K-steps in the same order: the pooled tensor is bit-identical.
constexpr int SP_IH = 2 * (SP_CH - 1) + 7, SP_IW = 2 * (SP_CW - 1) + 8;      // 39 x 36 input pixels (the 8th tap of a row has zero weights)
struct StemU8Args {
    const uint8_t* img;       // [B][H][W][3] BGR
    int H, W;                 // image size (the padded size the conv sees is a.H x a.W; beyond the valid size everything is 0)
    const int* img_hw;        // optional device [B][2]: per-image valid size
    float m0, m1, m2, s0, s1, s2;
};
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem_pool_u8_kernel(const ConvArgs a, const StemPoolArgs sp, const StemU8Args u,
                                                                                                       const unsigned int w_bytes) {
    constexpr int BM = 256, BN = 64;
    constexpr int WTM = 64, WTN = 32, NWN = 2;      // 8 waves: 4 x 2 wave tiles
    constexpr int SLD = BN + 4;
    constexpr int PLANE_BYTES = SP_IH * SP_IW * 8;                 // 11 232
    constexpr int BT_FLOATS = BN * BK;                              // one weight tile: 64 rows x 128 B
    constexpr int NKS = 7;                                          // kernel rows = K-steps: ALL weight tiles are resident (57 KB), so the K loop
    constexpr int OPER_BYTES = 2 * PLANE_BYTES + NKS * BT_FLOATS * 4;   // has no barrier and no wait (a K-step is 0.2 us of MFMAs, a weight DMA 1.1 us)
    constexpr int LDS_BYTES = (OPER_BYTES > BM * SLD * 4) ? OPER_BYTES : BM * SLD * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    unsigned char* plane_hi = lds_raw;
    unsigned char* plane_lo = lds_raw + PLANE_BYTES;
    float* Bt = reinterpret_cast<float*>(lds_raw + 2 * PLANE_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int tx = blockIdx.x % sp.tiles_x;
    const int tyb = blockIdx.x / sp.tiles_x;
    const int ty = tyb % sp.tiles_y, b = tyb / sp.tiles_y;
    const int oy_first = 2 * ty * SP_PH - 1, ox_first = 2 * tx * SP_PW - 1;      // conv pixel of patch row 0 / column 0
    const int iy_first = 2 * oy_first - 3, ix_first = 2 * ox_first - 3;           // input pixel of plane row 0 / column 0

    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, w_bytes, 0x00020000);
    const int srow = lane >> 3, spos = lane & 7;
    unsigned int b_voff;
    {
        const int r = wave * 8 + srow;
        b_voff = (r < a.Cout) ? (unsigned int)(((size_t)r * a.K + 4 * (spos ^ ((r >> 1) & 7))) * 4) : OOB_VOFF;
    }
    // (no lambda here: AMP_NO_PK changes the kernel's target features, and a closure compiled with the default ones is then CALLED, not inlined)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(Bt + ks * BT_FLOATS + (wave * 8) * BK), 16, (int)b_voff,
                                                 ks * (BK * 4), 0, 0);

    // ---- the input patch: normalise (preprocess_kernel's operations) and split into the two planes ----
    {
        const int vh = u.img_hw ? u.img_hw[2 * b] : u.H, vw = u.img_hw ? u.img_hw[2 * b + 1] : u.W;
        const uint8_t* ib = u.img + (size_t)b * u.H * u.W * 3;
        constexpr int NPX = (SP_IH * SP_IW + 511) / 512;            // pixels per thread: all their byte loads are issued before the first use
        unsigned char pxb[NPX][3];
        bool pv[NPX];
#pragma unroll
        for (int t = 0; t < NPX; ++t) {
            const int q = tid + 512 * t;
            const int r = q / SP_IW, cidx = q - r * SP_IW;
            const int iy = iy_first + r, ix = ix_first + cidx;
            pv[t] = q < SP_IH * SP_IW && (unsigned)iy < (unsigned)vh && (unsigned)ix < (unsigned)vw;
            const uint8_t* px = ib + ((size_t)(pv[t] ? iy : 0) * u.W + (pv[t] ? ix : 0)) * 3;
            pxb[t][0] = px[0]; pxb[t][1] = px[1]; pxb[t][2] = px[2];
        }
#pragma unroll
        for (int t = 0; t < NPX; ++t) {
            const int q = tid + 512 * t;
            f16x4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
            if (pv[t]) {
                float v[3];
                v[0] = __fdiv_rn(__fsub_rn((float)pxb[t][0], u.m0), u.s0);
                v[1] = __fdiv_rn(__fsub_rn((float)pxb[t][1], u.m1), u.s1);
                v[2] = __fdiv_rn(__fsub_rn((float)pxb[t][2], u.m2), u.s2);
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const _Float16 h = (_Float16)v[e];
                    hi[e] = h;
                    lo[e] = (_Float16)((v[e] - (float)h) * LO_SCALE);
                }
            }
            if (q < SP_IH * SP_IW) {
                *reinterpret_cast<f16x4*>(plane_hi + q * 8) = hi;
                *reinterpret_cast<f16x4*>(plane_lo + q * 8) = lo;
            }
        }
    }

    constexpr int MB = WTM / 16, NB = WTN / 16;
    f32x4 acc[MB][NB], acx[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[i][j][e] = 0.f; acx[i][j][e] = 0.f; }
    const int l15 = lane & 15, lq = lane >> 4;
    const int fo16_hi = 4 * (lq ^ (l15 >> 1)), fo16_lo = 4 * ((4 + lq) ^ (l15 >> 1));
    int a_off[MB];                                   // byte offset of this lane's A fragment inside a plane, kernel row 0
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int r = wm * WTM + i * 16 + l15;
        const int pr = min(r / SP_CW, SP_CH - 1), pc = r - (r / SP_CW) * SP_CW;      // (row 255 is no patch pixel: it reads patch row 16, its result is dropped)
        a_off[i] = ((2 * pr) * SP_IW + 2 * pc + 2 * lq) * 8;
    }

    __builtin_amdgcn_s_waitcnt(0x0F70);     // the weight tiles have landed
    AMP_SYNCTHREADS();                        // planes and weights complete
#pragma unroll
    for (int step = 0; step < NKS; ++step) {
        const int cur = step;
        F16x3Frags<MB, NB> f;
        const int krow = step * (SP_IW * 8);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            f.ah[i] = *reinterpret_cast<const f16x8*>(plane_hi + a_off[i] + krow);
            f.al[i] = *reinterpret_cast<const f16x8*>(plane_lo + a_off[i] + krow);
        }
        const float* Bs = Bt + cur * BT_FLOATS + (wn * WTN + l15) * BK;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            f.bh[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo16_hi);
            f.bl[j] = *reinterpret_cast<const f16x8*>(Bs + j * 16 * BK + fo16_lo);
        }
        f16x3_mfma16<MB, NB>(f, acc, acx);
    }
    AMP_SYNCTHREADS();                        // every wave is done with the planes and the weights: the patch takes their place

    // ---- the patch: fold the cross terms, scale / shift / ReLU (conv_epilogue_rows' arithmetic), -1 for pixels outside the image ----
    float* patch = reinterpret_cast<float*>(lds_raw);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = wn * WTN + j * 16 + l15;
        const float sc = (a.scale && n < a.Cout) ? a.scale[n] : 1.f, sh = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = wm * WTM + i * 16 + 4 * lq + e;
                const int pr = r / SP_CW, pc = r - pr * SP_CW;
                const bool v = pr < SP_CH && (unsigned)(oy_first + pr) < (unsigned)a.Ho && (unsigned)(ox_first + pc) < (unsigned)a.Wo;
                const float s_ = __fadd_rn(acc[i][j][e], __fmul_rn(acx[i][j][e], 1.0f / LO_SCALE));
                bad = bad || !(fabsf(s_) <= 3.0e38f);
                float t = __fadd_rn(__fmul_rn(s_, sc), sh);
                if (a.relu) t = fmaxf(t, 0.f);
                patch[r * SLD + n] = v ? t : -1.0f;
            }
    }
    if (bad) (void)__hip_atomic_fetch_or(a.range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the builtin: atomicOr() would be a call here, see AMP_SYNCTHREADS)
    AMP_SYNCTHREADS();

    // ---- pool: thread = (pooled pixel q, 8 channels) ----
    if (tid < SP_PH * SP_PW * 8) {
        const int q = tid >> 3, c8 = tid & 7;
        const int ppy = q / SP_PW, ppx = q - ppy * SP_PW;
        const int py = ty * SP_PH + ppy, px = tx * SP_PW + ppx;
        if (py < sp.Hq && px < sp.Wq) {
            float m[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) m[c] = -1.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* src = patch + ((2 * ppy + dy) * SP_CW + 2 * ppx + dx) * SLD + 8 * c8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { m[c] = fmaxf(m[c], v0[c]); m[4 + c] = fmaxf(m[4 + c], v1[c]); }
                }
            float* orow = sp.pool + ((size_t)(b * sp.Hq + py) * sp.Wq + px) * 64;
            if (sp.pool_split) {
                f16x8 hi, lo;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const _Float16 h = (_Float16)m[c];
                    hi[c] = h;
                    lo[c] = (_Float16)((m[c] - (float)h) * LO_SCALE);
                }
                const int ch = 8 * c8;
                char* base = reinterpret_cast<char*>(orow) + (ch >> 5) * 128 + (ch & 31) * 2;
                *reinterpret_cast<f16x8*>(base) = hi;
                *reinterpret_cast<f16x8*>(base + 64) = lo;
            } else {
                reinterpret_cast<f32x4*>(orow)[2 * c8] = f32x4{m[0], m[1], m[2], m[3]};
                reinterpret_cast<f32x4*>(orow)[2 * c8 + 1] = f32x4{m[4], m[5], m[6], m[7]};
            }
        }
    }
}

static int g_stem_pool = getenv("AMP_NO_STEM_POOL") ? 0 : 1;      // EXPERIMENT switch: 0 = stem and pool as two kernels
extern "C" void amp_debug_set_stem_pool(int v) { g_stem_pool = v; }
// Stem (7x7 stride 2 on the [B,H,W,4] input, weights [64][7][8][4], ReLU) + max-pool 3x3 stride 2 pad 1 in one kernel: AMP_CONV_F16X3 with
// pre-split weights only.  Returns 1 (nothing launched) when the fused form does not apply -- the caller runs the two kernels.
bool amp::stem_pool_applies(amp_ctx* ctx, const float* w_split) {
    return g_stem_pool && ctx->conv_mode == AMP_CONV_F16X3 && w_split && g_conv_ablate == 0;
}
int amp::stem_pool_run(amp_ctx* ctx, int B, int H, int W, const float* x, int x_split, const float* w_split, const float* scale, const float* shift,
                       float* pool, int pool_split) {
    if (!amp::stem_pool_applies(ctx, w_split)) return 1;
    ConvArgs a = ConvArgs();
    a.x = x; a.w = w_split; a.scale = scale; a.shift = shift;
    a.B = B; a.H = H; a.W = W; a.Cin = 4; a.Cout = 64;
    a.KH = 7; a.KW = 8; a.stride = 2; a.pad = 3;
    a.Ho = (H + 2 * 3 - 7) / 2 + 1;
    a.Wo = (W + 2 * 3 - 7) / 2 + 1;
    a.cin_win = 4;
    a.K = 7 * 8 * 4; a.nsteps = 7;
    a.relu = 1;
    a.in_scale = a.out_scale = 1.0f;
    a.range_flag = ctx->d_conv_flag;
    const size_t x_bytes = (size_t)B * H * W * 4 * sizeof(float), w_bytes = (size_t)64 * a.K * sizeof(float);
    if (a.Ho < 1 || a.Wo < 1 || x_bytes >= (size_t)OOB_VOFF || (long long)B * H * W >= (1ll << 27)) {
        AMP_REQUIRE(!x_split, "stem_pool_run: a split input needs the fused kernel, which this size does not fit");
        return 1;
    }
    StemPoolArgs sp;
    sp.pool = pool; sp.pool_split = pool_split;
    sp.Hq = (a.Ho + 2 - 3) / 2 + 1; sp.Wq = (a.Wo + 2 - 3) / 2 + 1;
    sp.tiles_y = amp::cdiv(sp.Hq, SP_PH); sp.tiles_x = amp::cdiv(sp.Wq, SP_PW);
    amp_prof_rec* rec = nullptr;
    if (ctx->prof_on) {
        if (ctx->prof_used < ctx->prof_pool.size()) {
            rec = &ctx->prof_pool[ctx->prof_used++];
            rec->flops = 2.0 * (double)B * a.Ho * a.Wo * 64.0 * 7.0 * 8.0 * 4.0;   // useful work of the stem (as conv_run counts it), not the halo
            rec->variant = 1;
            rec->bytes = (double)x_bytes + (double)w_bytes + 4.0 * (double)B * sp.Hq * sp.Wq * 64.0;   // input + weights + the pooled tensor
            rec->M = B * a.Ho * a.Wo; rec->N = 64; rec->K = 7 * 8 * 4;
        } else {
            ctx->prof_truncated = true;
        }
    }
    amp::ProfLaunchScope timed(rec ? rec->e0 : nullptr, rec ? rec->e1 : nullptr);      // attached to the launch below (AMP_TIMED_LAUNCH)
    if (x_split) AMP_TIMED_LAUNCH(stem_pool_f16x3_kernel<true>, dim3((unsigned)(B * sp.tiles_y * sp.tiles_x)), dim3(512), 0, ctx->stream, a, sp,
                                    (unsigned int)x_bytes, (unsigned int)w_bytes);
    else AMP_TIMED_LAUNCH(stem_pool_f16x3_kernel<false>, dim3((unsigned)(B * sp.tiles_y * sp.tiles_x)), dim3(512), 0, ctx->stream, a, sp,
                            (unsigned int)x_bytes, (unsigned int)w_bytes);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

static int g_stem_u8 = getenv("AMP_NO_STEM_U8") ? 0 : 1;      // EXPERIMENT switch: 0 = preprocess_kernel + stem_pool_f16x3_kernel
extern "C" void amp_debug_set_stem_u8(int v) { g_stem_u8 = v; }
bool amp::stem_u8_applies(amp_ctx* ctx, const float* w_split) { return g_stem_u8 && amp::stem_pool_applies(ctx, w_split); }
// The fused stem + pool from the uint8 image (stem_pool_u8_kernel): Hp x Wp is the padded frame the convolution sees.
int amp::stem_pool_u8_run(amp_ctx* ctx, const uint8_t* img, int B, int H, int W, int Hp, int Wp, const float mean[3], const float std_[3],
                          const int* img_hw, const float* w_split, const float* scale, const float* shift, float* pool, int pool_split) {
    AMP_REQUIRE(amp::stem_u8_applies(ctx, w_split) && img && mean && std_ && pool && B > 0 && Hp >= H && Wp >= W, "stem_pool_u8_run: bad argument");
    ConvArgs a = ConvArgs();
    a.w = w_split; a.scale = scale; a.shift = shift;
    a.B = B; a.H = Hp; a.W = Wp; a.Cin = 4; a.Cout = 64;
    a.KH = 7; a.KW = 8; a.stride = 2; a.pad = 3;
    a.Ho = (Hp + 2 * 3 - 7) / 2 + 1;
    a.Wo = (Wp + 2 * 3 - 7) / 2 + 1;
    a.cin_win = 4;
    a.K = 7 * 8 * 4; a.nsteps = 7;
    a.relu = 1;
    a.in_scale = a.out_scale = 1.0f;
    a.range_flag = ctx->d_conv_flag;
    const size_t w_bytes = (size_t)64 * a.K * sizeof(float);
    StemPoolArgs sp;
    sp.pool = pool; sp.pool_split = pool_split;
    sp.Hq = (a.Ho + 2 - 3) / 2 + 1; sp.Wq = (a.Wo + 2 - 3) / 2 + 1;
    sp.tiles_y = amp::cdiv(sp.Hq, SP_PH); sp.tiles_x = amp::cdiv(sp.Wq, SP_PW);
    StemU8Args u;
    u.img = img; u.H = H; u.W = W; u.img_hw = img_hw;
    u.m0 = mean[0]; u.m1 = mean[1]; u.m2 = mean[2]; u.s0 = std_[0]; u.s1 = std_[1]; u.s2 = std_[2];
    amp_prof_rec* rec = nullptr;
    if (ctx->prof_on) {
        if (ctx->prof_used < ctx->prof_pool.size()) {
            rec = &ctx->prof_pool[ctx->prof_used++];
            rec->flops = 2.0 * (double)B * a.Ho * a.Wo * 64.0 * 7.0 * 8.0 * 4.0;   // useful work of the stem (as conv_run counts it), not the halo
            rec->variant = 1;
            rec->bytes = 3.0 * (double)B * H * W + (double)w_bytes + 4.0 * (double)B * sp.Hq * sp.Wq * 64.0;       // uint8 pixels + weights + the pooled tensor
            rec->M = B * a.Ho * a.Wo; rec->N = 64; rec->K = 7 * 8 * 4;
        } else {
            ctx->prof_truncated = true;
        }
    }
    amp::ProfLaunchScope timed(rec ? rec->e0 : nullptr, rec ? rec->e1 : nullptr);
    AMP_TIMED_LAUNCH(stem_pool_u8_kernel, dim3((unsigned)(B * sp.tiles_y * sp.tiles_x)), dim3(512), 0, ctx->stream, a, sp, u, (unsigned int)w_bytes);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

extern "C" int amp_conv2d_nhwc(amp_ctx* ctx, const amp_conv_desc* d, const float* x, const float* w,
                               const float* scale, const float* shift, const float* res, float* y) {
    return amp_conv2d_nhwc_ex(ctx, d, x, w, scale, shift, res, nullptr, y);
}

int amp::conv_run(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w, const float* w_split, int force_f32,
                  const float* scale, const float* shift, const float* res, const float* mask, float* y, int in_shift, int fmt,
                  const amp::PredictFuse* fuse, const amp::RpnFuse* rpn) {
    AMP_REQUIRE(ctx && d && x && w && y, "amp_conv2d_nhwc: null argument");
    AMP_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "amp_conv2d_nhwc: bad shape");
    AMP_REQUIRE(d->Cin % 4 == 0, "amp_conv2d_nhwc: Cin=%d must be a multiple of 4 (pad the input)", d->Cin);
    AMP_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "amp_conv2d_nhwc: bad window");
    AMP_REQUIRE(d->res_mode >= 0 && d->res_mode <= 2 && d->out_mode >= 0 && d->out_mode <= 2,
                "amp_conv2d_nhwc: bad res_mode/out_mode");
    AMP_REQUIRE(d->res_mode == 0 || res != nullptr, "amp_conv2d_nhwc: res_mode=%d needs res", d->res_mode);
    ConvArgs a;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.mask = mask; a.y = y;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    a.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    AMP_REQUIRE(a.Ho > 0 && a.Wo > 0, "amp_conv2d_nhwc: empty output");
    AMP_REQUIRE(d->res_mode != 2 || (a.Ho % 2 == 0 && a.Wo % 2 == 0),
                "amp_conv2d_nhwc: res_mode=2 needs even output size, got %dx%d", a.Ho, a.Wo);
    AMP_REQUIRE(d->out_mode != 1 || d->Cout % 4 == 0, "amp_conv2d_nhwc: out_mode=1 needs Cout %% 4 == 0");
    const long long Mll = (long long)a.B * a.Ho * a.Wo;
    AMP_REQUIRE(Mll < (1ll << 31) / 4 && (long long)a.B * a.H * a.W < (1ll << 31),
                "amp_conv2d_nhwc: tensor too large for 32-bit pixel indices");
    a.M = (int)Mll;
    a.grouped = groups > 1;
    a.cin_win = a.Cin;
    int cpg = a.Cin;
    if (a.grouped) {
        AMP_REQUIRE(d->Cin == d->Cout && d->Cin % groups == 0 && d->Cout % 64 == 0 && d->out_mode == 0,
                    "amp_conv2d_grouped_nhwc: needs Cin == Cout, a multiple of 64 and of groups, out_mode 0");
        cpg = d->Cin / groups;
        AMP_REQUIRE(cpg == 8 || cpg == 16 || cpg == 32 || cpg == 64, "amp_conv2d_grouped_nhwc: %d channels per group (8/16/32/64 supported)", cpg);
        a.cin_win = 64;
    }
    a.K = a.KH * a.KW * a.cin_win;
    a.nsteps = amp::cdiv(a.K, BK);
    a.relu = d->relu; a.res_mode = d->res_mode; a.out_mode = d->out_mode;

    set_fastdiv((unsigned int)(a.Ho * a.Wo), &a.div_howo_mul, &a.div_howo_shr);
    set_fastdiv((unsigned int)a.Wo, &a.div_wo_mul, &a.div_wo_shr);
    const int epi = (g_conv_generic_epi || (a.Cout & 3) != 0) ? 0 : ((a.res_mode == 2 || a.out_mode != 0) ? 2 : 1);

    constexpr int BM = 128;
    const int ntm = amp::cdiv(a.M, BM);
    amp_prof_rec* rec = nullptr;
    if (ctx->prof_on) {
        if (ctx->prof_used < ctx->prof_pool.size()) {
            rec = &ctx->prof_pool[ctx->prof_used++];
            rec->flops = 2.0 * (double)a.M * (double)a.Cout * (double)d->KH * (double)d->KW * (double)cpg;   // useful work
            rec->variant = 1;      // 0 = a launch of the DOMINANT kernel (set where it is launched: conv_split_kernel<128x256>, fp32 mode conv_glds_kernel<128>)
        } else {
            ctx->prof_truncated = true;
        }
    }
    amp::ProfLaunchScope timed(rec ? rec->e0 : nullptr, rec ? rec->e1 : nullptr);      // the convolution kernel launched below carries the events
    const size_t x_bytes = (size_t)a.B * a.H * a.W * a.Cin * sizeof(float);
    const size_t w_bytes = (size_t)a.Cout * a.K * sizeof(float);
    // LDS-DMA kernel: every layer but the stem (Cin = 4); buffers must stay below the out-of-range marker (2 GiB)
    const bool small = g_conv_ablate == 0 && x_bytes < (size_t)OOB_VOFF && w_bytes < (size_t)OOB_VOFF;
    const bool glds = (a.Cin % BK == 0) && small;
    AMP_REQUIRE(!a.grouped || glds, "amp_conv2d_grouped_nhwc: operands must stay below 2 GiB");
    const bool stem = a.Cin == 4 && a.KW == 8 && a.Cout <= 64 && small;   // the padded 7x7 stem
    a.range_flag = ctx->d_conv_flag;
    const bool x_is_split = (fmt & 1) != 0;
    a.y_split = (fmt & 2) ? 1 : 0;
    a.res_split = (fmt & 4) ? 1 : 0;
    a.mask_split = (fmt & 8) ? 1 : 0;
    a.stagger = g_stagger;
    a.korder = (g_korder && a.KH * a.KW > 1) ? 1 : 0;
    a.direct_epi = g_direct_epi;
    a.pred_w = a.pred_b = nullptr; a.pred_cls = nullptr; a.pred_K = 0; a.prob = nullptr;
    if (fuse) {     // the mask head's deconv with ReLU + predictor + sigmoid in its epilogue (conv_epilogue_predict)
        AMP_REQUIRE(x_is_split && d->out_mode == 1 && a.Cout == 1024 && a.KH == 1 && a.KW == 1 && a.relu && !res && !mask && !scale && g_split_ring,
                    "conv: the fused deconv-predict epilogue needs a split input, Cout = 4 x 256, ReLU and the ring kernel");
        AMP_REQUIRE(fuse->pred_w && fuse->pred_b && fuse->cls && fuse->prob && fuse->K >= 1, "conv: incomplete PredictFuse");
        a.out_mode = 3;
        a.pred_w = fuse->pred_w; a.pred_b = fuse->pred_b; a.pred_cls = fuse->cls; a.pred_K = fuse->K; a.prob = fuse->prob;
    }
    a.rpn_w = a.rpn_b = nullptr; a.rpn_pred = nullptr;
    if (rpn) {      // the RPN head's predictors in the 3x3 conv's epilogue (conv_epilogue_rpn): one 128 x 256 tile = all hidden channels of 128 pixels
        AMP_REQUIRE(!fuse && x_is_split && d->out_mode == 0 && a.Cout == 256 && a.relu && !res && !mask && g_split_ring && in_shift == 0,
                    "conv: the fused RPN epilogue needs a split input, Cout = 256, ReLU and the ring kernel");
        AMP_REQUIRE(rpn->w_split && rpn->bias && rpn->pred, "conv: incomplete RpnFuse");
        a.out_mode = 4;
        a.y_split = 0;
        a.rpn_w = rpn->w_split; a.rpn_b = rpn->bias; a.rpn_pred = rpn->pred;
    }
    if (rec) {      // algorithmic bytes of this launch: every operand once (amp_prof_launches; a 1x1 conv reads only the pixels its stride samples)
        const double in_rows = (a.KH == 1 && a.KW == 1) ? (double)a.M : (double)a.B * a.H * a.W;
        const double out_vals = a.out_mode == 3 ? (double)a.M * 4.0 : a.out_mode == 4 ? (double)a.M * 16.0 : (double)a.M * a.Cout;
        const double res_vals = !res ? 0.0 : (a.res_mode == 2 ? (double)a.M * a.Cout / 4.0 : (double)a.M * a.Cout);
        rec->bytes = 4.0 * (in_rows * a.Cin + (double)a.Cout * a.K + out_vals + res_vals + (mask ? (double)a.M * a.Cout : 0.0));
        rec->M = a.M; rec->N = a.Cout; rec->K = d->KH * d->KW * cpg;
    }
    AMP_REQUIRE(!a.mask_split || (mask != nullptr && epi != 0 && a.Cout % 32 == 0 && a.out_mode == 0), "conv: a split-format mask needs mask, Cout %% 32 == 0, out_mode 0 and a fast epilogue");
    AMP_REQUIRE(!a.res_split || (res != nullptr && epi != 0 && a.Cout % 32 == 0), "conv: a split-format residual needs res, Cout %% 32 == 0 and a fast epilogue");
    // split output: out_mode 0; or the 2x2 deconv scatter (out_mode 1) straight from the ring kernel's accumulators -- checked again where the kernel is chosen
    const bool deconv_split = a.y_split && a.out_mode == 1;
    AMP_REQUIRE(!a.y_split || ((a.out_mode == 0 || a.out_mode == 1) && a.Cout % 32 == 0 && epi != 0), "conv: split output needs out_mode 0 (or the deconv scatter) and Cout %% 32 == 0");
    AMP_REQUIRE(!deconv_split || ((a.Cout / 4) % 64 == 0 && !res && !mask && x_is_split && g_split_ring && g_direct_epi && a.Cout % 256 == 0),
                "conv: a split deconv output needs Cout / 4 %% 64 == 0, a split input, no residual / mask and the ring kernel's direct epilogue");
    AMP_REQUIRE(!x_is_split || (ctx->conv_mode == AMP_CONV_F16X3 && !force_f32 && a.Cin % 32 == 0 && glds),
                "conv: a split-format input needs AMP_CONV_F16X3, Cin %% 32 == 0 and operands below 2 GiB");
    a.in_scale = (in_shift != 0) ? ldexpf(1.0f, in_shift) : 1.0f;
    a.out_scale = (in_shift != 0) ? ldexpf(1.0f, -in_shift) : 1.0f;
    if (ctx->conv_mode == AMP_CONV_F16X3 && !force_f32 && g_conv_ablate == 0 && (glds || stem)) {
        if (!w_split) {   // per-call split into the context's scratch (stream order makes the reuse safe)
            if (ctx->split_bytes < w_bytes) {
                AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (ctx->split_scratch) AMP_HIP_CHECK(hipFree(ctx->split_scratch));
                ctx->split_scratch = nullptr; ctx->split_bytes = 0;
                AMP_HIP_CHECK(hipMalloc(&ctx->split_scratch, w_bytes));
                ctx->split_bytes = w_bytes;
            }
            hipLaunchKernelGGL(split_weights_kernel, dim3(2048), dim3(256), 0, ctx->stream, w, (size_t)a.Cout, a.K,
                               reinterpret_cast<unsigned int*>(ctx->split_scratch));
            w_split = ctx->split_scratch;
        }
        a.w = w_split;
        const int nblk128 = ntm * amp::cdiv(a.Cout, 128);
        // 256-wide tiles (8 waves, one workgroup per CU) when they fill the chip twice -- or once, if the K loop is long enough to
        // amortise a single round (fc1: M = 8000, K = 12544: 64-wide tiles re-read the 400 MB activation matrix from HBM)
        const int nblk256 = (a.Cout % 256 == 0) ? ntm * (a.Cout / 256) : 0;
        const bool wide256 = g_f16x3_bn256 && !a.grouped && !stem && (nblk256 >= 512 || (nblk256 >= 192 && a.nsteps >= g_wide_nsteps));
        const int ntm256 = amp::cdiv(a.M, 256);
        // a split input that carries a 2^in_shift (scaled loss gradients): only the ring kernel undoes it (a.out_scale in its fold)
        AMP_REQUIRE(!(x_is_split && in_shift != 0) || (g_split_ring && epi != 0 && wide256 && a.out_mode != 3),
                    "conv: a scaled split input needs a layer the 128 x 256 ring kernel takes (Cout %% 256 == 0, enough tiles)");
        // N tiles per workgroup for the short-K 1x1 layers on the ring kernel (0: not such a layer): as many as leave a full round of workgroups
        int nloop_nt = 0;
        if (a.KH == 1 && a.KW == 1 && a.pad == 0 && !a.grouped && a.Cout % 256 == 0 && a.Cout >= 512 && a.Cin % BK == 0 && a.nsteps >= 2 && a.nsteps <= 16 && !a.korder &&
            a.y_split && g_direct_epi && !mask && a.res_mode == 0 && (a.out_mode == 0 || a.out_mode == 1)) {      // (no residual, no mask: see conv_epilogue_direct_rows PLAIN)
            const int ntn_ = a.Cout / 256;
            nloop_nt = ntn_ < 8 ? ntn_ : 8;
            while (nloop_nt > 1 && (ntn_ % nloop_nt != 0 || (long long)ntm * (ntn_ / nloop_nt) < 256)) --nloop_nt;
        }
        const long long p256_tiles = (long long)a.B * amp::cdiv(a.Ho, 8) * amp::cdiv(a.Wo, 16) * (a.Cout / 256);
        const bool patch256_ok = g_patch256 != 0 && x_is_split && g_split_ring && epi != 0 && !a.grouped && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 &&
            a.Cin % 32 == 0 && a.Cin >= 64 && a.Cout % 256 == 0 && a.nsteps == 9 * (a.Cin / 32) &&
            (a.out_mode == 4 || (a.out_mode == 0 && a.y_split && g_direct_epi && (!mask || a.mask_split) && (a.res_mode == 0 || (a.res_mode == 1 && a.res_split)))) &&
            (g_patch256 == 2 || ((p256_tiles >= 512 || (p256_tiles >= 192 && a.nsteps >= 64)) && p256_tiles * 100 <= (long long)nblk256 * 108));
        if (g_patch_conv != 0 && x_is_split && a.y_split && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.cin_win == 64 && (a.grouped || (a.Cin == 64 && a.Cout == 64)) &&
            a.out_mode == 0 && a.res_mode == 0 && epi != 0 && g_direct_epi && (!mask || a.mask_split) && a.in_scale == 1.0f && a.Cout % 64 == 0 &&
            (g_patch_conv == 2 || (long long)a.B * amp::cdiv(a.Ho, 8) * amp::cdiv(a.Wo, 16) * (a.Cout / 64) >= 512)) {
            // res2's dense 64 -> 64 layers and the ResNeXt conv2: a pixel patch staged once, nine taps read out of it (conv3x3_c64_kernel)
            const int tiles_x = amp::cdiv(a.Wo, 16), tiles_y = amp::cdiv(a.Ho, 8);
            a.ntn = a.Cout / 64; a.nblk = a.B * tiles_x * tiles_y * a.ntn;
            const Fuse3Args nofuse = Fuse3Args();
            if (a.grouped && cpg <= 32) AMP_TIMED_LAUNCH(conv3x3_c64_kernel<true>, dim3(a.nblk), dim3(256), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, tiles_x, tiles_y, nofuse);
            else AMP_TIMED_LAUNCH(conv3x3_c64_kernel<false>, dim3(a.nblk), dim3(256), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, tiles_x, tiles_y, nofuse);
        } else
        if (patch256_ok) {
            // FPN output / RPN / res4 / res5 3x3: 8 x 16 pixel tiles, the patch of a 32-channel chunk staged once for its nine taps (conv3x3_patch_kernel)
            a.ntn = a.Cout / 256; a.nblk = (int)p256_tiles;
            if (rec) rec->variant = 0;
            const int tiles_x = amp::cdiv(a.Wo, 16), tiles_y = amp::cdiv(a.Ho, 8);
            if (a.out_mode == 4) AMP_TIMED_LAUNCH(conv3x3_patch_kernel<3>, dim3(a.nblk), dim3(512), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, tiles_x, tiles_y);
            else AMP_TIMED_LAUNCH(conv3x3_patch_kernel<1>, dim3(a.nblk), dim3(512), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, tiles_x, tiles_y);
        } else
        if (a.out_mode == 3 && g_mask_tail_loop && a.Cin % BK == 0 && !a.korder && a.Ho * a.Wo >= 128 /* a 128-pixel block spans at most two RoIs */) {     // fused mask-head tail: the four taps of a pixel block in one workgroup
            a.ntn = 1; a.nblk = ntm;      // (a kernel of its own: not tagged as a launch of the dominant kernel)
            AMP_TIMED_LAUNCH(mask_tail_kernel, dim3(a.nblk), dim3(512), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (a.out_mode == 3) {                                        // ... one (pixel block, tap) tile per workgroup on the 128 x 256 ring kernel
            a.ntn = 4; a.nblk = ntm * 4;
            if (rec) rec->variant = 0;
            launch_split<128, 256>(a, 3, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (a.out_mode == 4) {                                        // fused RPN tail: one N tile
            a.ntn = 1; a.nblk = ntm;
            if (rec) rec->variant = 0;
            launch_split<128, 256>(a, 3, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (x_is_split && g_split_ring && g_short_k && epi != 0 && !a.grouped && a.nsteps <= g_short_k_steps && res && a.res_mode == 1 && a.res_split && a.y_split && !mask &&
                   a.Cout % 128 == 0 && ntm * (a.Cout / 128) >= 1024) {
            // the trunk's conv3 + residual (K = 64 ... 256 into 4 K channels: byte-bound): 128 x 128 tiles on TWO buffers -- tile s + 2 goes into
            // the buffer of tile s once every wave holds its fragments, same prefetch distance as the three-buffer ring -- so that TWO
            // workgroups fit a CU and one's residual loads and stores overlap the other's staging: res2 -5 %, res3 -8 %, res4 -3 % on these
            // launches (A/B in one call; bit-identical); without a residual it is 4 % slower
            a.ntn = a.Cout / 128; a.nblk = ntm * a.ntn;
            a.stagger = 0;
            launch_split_short(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (g_nloop && nloop_nt >= 2 && x_is_split && g_split_ring && epi != 0 && (wide256 || deconv_split)) {
            // short-K 1x1 layers: several N tiles of a pixel block in one workgroup, the ring carried across them (conv1x1_nloop_kernel)
            a.ntn = a.Cout / 256; a.nblk = ntm * (a.ntn / nloop_nt);
            if (a.out_mode == 1) AMP_TIMED_LAUNCH(conv1x1_nloop_kernel<true>, dim3(a.nblk), dim3(512), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, nloop_nt);
            else AMP_TIMED_LAUNCH(conv1x1_nloop_kernel<false>, dim3(a.nblk), dim3(512), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, nloop_nt);
        } else if (x_is_split && g_split_ring && epi != 0 && (wide256 || deconv_split)) {            // 128 x 256 tiles, 3-buffer ring
            // (the K = 256 deconv with its scatter epilogue on two-buffer 128 x 128 tiles, two workgroups per CU: 1009 against 931 us -- measured, not kept)
            a.ntn = a.Cout / 256; a.nblk = ntm * a.ntn;
            if (rec) rec->variant = 0;
            launch_split<128, 256>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (x_is_split && g_split_ring && epi != 0 && !a.grouped && a.Cout % 128 == 0 && (ntm256 * (a.Cout / 128) >= 512 || (ntm256 * (a.Cout / 128) >= 192 && a.nsteps >= 64))) {
            a.ntn = a.Cout / 128; a.nblk = ntm256 * a.ntn;                  // Cout = 128 (or 384, ...): 256 x 128 tiles
            launch_split<256, 128>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (x_is_split && g_split_ring && g_tall64 && epi == 1 && !a.grouped && a.Cout == 64 && in_shift == 0 && amp::cdiv(a.M, 256) >= 1024) {
            a.ntn = 1; a.nblk = amp::cdiv(a.M, 256);
            a.stagger = 0;
            launch_split_tall64(a, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (x_is_split) {      // both operands by LDS-DMA
            if (wide256) {
                a.ntn = a.Cout / 256; a.nblk = ntm * a.ntn;
                launch_f16x3s<256>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
            } else if (!a.grouped && a.Cout > 64 && nblk128 >= 512) {
                a.ntn = amp::cdiv(a.Cout, 128); a.nblk = ntm * a.ntn;
                launch_f16x3s<128>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
            } else {    // (grouped: the window of a 64-wide N tile is the tile's own 64 input channels = 256 B of a split row too)
                a.ntn = amp::cdiv(a.Cout, 64); a.nblk = ntm * a.ntn;
                static const bool no_g32 = getenv("AMP_NO_G32") != nullptr;      // EXPERIMENT switch: two K-steps per tap for every grouped layer
                if (a.grouped && cpg <= 32 && !no_g32) launch_f16x3s_g32(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
                else launch_f16x3s<64>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
            }
        } else if (stem) {
            a.ntn = 1;
            a.nblk = ntm;
            launch_f16x3_stem(a, epi == 2 ? 0 : epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (a.grouped) {     // the window of a 64-wide N tile is the tile's own 64 input channels
            a.ntn = a.Cout / 64;
            a.nblk = ntm * a.ntn;
            launch_f16x3<64>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (wide256) {
            a.ntn = a.Cout / 256;
            a.nblk = ntm * a.ntn;
            launch_f16x3<256>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else if (a.Cout > 64 && nblk128 >= 512) {
            a.ntn = amp::cdiv(a.Cout, 128);
            a.nblk = ntm * a.ntn;
            launch_f16x3<128>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else {
            a.ntn = amp::cdiv(a.Cout, 64);
            a.nblk = ntm * a.ntn;
            launch_f16x3<64>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        }
    } else if (stem) {
        a.ntn = 1;
        a.nblk = ntm;
        launch_glds<64, true>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
    } else if (glds) {
        const int nblk128 = ntm * amp::cdiv(a.Cout, 128);
        // BN = 64 also for wide layers whose 128-wide grid would leave CUs idle (2 workgroups fit per CU)
        if (!a.grouped && a.Cout > 64 && nblk128 >= 512) {
            a.ntn = amp::cdiv(a.Cout, 128);
            a.nblk = ntm * a.ntn;
            if (rec) rec->variant = 0;
            launch_glds<128, false>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        } else {
            a.ntn = amp::cdiv(a.Cout, 64);
            a.nblk = ntm * a.ntn;
            launch_glds<64, false>(a, epi, ctx->stream, (unsigned int)x_bytes, (unsigned int)w_bytes);
        }
    } else if (a.Cout > 64) {
        a.ntn = amp::cdiv(a.Cout, 128);
        a.nblk = ntm * a.ntn;
        switch (g_conv_ablate) {
            case 1: AMP_TIMED_LAUNCH((conv_mfma_kernel<BM, 128, 1>), dim3(a.nblk), dim3(256), 0, ctx->stream, a); break;
            case 2: AMP_TIMED_LAUNCH((conv_mfma_kernel<BM, 128, 2>), dim3(a.nblk), dim3(256), 0, ctx->stream, a); break;
            case 3: AMP_TIMED_LAUNCH((conv_mfma_kernel<BM, 128, 3>), dim3(a.nblk), dim3(256), 0, ctx->stream, a); break;
            default: AMP_TIMED_LAUNCH((conv_mfma_kernel<BM, 128, 0>), dim3(a.nblk), dim3(256), 0, ctx->stream, a);
        }
    } else {
        a.ntn = 1;
        a.nblk = ntm;
        AMP_TIMED_LAUNCH((conv_mfma_kernel<BM, 64>), dim3(a.nblk), dim3(256), 0, ctx->stream, a);
    }
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}


// conv2 (3x3, 64 -> 64, FrozenBN, ReLU) + conv3 (1x1, 64 -> C3, FrozenBN, + residual, ReLU) of a res2 bottleneck in one launch (conv3x3_c64_kernel<false, true>):
// every operand in the split row format; returns 1 when the fused kernel does not apply (the caller launches the two convolutions).
static int g_fuse23 = getenv("AMP_NO_FUSE23") ? 0 : 1;      // EXPERIMENT switch
extern "C" void amp_debug_set_fuse23(int v) { g_fuse23 = v; }
int amp::conv_c64_fused3_run(amp_ctx* ctx, int B, int H, int W, const float* x_split, const float* w2_split, const float* scale2, const float* shift2,
                             const float* w3_split, const float* scale3, const float* shift3, int C3, const float* res_split, float* y_split) {
    if (!g_fuse23 || g_patch_conv == 0 || !g_direct_epi || ctx->conv_mode != AMP_CONV_F16X3 || !w2_split || !w3_split) return 1;
    if (C3 % 64 != 0 || C3 > 256 || B < 1 || H < 1 || W < 1) return 1;
    const size_t x_bytes = (size_t)B * H * W * 64 * sizeof(float), w_bytes = (size_t)64 * 576 * sizeof(float), r_bytes = (size_t)B * H * W * C3 * sizeof(float);
    const long long tiles = (long long)B * amp::cdiv(H, 8) * amp::cdiv(W, 16);
    if (x_bytes >= (size_t)OOB_VOFF || r_bytes >= ((size_t)1 << 32) || (long long)B * H * W >= (1ll << 29) / 4 || (g_patch_conv != 2 && tiles < 512)) return 1;
    ConvArgs a = ConvArgs();
    a.x = x_split; a.w = w2_split; a.scale = scale2; a.shift = shift2;
    a.B = B; a.H = H; a.W = W; a.Cin = 64; a.Cout = 64; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.Ho = H; a.Wo = W;
    a.M = B * H * W; a.K = 576; a.nsteps = 18; a.cin_win = 64; a.grouped = 0;
    a.relu = 1; a.in_scale = a.out_scale = 1.0f; a.y_split = 1; a.direct_epi = 1;
    a.range_flag = ctx->d_conv_flag;
    a.ntn = 1; a.nblk = (int)tiles;
    set_fastdiv((unsigned int)(a.Ho * a.Wo), &a.div_howo_mul, &a.div_howo_shr);
    set_fastdiv((unsigned int)a.Wo, &a.div_wo_mul, &a.div_wo_shr);
    Fuse3Args f3;
    f3.w3 = w3_split; f3.scale3 = scale3; f3.shift3 = shift3; f3.res = res_split; f3.y = y_split; f3.C3 = C3; f3.w3_bytes = (unsigned int)((size_t)C3 * 64 * sizeof(float));
    amp_prof_rec* rec = nullptr;
    if (ctx->prof_on) {
        if (ctx->prof_used < ctx->prof_pool.size()) {
            rec = &ctx->prof_pool[ctx->prof_used++];
            rec->flops = 2.0 * (double)a.M * (64.0 * 576.0 + (double)C3 * 64.0);
            rec->variant = 1;
            rec->bytes = (double)x_bytes + (double)w_bytes + (double)f3.w3_bytes + 2.0 * (double)r_bytes;      // input + weights + residual + output (t2 is never in memory)
            rec->M = a.M; rec->N = C3; rec->K = 576 + 64;
        } else {
            ctx->prof_truncated = true;
        }
    }
    amp::ProfLaunchScope timed(rec ? rec->e0 : nullptr, rec ? rec->e1 : nullptr);
    AMP_TIMED_LAUNCH((conv3x3_c64_kernel<false, true>), dim3(a.nblk), dim3(256), 0, ctx->stream, a, (unsigned int)x_bytes, (unsigned int)w_bytes, amp::cdiv(W, 16), amp::cdiv(H, 8), f3);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

// C-ABI of the fused block tail (tests, and a host that drives res2 itself): all tensors split rows; AMP_ERR_STATE when the kernel does not apply.
extern "C" int amp_bottleneck64_tail(amp_ctx* ctx, int B, int H, int W, const float* x_split, const float* w2_split, const float* scale2, const float* shift2,
                                     const float* w3_split, const float* scale3, const float* shift3, int C3, const float* res_split, float* y_split) {
    AMP_REQUIRE(ctx && x_split && w2_split && w3_split && res_split && y_split, "%s", "amp_bottleneck64_tail: null argument");
    const int st = amp::conv_c64_fused3_run(ctx, B, H, W, x_split, w2_split, scale2, shift2, w3_split, scale3, shift3, C3, res_split, y_split);
    if (st == 1) { amp::set_error("amp_bottleneck64_tail: the fused kernel does not apply (AMP_CONV_F16X3, C3 %% 64 == 0, C3 <= 256, >= 512 tiles of 8 x 16)"); return AMP_ERR_STATE; }
    return st;
}
