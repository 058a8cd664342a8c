// Shared helpers for the ampis_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#ifdef __HIPCC__
#include <hip/hip_ext.h>      // hipExtLaunchKernelGGL (AMP_TIMED_LAUNCH); not a header for the host-only sanitizer build (tests/test_sanitize.py)
#endif
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
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

#define AMP_TRY_STATUS(expr) do { int _s = (expr); if (_s != AMP_OK) return _s; } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__   // device helpers: absent from the host-only sanitizer build of the RLE codec (tests/sanitize, plain g++)
// Bijective XCD-aware block remap: blocks b and b+8 share an XCD under the observed
// round-robin dispatch, so hand each XCD a contiguous chunk of the logical grid
// (neighbouring tiles share operand panels in that XCD's L2). Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Order-preserving float <-> uint32 (ascending float order == ascending unsigned order).
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;   // -0.0 ties with +0.0 (as in torch / numpy comparisons)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// 64-bit sort word: [63:32] ordered score, [31:8] ~position (ties -> smaller position first when sorting
// descending), [7:0] category payload (below the tie-break, positions are unique). 0 = invalid / unused slot.
__device__ __forceinline__ unsigned long long make_sortkey(uint32_t ord_score, int pos, int cat) {
    unsigned long long k = ((unsigned long long)ord_score << 32) |
                           ((unsigned long long)(0xffffffu - (uint32_t)pos) << 8) | (uint32_t)(cat & 0xff);
    if ((k >> 32) == 0ull) k |= (1ull << 32);
    return k;
}
__device__ __forceinline__ int sortkey_pos(unsigned long long k) { return (int)(0xffffffu - (uint32_t)((k >> 8) & 0xffffffu)); }
__device__ __forceinline__ int sortkey_cat(unsigned long long k) { return (int)(k & 0xffu); }
#endif  // __HIPCC__

}  // namespace amp

// Live profile (amp_prof_begin): the timed kernels are launched with their start / stop events ATTACHED to the dispatch
// (hipExtLaunchKernelGGL) instead of a hipEventRecord in front of and behind them: a recorded event is a packet of its own and cost
// ~5 us of idle GPU each -- 10 us per convolution, 0.94 ms on every sampled step of the bench.  amp::prof_e0 / prof_e1 are the events of
// the launch being issued (null outside a profile: a plain launch).
namespace amp {
extern thread_local hipEvent_t prof_e0, prof_e1;
}
namespace amp {
struct ProfLaunchScope {          // attaches a profile record's events to the timed launches of the enclosing scope
    explicit ProfLaunchScope(hipEvent_t e0, hipEvent_t e1) { prof_e0 = e0; prof_e1 = e1; }
    ~ProfLaunchScope() { prof_e0 = prof_e1 = nullptr; }
};
}
#define AMP_TIMED_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, amp::prof_e0, amp::prof_e1, 0, __VA_ARGS__)

// __syncthreads() spelled with builtins: inside an AMP_NO_PK kernel the header's inline function has other target features than the
// kernel and is CALLED instead of inlined
#define AMP_SYNCTHREADS() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); \
                               __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
// Packed FP32 on this hardware (rounds 3-4; tools/pk_probe/pk_opsel_probe.hip, tools/_probe ISA variants): with kernels of OTHER queues
// running on the card, v_pk_add / v_pk_mul / v_pk_fma_f32 whose op_sel picks src1's HIGH dword for the LOW result read that operand as 0 in
// one quarter wave (~1e-4 of the executions; never alone; every other operand form, v_fma_mix_f32 and v_pk_mov_b32: 0 of 1.7e9).  Round 3
// saw it as box coordinates replaced by the box centre (box_candidates_kernel: the SLP vectoriser's `x2 = cx + 0.5 w` was such an
// instruction) and blamed VALU-mask wait states; hand-edited ISA refuted that.  The fence is at build level: files of scalar fp32 code are
// compiled with -fno-slp-vectorize (csrc/Makefile NOSLP), hand-packed code never shuffles src1, and tests/test_isa_hazard.py scans the
// listing of every file, built with the Makefile's own flags, for the form (tools/scan_pk_opsel.py).
// AMP_NO_PK (a kernel without any packed-FP32 instruction) was round 3's fix; it made the HIP headers' inline functions CALLS inside the
// kernel (other target features) and is no longer used.  Kept for experiments.
#if defined(__HIP_DEVICE_COMPILE__)
#define AMP_NO_PK __attribute__((target("no-packed-fp32-ops")))
#else
#define AMP_NO_PK
#endif

#include <vector>
struct amp_comm;   // comm.hip: RCCL communicator + its stream and events
struct amp_prof_rec { hipEvent_t e0, e1; double flops; int variant; double bytes = 0; int M = 0, N = 0, K = 0; };   // bytes: algorithmic (every operand once)
struct amp_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    hipEvent_t ev0, ev1;
    float* zero_page = nullptr;            // 256 zero bytes: source of out-of-image taps for the LDS-DMA conv
    // live kernel profile (amp_prof_begin/end): HIP-event pairs around every conv launch on this stream
    bool prof_on = false;
    std::vector<amp_prof_rec> prof_pool;   // pre-created events
    size_t prof_used = 0;
    bool prof_truncated = false;
    // convolution arithmetic (amp_set_conv_mode): AMP_CONV_F32 = fp32 MFMA, AMP_CONV_F16X3 = split-operand f16 MFMA (default)
    int conv_mode = 1;
    int* d_conv_flag = nullptr;            // device int: an f16x3 convolution produced a non-finite accumulator (operand beyond fp16 range)
    float* split_scratch = nullptr;        // split copy of the weights of a per-call f16x3 convolution
    void* topk_scratch = nullptr;          // chunk candidates of amp_rpn_topk (levels cut into several workgroups)
    size_t topk_bytes = 0;
    size_t split_bytes = 0;
    amp_comm* comm = nullptr;              // amp_comm_init: RCCL communicator of this context (one rank per context)
    int* roi_order = nullptr;              // roi_align.hip: XCD-major processing order of the RoIs of a call (8 x (R/8 + 64) + 8 ints)
    size_t roi_order_ints = 0;
    // wgrad row tables (wgrad.hip): they depend on the layer geometry only, so a training loop computes each once, not once per step
    // ONE arena per context, allocated at the first weight-gradient call (not inside a later step), in two generations: tables are bump-allocated
    // in the current generation; when it is full the OTHER generation's tables are dropped and it becomes the current one -- what was used
    // during the last generation survives, memory is bounded, no allocation or free ever happens in a training step (stream order makes the
    // re-use safe: tables are written and read on `stream` only).  amp_debug_rowtab_stats reports hits / misses / resident bytes.
    struct RowTab { int key[9]; unsigned int* tab; size_t n; int gen; };
    std::vector<RowTab> rowtabs;
    unsigned int* rowtab_arena = nullptr;
    size_t rowtab_gen_words = 0;           // capacity of one generation, in 4-byte words (AMP_ROWTAB_MB: total arena, default 512 MiB)
    size_t rowtab_used[2] = {0, 0};        // words used in each generation
    int rowtab_gen = 0;
    unsigned long long rowtab_hits = 0, rowtab_misses = 0, rowtab_flushes = 0;
    // weight-gradient reductions on a second stream (wgrad.hip amp::wgrad_async_*): the MFMA kernel of a layer stays on `stream`, its
    // slab reduction runs on `side` behind an event, two scratch buffers alternate; the next layers' kernels do not wait for it
    bool reduce_async = false;
    hipStream_t side = nullptr;
    hipEvent_t wg_ev[2] = {nullptr, nullptr};      // the MFMA kernel that wrote scratch p is done
    hipEvent_t side_ev[2] = {nullptr, nullptr};    // the reduction that read scratch p is done
    bool side_used[2] = {false, false};
    int side_parity = 0, side_last = -1;
    float* side_scratch1 = nullptr;                // the second scratch buffer (the first is the one the calls pass)
};

namespace amp {
// internal convolution entry (conv.hip): w_split = weights already in the f16x3 split layout (amp_split_weights) or null;
// force_f32 = 1 runs the fp32-MFMA kernel whatever the context mode is; in_shift = s: (AMP_CONV_F16X3 only) the input is multiplied
// by 2^s before the operand split and the sum by 2^-s (data gradients: tiny values would otherwise sit in the f16 subnormals).
struct PredictFuse {              // the mask head's tail fused into the deconv's epilogue (conv.hip conv_epilogue_predict)
    const float* pred_w;          // [K][256] predictor weights, fp32
    const float* pred_b;          // [K]
    const int* cls;               // [N] class of each RoI
    int K;
    float* prob;                  // [N][28][28]
};
struct RpnFuse {                  // the RPN head's 1x1 predictors fused into the 3x3 conv's epilogue (conv.hip conv_epilogue_rpn)
    const float* w_split;         // [16][256] objectness + anchor-delta rows (+ a zero row) in the split row format
    const float* bias;            // [16]
    float* pred;                  // [M][16]
};
// rpn_sparse.hip: the RPN head's backward pass over the sampled anchors' pixels only (every other entry of the predictor gradients is an exact zero)
struct RpnSparseArgs {
    int B, batch, ld, K, C;                    // images, RPN.BATCH_SIZE_PER_IMAGE, predictor row length (16) and real rows, hidden channels (256)
    int fh[5], fw[5];
    const int* sampled;                        // [B][batch] sampled anchor indices (amp_rpn_sample_loss)
    const int* counts;                         // [B][2] positives, negatives
    const float* dpred[5];                     // d loss / d predictions [B * h * w][ld], zero except at the sampled anchors
    const float* t[5];                         // hidden activations [B * h * w][C]
    const float* feat[5];                      // FPN features [B * h * w][C]
    int t_split, feat_split;                   // ... in the split row format
    int recompute_t;                           // 1: t was not saved (the forward pass ran the fused head): the rows' hidden activations = relu(patch . W_conv + shift), recomputed
    const float* w_conv_split;                 // the conv's weights in the split operand layout (or null: split per call)
    const float* conv_shift;                   // the conv's bias (FrozenBN shift) or null
    const float* w_pred;                       // [K][C]
    const float* w_conv;                       // [C][3][3][C]
    const float* conv_scale;                   // FrozenBN scale of the conv or null
    float* gw_pred; float* gb_pred; float* gw_conv; float* gb_conv;      // gradients (written, not accumulated)
    float* dfeat[5];                           // gradient maps of the FPN features (accumulated in place)
    unsigned int* rows; int* nrows;            // workspace: [B * batch], [B]
    float* dpred_rows; float* act_rows; float* dt_rows;                  // [B * batch][16], [..][C], [..][C]
    float* xg; float* G; float* wt;            // [B * batch][9 C], [B * batch][9 C], [9 C][C]
    float* wg_scratch; size_t wg_scratch_floats;
};
int rpn_sparse_backward(amp_ctx* ctx, const RpnSparseArgs& a);
int conv_run(amp_ctx* ctx, const amp_conv_desc* d, int groups, const float* x, const float* w, const float* w_split, int force_f32,
             const float* scale, const float* shift, const float* res, const float* mask, float* y, int in_shift = 0, int fmt = 0,
             const PredictFuse* fuse = nullptr, const RpnFuse* rpn = nullptr);
// comm.hip (all no-ops / errors are explicit when the context has no communicator)
// grouped in-place SUM, after the compute stream's work so far; slot in [0, AMP_GRAD_BUCKETS): timed for amp_comm_bucket_stats
int comm_allreduce_ranges(amp_ctx* ctx, float* base, const size_t* off, const size_t* n, int nr, int slot = -1);
int comm_mark_producer_end(amp_ctx* ctx);      // the backward pass is complete on the compute stream (exposed-time reference)
int comm_wait_done(amp_ctx* ctx);              // compute stream waits (device side) for every collective issued so far
int comm_agree_flag(amp_ctx* ctx, int* d_flag); // MAX of a device int over the ranks, complete on return
int rle_strings_run(amp_ctx* ctx, const unsigned int* pool, const unsigned long long* off, const int* len, int n, char* str,
                    unsigned long long cap, unsigned long long* str_off, int* str_len, unsigned long long* total);
int roi_align_run(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count, int R, int P,
                  float* out, int* level_out, int out_split, int in_split = 0);   // out_split / in_split = 1: pooled tensor / feature maps in the split row format
int box_candidates_run(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B, int Rcap, int K,
                       const float reg_weights[4], float score_thresh, int img_h, int img_w, const int* img_hw, const float* thresh_img,
                       float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow);
// conv.hip: the f16x3 split copies of many weight tensors in one launch (weight_jobs_kernel).  chunks_dev: uint2 {job, first pair}, 8192 pairs each
struct WeightJob {
    const float* w;               // [N][KH][KW][C] fp32
    const float* scale;           // transpose = 1: per-n factor (folded FrozenBN) or null
    unsigned int* out;            // split rows
    int N, KH, KW, C;
    int transpose;                // 1: the data-gradient form [C][KH flipped][KW flipped][N], N % 32 == 0; 0: rows of w, (KH*KW*C) % 32 == 0;
                                  // 2: as 1 through 64 x 64 LDS tiles (N % 64 == 0, C % 64 == 0; chunk = {job, tile})
};
int weight_jobs_run(amp_ctx* ctx, const WeightJob* jobs_dev, const void* chunks_dev, int nchunks);
// train_bwd.hip: amp_sgd_update over a device table of chunks (low 32 bits: offset in floats, a multiple of 4; high 32: length) of the arenas p / g / v
int sgd_chunks_run(amp_ctx* ctx, const unsigned long long* chunks_dev, int nchunks, float* p, const float* g, float* v, float lr,
                   float momentum, float weight_decay, float grad_scale);
// wgrad.hip: slab reductions on a second stream.  begin(scratch1): every amp_conv2d_wgrad* call from here on launches its reduction on
// ctx->side (alternating between the scratch it is given and scratch1); join: `stream` waits for every reduction issued so far (before
// anything on it reads or adds to a gradient); end: join + off.
int wgrad_async_begin(amp_ctx* ctx, float* scratch1);
int wgrad_async_join(amp_ctx* ctx);
int wgrad_async_end(amp_ctx* ctx);
int roi_align_bwd_run(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                      const int* batch_idx, int R, int P, const float* dout, int B, int init);   // train_bwd.hip: amp_roi_align_bwd_batched + first-writer mode
int upsample2_bwd_run(amp_ctx* ctx, const float* dfine, float* dcoarse, int B, int Hc, int Wc, int C, int init);   // train_bwd.hip
int maxpool_run(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y, int y_split);
int compact_dets_run(amp_ctx* ctx, int B, int D, const int* det_count, const float* det_boxes, const float* det_scores, const int* det_classes,
                     float* boxes, float* scores, int* classes, int* batch, int* n_total /* device, optional: length of the compact list */);   // box_infer.hip
// conv.hip: stem conv + ReLU + max-pool fused (AMP_CONV_F16X3, pre-split weights); returns 1 when it does not apply (caller: conv, then maxpool_run)
// x_split: the input pixels are in preprocess_run's split form (16 B = 4 hi halves | 4 lo' halves)
int stem_pool_run(amp_ctx* ctx, int B, int H, int W, const float* x, int x_split, const float* w_split, const float* scale, const float* shift,
                  float* pool, int pool_split);
// conv.hip: conv2 + conv3 of a res2 bottleneck in one launch (conv3x3_c64_kernel<false, true>); 1 = does not apply (caller: two convolutions)
int conv_c64_fused3_run(amp_ctx* ctx, int B, int H, int W, const float* x_split, const float* w2_split, const float* scale2, const float* shift2,
                        const float* w3_split, const float* scale3, const float* shift3, int C3, const float* res_split, float* y_split);
bool stem_pool_applies(amp_ctx* ctx, const float* w_split);     // would stem_pool_run launch the fused kernel (mode, switches)?
bool stem_u8_applies(amp_ctx* ctx, const float* w_split);       // ... and straight from the uint8 image (stem_pool_u8_kernel: no preprocess pass)?
int stem_pool_u8_run(amp_ctx* ctx, const uint8_t* img, int B, int H, int W, int Hp, int Wp, const float mean[3], const float std_[3],
                     const int* img_hw, const float* w_split, const float* scale, const float* shift, float* pool, int pool_split);
int preprocess_run(amp_ctx* ctx, const uint8_t* img_bgr, int B, int H, int W, int Hp, int Wp, const float mean[3], const float std[3],
                   const int* img_hw, float* out, int out_split);   // pointwise.hip   // pointwise.hip: y_split = 1 writes split rows
// fmt bit 0: x is in the split hi|lo' row format (written by a producer with bit 1); bit 1: write y in that format (AMP_CONV_F16X3
// only; a [rows][C] fp32 tensor and its split form have the same byte size and row offsets)
}
