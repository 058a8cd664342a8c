// Training-mode forward: label assignment, seeded sub-sampling and the five Mask R-CNN losses (SURVEY.md §8a rows a2, a18;
// detectron2 rpn.py label_and_sample_anchors / losses, matcher.py, sampling.py, roi_heads.py label_and_sample_proposals,
// fast_rcnn.py losses, mask_head.py mask_rcnn_loss, masks.py rasterize_polygons_within_box + pycocotools rleFrPoly) --
// what `self._model(data)` returns to AMPIS's LossEvalHook (ampis/data_utils.py:111-122).
//
// Every loss kernel also writes the gradient of its loss w.r.t. the network outputs it reads (sparse for the RPN), so the
// backward pass starts from these buffers.  Sub-sampling is a counter-based hash (oracle/train.py hash32): take the k
// candidates with the largest key = max(0xffffffff - hash32(seed, image, stream, index), 1), ties by index.
// Reductions are fixed-order trees inside one workgroup per image / RoI; the host adds the per-image partials in order.
#include "common.h"
#include "select.h"

namespace {

constexpr int NL = 5;
constexpr int MAXG_TILE = 256;

__device__ __forceinline__ uint32_t hash32(uint32_t seed, uint32_t image, uint32_t stream, uint32_t idx) {
    uint32_t x = seed * 0x9E3779B1u + image * 0x85EBCA77u + stream * 0xC2B2AE3Du + idx * 0x27D4EB2Fu;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t sample_key(uint32_t seed, uint32_t image, uint32_t stream, uint32_t idx) {
    const uint32_t k = 0xffffffffu - hash32(seed, image, stream, idx);
    return k ? k : 1u;
}

struct AnchorGeom {
    int hw[NL], fw[NL], stride[NL], off[NL + 1];   // off: first anchor index of each level (x3 anchors per location)
    float cell[NL][3][4];
    int total;
};

__device__ __forceinline__ void anchor_box(const AnchorGeom& g, int a, float& x1, float& y1, float& x2, float& y2, int& lvl, int& local) {
    lvl = 0;
    while (lvl + 1 < NL && a >= g.off[lvl + 1]) ++lvl;
    local = a - g.off[lvl];
    const int pix = local / 3, an = local - pix * 3;
    const int py = pix / g.fw[lvl], px = pix - py * g.fw[lvl];
    const float sx = (float)(px * g.stride[lvl]), sy = (float)(py * g.stride[lvl]);
    x1 = __fadd_rn(sx, g.cell[lvl][an][0]); y1 = __fadd_rn(sy, g.cell[lvl][an][1]);
    x2 = __fadd_rn(sx, g.cell[lvl][an][2]); y2 = __fadd_rn(sy, g.cell[lvl][an][3]);
}

// detectron2 pairwise_iou(gt, box): 0 where the intersection is empty
__device__ __forceinline__ float iou_gt_box(float gx1, float gy1, float gx2, float gy2, float garea, float x1, float y1, float x2,
                                            float y2, float area) {
    const float w = fmaxf(__fsub_rn(fminf(gx2, x2), fmaxf(gx1, x1)), 0.f);
    const float h = fmaxf(__fsub_rn(fminf(gy2, y2), fmaxf(gy1, y1)), 0.f);
    const float inter = __fmul_rn(w, h);
    return inter > 0.f ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(garea, area), inter)) : 0.f;
}

// The same value for the anchor sweeps, where a wave is 64 neighbouring anchors and a GT box meets few of them: the division (most of
// the arithmetic) sits behind a wave-uniform test, so a wave pays it only for the GT boxes that touch one of its anchors.
// `any` = some lane of the wave has a non-empty intersection.
__device__ __forceinline__ float iou_gt_box_sweep(float gx1, float gy1, float gx2, float gy2, float garea, float x1, float y1, float x2,
                                                  float y2, float area, bool& any) {
    const float w = fmaxf(__fsub_rn(fminf(gx2, x2), fmaxf(gx1, x1)), 0.f);
    const float h = fmaxf(__fsub_rn(fminf(gy2, y2), fmaxf(gy1, y1)), 0.f);
    const float inter = __fmul_rn(w, h);
    float v = 0.f;
    any = __builtin_amdgcn_ballot_w64(inter > 0.f) != 0ull;
    if (any) v = inter > 0.f ? __fdiv_rn(inter, __fsub_rn(__fadd_rn(garea, area), inter)) : 0.f;
    return v;
}

// ---- anchors <-> GT: best GT per anchor and best IoU per GT (Matcher inputs) -------------------------------------------------
// Both sweeps walk the GT boxes of an image per wave of 64 neighbouring anchors.  A wave first takes the bounding box of its anchors;
// of every 64 GT boxes (one per lane) only those that overlap it are evaluated -- a GT box outside the bounding box has IoU 0 with
// each of the wave's anchors, which changes neither a maximum that starts at 0 nor an equality test against a positive best IoU.
struct WaveBox { float x1, y1, x2, y2; };
__device__ __forceinline__ WaveBox wave_bounds(float x1, float y1, float x2, float y2) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x1 = fminf(x1, __shfl_xor(x1, o)); y1 = fminf(y1, __shfl_xor(y1, o));
        x2 = fmaxf(x2, __shfl_xor(x2, o)); y2 = fmaxf(y2, __shfl_xor(y2, o));
    }
    return WaveBox{x1, y1, x2, y2};
}

__global__ __launch_bounds__(256) void anchor_match_kernel(const AnchorGeom g, const float* __restrict__ gt_boxes,
                                                           const int* __restrict__ gt_off, float* match_val, int* match_idx,
                                                           unsigned int* gt_best /* [sum G] float bits, zeroed */) {
    __shared__ float sg[MAXG_TILE][5];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int g0 = gt_off[b], G = gt_off[b + 1] - g0;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    int lvl, local;
    const bool av = a < g.total;
    if (av) anchor_box(g, a, x1, y1, x2, y2, lvl, local);      // lanes beyond the last anchor hold an empty box at the origin: IoU 0
    const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    const WaveBox wb = wave_bounds(x1, y1, x2, y2);
    float best = 0.f;     // torch.max(dim=0) over IoUs >= 0: the first GT unless a later one is strictly larger
    int besti = 0;
    for (int t0 = 0; t0 < G; t0 += MAXG_TILE) {
        const int tn = min(MAXG_TILE, G - t0);
        __syncthreads();
        if ((int)threadIdx.x < tn) {
            const float* p = gt_boxes + (size_t)(g0 + t0 + threadIdx.x) * 4;
            sg[threadIdx.x][0] = p[0]; sg[threadIdx.x][1] = p[1]; sg[threadIdx.x][2] = p[2]; sg[threadIdx.x][3] = p[3];
            sg[threadIdx.x][4] = __fmul_rn(__fsub_rn(p[2], p[0]), __fsub_rn(p[3], p[1]));
        }
        __syncthreads();
        for (int c0 = 0; c0 < tn; c0 += 64) {
            const int jl = c0 + lane;
            const bool hit = jl < tn && sg[jl][0] < wb.x2 && sg[jl][2] > wb.x1 && sg[jl][1] < wb.y2 && sg[jl][3] > wb.y1;
            unsigned long long pend = __builtin_amdgcn_ballot_w64(hit);
            while (pend) {                                     // ascending GT index: the first maximum wins
                const int j = c0 + __builtin_ctzll(pend);
                pend &= pend - 1;
                bool any;
                const float v = iou_gt_box_sweep(sg[j][0], sg[j][1], sg[j][2], sg[j][3], sg[j][4], x1, y1, x2, y2, area, any);
                if (v > best) { best = v; besti = t0 + j; }
                // this GT's best IoU over all anchors: wave maximum -> one atomicMax per (wave, GT), and only for the few GTs the wave's
                // 64 neighbouring anchors touch at all (a max is order-independent: deterministic).  A workgroup-level reduction here
                // cost two barriers per GT and 2/3 of the kernel.
                if (any) {
                    float m = v;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                    if (lane == 0) atomicMax(&gt_best[g0 + t0 + j], __float_as_uint(m));   // IoU >= 0: uint order == float order
                }
            }
        }
    }
    if (av) {
        match_val[(size_t)b * g.total + a] = best;
        match_idx[(size_t)b * g.total + a] = besti;
    }
}

// Matcher labels (thresholds lo/hi -> 0 / -1 / 1) + set_low_quality_matches_ (IoU == that GT's best over all anchors)
__global__ __launch_bounds__(256) void anchor_label_kernel(const AnchorGeom g, const float* __restrict__ gt_boxes,
                                                           const int* __restrict__ gt_off, const float* match_val,
                                                           const unsigned int* gt_best, float lo, float hi, signed char* label) {
    __shared__ float sg[MAXG_TILE][6];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int g0 = gt_off[b], G = gt_off[b + 1] - g0;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    int lvl, local;
    const bool av = a < g.total;
    if (av) anchor_box(g, a, x1, y1, x2, y2, lvl, local);
    const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    const WaveBox wb = wave_bounds(x1, y1, x2, y2);
    bool lowq = false;
    for (int t0 = 0; t0 < G; t0 += MAXG_TILE) {
        const int tn = min(MAXG_TILE, G - t0);
        __syncthreads();
        bool zero_best = false;
        if ((int)threadIdx.x < tn) {
            const float* p = gt_boxes + (size_t)(g0 + t0 + threadIdx.x) * 4;
            sg[threadIdx.x][0] = p[0]; sg[threadIdx.x][1] = p[1]; sg[threadIdx.x][2] = p[2]; sg[threadIdx.x][3] = p[3];
            sg[threadIdx.x][4] = __fmul_rn(__fsub_rn(p[2], p[0]), __fsub_rn(p[3], p[1]));
            sg[threadIdx.x][5] = __uint_as_float(gt_best[g0 + t0 + threadIdx.x]);
            zero_best = sg[threadIdx.x][5] == 0.f;
        }
        // a GT box that no anchor touches has best IoU 0 and "equals" the IoU 0 of every anchor (the reference marks them all)
        if (__syncthreads_or(zero_best ? 1 : 0)) lowq = true;
        for (int c0 = 0; c0 < tn; c0 += 64) {
            const int jl = c0 + lane;
            const bool hit = jl < tn && sg[jl][0] < wb.x2 && sg[jl][2] > wb.x1 && sg[jl][1] < wb.y2 && sg[jl][3] > wb.y1;
            unsigned long long pend = __builtin_amdgcn_ballot_w64(hit);
            while (pend) {
                const int j = c0 + __builtin_ctzll(pend);
                pend &= pend - 1;
                bool any;
                const float v = iou_gt_box_sweep(sg[j][0], sg[j][1], sg[j][2], sg[j][3], sg[j][4], x1, y1, x2, y2, area, any);
                lowq = lowq || (v == sg[j][5]);
            }
        }
    }
    if (av) {
        signed char l = 0;
        if (G > 0) {
            const float v = match_val[(size_t)b * g.total + a];
            l = (v >= hi) ? 1 : (v >= lo ? -1 : 0);
            if (lowq) l = 1;
        }
        label[(size_t)b * g.total + a] = l;
    }
}

// ---- RPN: sample 256 anchors / image, BCE + L1 losses and their gradients -----------------------------------------------------
struct RpnLossArgs {
    AnchorGeom g;
    const float* pred[NL];        // [B, hw, ld]: 3 logits, 12 deltas (+ padding)
    int ld;
    float* dpred[NL];             // same shape, zero-filled by the caller; receives d(loss)/d(pred) (may be null)
    const float* gt_boxes;
    const int* gt_off;
    const signed char* label;
    const int* match_idx;
    uint32_t* keys_scratch;       // [B][total]
    int batch, num_pos_max;       // 256, 128
    uint32_t seed;
    float inv_norm;               // 1 / (batch * B)
    int* sampled;                 // [B][batch] anchor indices: positives then negatives, ascending hash
    int* counts;                  // [B][2] npos, nneg
    float* partial;               // [B][2] sum BCE, sum L1 (unnormalised)
    // chunked selection (rpn_sample_select_kernel): the anchors of an image are cut into nch chunks of <= SAMPLE_CHUNK, every (image,
    // positives | negatives, chunk) keeps its best `batch` words; rpn_sample_loss_kernel then orders the <= nch * batch words of a class
    unsigned long long* cand;     // [B][2][nch][batch] sorted words (key << 32 | ~anchor), null: one workgroup per image selects from all anchors
    int* cand_count;              // [B][2][nch]
    int nch;
};
constexpr int SAMPLE_CHUNK = 49152;      // = 48 keys per thread in registers (select_topk_reg), as the RPN top-k

// One workgroup per (image, class, chunk).  The same keys and the same order as the one-workgroup selection (key descending, anchor ascending):
// the best `batch` of a class are among the best `batch` of its chunks.
__global__ __launch_bounds__(1024) void rpn_sample_select_kernel(const RpnLossArgs a) {
    __shared__ amp::SelectSmem sm;
    const int c = blockIdx.x % a.nch, which = (blockIdx.x / a.nch) & 1, b = blockIdx.x / (2 * a.nch);
    const int n = a.g.total;
    const int i0 = c * SAMPLE_CHUNK, nc = min(SAMPLE_CHUNK, n - i0);
    const signed char* lab = a.label + (size_t)b * n + i0;
    const uint32_t seed = a.seed;
    const signed char want = which == 0 ? 1 : 0;
    const int kmax = which == 0 ? a.num_pos_max : a.batch;
    const int k = amp::select_topk_reg<SAMPLE_CHUNK / 1024>(sm, nc, kmax, [&](int i) { return lab[i] == want ? sample_key(seed, b, which, i0 + i) : 0u; });
    unsigned long long* out = a.cand + (size_t)blockIdx.x * a.batch;
    for (int i = threadIdx.x; i < k; i += 1024) out[i] = sm.sorted[i] - (unsigned long long)(uint32_t)i0;      // chunk-local index -> anchor index (~(i0 + i) = ~i - i0)
    if (threadIdx.x == 0) a.cand_count[blockIdx.x] = k;
}

__global__ __launch_bounds__(1024) void rpn_sample_loss_kernel(const RpnLossArgs a) {
    __shared__ amp::SelectSmem sm;
    __shared__ int s_idx[512];
    __shared__ float s_red[2][1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.g.total;
    const signed char* lab = a.label + (size_t)b * n;
    uint32_t* keys = a.keys_scratch + (size_t)b * n;
    const uint32_t seed = a.seed;
    int npos, nneg;
    if (a.cand) {
        // the chunks' candidates of a class: at most nch * batch <= 2048 sorted words -> one LDS sort, the first k are the selection
        auto from_chunks = [&](int which, int kmax) -> int {
            __syncthreads();
            const int base = (b * 2 + which) * a.nch;
            int total = 0;
            for (int c = 0; c < a.nch; ++c) total += a.cand_count[base + c];
            int N = 64;
            while (N < a.nch * a.batch) N <<= 1;
            for (int i = tid; i < N; i += 1024) {
                const int c = i / a.batch, j = i - c * a.batch;
                sm.sorted[i] = (c < a.nch && j < a.cand_count[base + c]) ? a.cand[(size_t)(base + c) * a.batch + j] : 0ull;
            }
            amp::bitonic_desc<1024>(sm.sorted, N);
            return min(kmax, total);
        };
        npos = from_chunks(0, a.num_pos_max);
        for (int i = tid; i < npos; i += 1024) s_idx[i] = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
        nneg = from_chunks(1, a.batch - npos);
        for (int i = tid; i < nneg; i += 1024) s_idx[npos + i] = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
        __syncthreads();
    } else {
    npos = amp::select_topk(sm, n, a.num_pos_max, keys, [&](int i) { return lab[i] == 1 ? sample_key(seed, b, 0, i) : 0u; });
    for (int i = tid; i < npos; i += 1024) s_idx[i] = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
    __syncthreads();
    nneg = amp::select_topk(sm, n, a.batch - npos, keys, [&](int i) { return lab[i] == 0 ? sample_key(seed, b, 1, i) : 0u; });
    for (int i = tid; i < nneg; i += 1024) s_idx[npos + i] = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
    __syncthreads();
    }
    float bce = 0.f, l1 = 0.f;
    if (tid < npos + nneg) {
        const int an = s_idx[tid];
        a.sampled[(size_t)b * a.batch + tid] = an;
        float x1, y1, x2, y2;
        int lvl, local;
        anchor_box(a.g, an, x1, y1, x2, y2, lvl, local);
        const int pix = local / 3, k = local - pix * 3;
        const size_t row = ((size_t)b * a.g.hw[lvl] + pix) * a.ld;
        const float* p = a.pred[lvl] + row;
        float* dp = a.dpred[lvl] ? a.dpred[lvl] + row : nullptr;
        const float x = p[k];
        const float y = tid < npos ? 1.f : 0.f;
        // binary_cross_entropy_with_logits: max(x,0) - x*y + log1p(exp(-|x|)); d/dx = sigmoid(x) - y
        bce = __fadd_rn(__fsub_rn(fmaxf(x, 0.f), __fmul_rn(x, y)), log1pf(expf(-fabsf(x))));
        if (dp) dp[k] = __fmul_rn(__fsub_rn(__fdiv_rn(1.f, __fadd_rn(1.f, expf(-x))), y), a.inv_norm);
        if (tid < npos) {
            const float* gb = a.gt_boxes + (size_t)(a.gt_off[b] + a.match_idx[(size_t)b * n + an]) * 4;
            const float sw = __fsub_rn(x2, x1), sh = __fsub_rn(y2, y1);
            const float sx = __fadd_rn(x1, __fmul_rn(0.5f, sw)), sy = __fadd_rn(y1, __fmul_rn(0.5f, sh));
            const float tw = __fsub_rn(gb[2], gb[0]), th = __fsub_rn(gb[3], gb[1]);
            const float tx = __fadd_rn(gb[0], __fmul_rn(0.5f, tw)), ty = __fadd_rn(gb[1], __fmul_rn(0.5f, th));
            const float t[4] = {__fdiv_rn(__fsub_rn(tx, sx), sw), __fdiv_rn(__fsub_rn(ty, sy), sh), logf(__fdiv_rn(tw, sw)),
                                logf(__fdiv_rn(th, sh))};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float d = __fsub_rn(p[3 + 4 * k + q], t[q]);
                l1 = __fadd_rn(l1, fabsf(d));
                if (dp) dp[3 + 4 * k + q] = (d > 0.f ? a.inv_norm : (d < 0.f ? -a.inv_norm : 0.f));
            }
        }
    }
    s_red[0][tid] = bce; s_red[1][tid] = l1;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { s_red[0][tid] = __fadd_rn(s_red[0][tid], s_red[0][tid + o]); s_red[1][tid] = __fadd_rn(s_red[1][tid], s_red[1][tid + o]); }
        __syncthreads();
    }
    if (tid == 0) {
        a.counts[2 * b] = npos; a.counts[2 * b + 1] = nneg;
        a.partial[2 * b] = s_red[0][0]; a.partial[2 * b + 1] = s_red[1][0];
    }
}

// ---- RoI heads: append GT, match (IoU >= 0.5), sample 512 / image with <= 25 % foreground -----------------------------------------
struct RoiSampleArgs {
    const float* prop_boxes;      // [B][Pcap][4]
    const int* prop_count;        // [B]
    const float* gt_boxes;
    const int* gt_classes;
    const int* gt_off;
    int Pcap, K, batch, num_fg_max;
    float iou_thresh;
    uint32_t seed;
    uint32_t* keys_scratch;       // [B][ncap]
    int* cls_scratch;             // [B][ncap]
    int* gti_scratch;             // [B][ncap]
    int ncap;
    float* rois;                  // [B][batch][4]
    int* roi_cls;                 // [B][batch]  (K = background, -1 = unused slot)
    int* roi_gti;                 // [B][batch]  matched GT (index within the image)
    int* counts;                  // [B][2] nfg, nbg
    const int* prop_anchor;       // [B][Pcap] originating anchor of each proposal: the hash identity (order-independent)
    int num_anchors;              // GT box j hashes as num_anchors + j
};

__global__ __launch_bounds__(1024) void roi_sample_kernel(const RoiSampleArgs a) {
    __shared__ amp::SelectSmem sm;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int P = min(a.prop_count[b], a.Pcap);
    const int g0 = a.gt_off[b], G = a.gt_off[b + 1] - g0;
    const int n = P + G;
    int* cls = a.cls_scratch + (size_t)b * a.ncap;
    int* gti = a.gti_scratch + (size_t)b * a.ncap;
    uint32_t* keys = a.keys_scratch + (size_t)b * a.ncap;
    auto box_of = [&](int i) -> const float* { return i < P ? a.prop_boxes + ((size_t)b * a.Pcap + i) * 4 : a.gt_boxes + (size_t)(g0 + i - P) * 4; };
    for (int i = tid; i < n; i += 1024) {
        const float* p = box_of(i);
        const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
        const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
        float best = -1.f;
        int bi = 0;
        for (int j = 0; j < G; ++j) {
            const float* q = a.gt_boxes + (size_t)(g0 + j) * 4;
            const float ga = __fmul_rn(__fsub_rn(q[2], q[0]), __fsub_rn(q[3], q[1]));
            const float v = iou_gt_box(q[0], q[1], q[2], q[3], ga, x1, y1, x2, y2, area);
            if (v > best) { best = v; bi = j; }
        }
        const bool fg = G > 0 && best >= a.iou_thresh;
        cls[i] = fg ? a.gt_classes[g0 + bi] : a.K;
        gti[i] = bi;
    }
    __syncthreads();
    const uint32_t seed = a.seed;
    const int K = a.K;
    auto ident = [&](int i) -> uint32_t { return (uint32_t)(i < P ? a.prop_anchor[(size_t)b * a.Pcap + i] : a.num_anchors + (i - P)); };
    const int nfg = amp::select_topk(sm, n, a.num_fg_max, keys, [&](int i) { return cls[i] != K ? sample_key(seed, b, 2, ident(i)) : 0u; });
    for (int i = tid; i < a.batch; i += 1024) a.roi_cls[(size_t)b * a.batch + i] = -1;
    __syncthreads();
    for (int i = tid; i < nfg; i += 1024) {
        const int src = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
        const float* p = box_of(src);
        float* o = a.rois + ((size_t)b * a.batch + i) * 4;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
        a.roi_cls[(size_t)b * a.batch + i] = cls[src];
        a.roi_gti[(size_t)b * a.batch + i] = gti[src];
    }
    const int nbg = amp::select_topk(sm, n, a.batch - nfg, keys, [&](int i) { return cls[i] == K ? sample_key(seed, b, 3, ident(i)) : 0u; });
    for (int i = tid; i < nbg; i += 1024) {
        const int src = (int)(0xffffffffu - (uint32_t)(sm.sorted[i] & 0xffffffffu));
        const float* p = box_of(src);
        float* o = a.rois + ((size_t)b * a.batch + nfg + i) * 4;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3];
        a.roi_cls[(size_t)b * a.batch + nfg + i] = K;
        a.roi_gti[(size_t)b * a.batch + nfg + i] = gti[src];
    }
    for (int i = tid + nfg + nbg; i < a.batch; i += 1024) {
        float* o = a.rois + ((size_t)b * a.batch + i) * 4;
        o[0] = o[1] = o[2] = o[3] = 0.f;
    }
    if (tid == 0) { a.counts[2 * b] = nfg; a.counts[2 * b + 1] = nbg; }
}

// ---- box head losses: softmax CE (mean over all sampled RoIs) + L1 on the GT-class deltas of the foreground RoIs ---------------------
struct BoxLossArgs {
    const float* pred;            // [B*batch][ld]: K+1 logits then 4K deltas
    float* dpred;                 // same shape (may be null)
    const float* rois;
    const int* roi_cls;
    const int* roi_gti;
    const float* gt_boxes;
    const int* gt_off;
    int batch, K, ld;
    float wx, wy, ww, wh;
    float inv_total;              // 1 / (number of sampled RoIs in the batch)
    float* partial;               // [B][2] sum CE, sum L1
};

__global__ __launch_bounds__(512) void box_loss_kernel(const BoxLossArgs a) {
    __shared__ float s_red[2][512];
    const int b = blockIdx.x, tid = threadIdx.x;
    float ce = 0.f, l1 = 0.f;
    for (int r = tid; r < a.batch; r += 512) {
        const size_t row = (size_t)b * a.batch + r;
        const int c = a.roi_cls[row];
        float* dp = a.dpred ? a.dpred + row * a.ld : nullptr;
        if (c < 0) {
            if (dp) for (int k = 0; k < a.ld; ++k) dp[k] = 0.f;
            continue;
        }
        const float* p = a.pred + row * a.ld;
        float mx = p[0];
        for (int k = 1; k <= a.K; ++k) mx = fmaxf(mx, p[k]);
        float sum = 0.f;
        for (int k = 0; k <= a.K; ++k) sum = __fadd_rn(sum, expf(__fsub_rn(p[k], mx)));
        const float lse = __fadd_rn(mx, logf(sum));
        ce = __fadd_rn(ce, __fsub_rn(lse, p[c]));
        if (dp) {
            for (int k = 0; k <= a.K; ++k)
                dp[k] = __fmul_rn(__fsub_rn(__fdiv_rn(expf(__fsub_rn(p[k], mx)), sum), k == c ? 1.f : 0.f), a.inv_total);
            for (int k = a.K + 1; k < a.ld; ++k) dp[k] = 0.f;
        }
        if (c < a.K) {
            const float* s = a.rois + row * 4;
            const float* g = a.gt_boxes + (size_t)(a.gt_off[b] + a.roi_gti[row]) * 4;
            const float sw = __fsub_rn(s[2], s[0]), sh = __fsub_rn(s[3], s[1]);
            const float sx = __fadd_rn(s[0], __fmul_rn(0.5f, sw)), sy = __fadd_rn(s[1], __fmul_rn(0.5f, sh));
            const float tw = __fsub_rn(g[2], g[0]), th = __fsub_rn(g[3], g[1]);
            const float tx = __fadd_rn(g[0], __fmul_rn(0.5f, tw)), ty = __fadd_rn(g[1], __fmul_rn(0.5f, th));
            const float t[4] = {__fdiv_rn(__fmul_rn(a.wx, __fsub_rn(tx, sx)), sw), __fdiv_rn(__fmul_rn(a.wy, __fsub_rn(ty, sy)), sh),
                                __fmul_rn(a.ww, logf(__fdiv_rn(tw, sw))), __fmul_rn(a.wh, logf(__fdiv_rn(th, sh)))};
            for (int q = 0; q < 4; ++q) {
                const int col = a.K + 1 + 4 * c + q;
                const float d = __fsub_rn(p[col], t[q]);
                l1 = __fadd_rn(l1, fabsf(d));
                if (dp) dp[col] = (d > 0.f ? a.inv_total : (d < 0.f ? -a.inv_total : 0.f));
            }
        }
    }
    s_red[0][tid] = ce; s_red[1][tid] = l1;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
        if (tid < o) { s_red[0][tid] = __fadd_rn(s_red[0][tid], s_red[0][tid + o]); s_red[1][tid] = __fadd_rn(s_red[1][tid], s_red[1][tid + o]); }
        __syncthreads();
    }
    if (tid == 0) { a.partial[2 * b] = s_red[0][0]; a.partial[2 * b + 1] = s_red[1][0]; }
}

// ---- mask targets: rasterize_polygons_within_box (28x28) by pycocotools' rleFrPoly; mask BCE loss + gradient ---------------------------
struct MaskLossArgs {
    const float* logits;          // [N][28][28][K]
    float* dlogits;               // same shape (may be null)
    const float* rois;            // [N][4]
    const int* cls;               // [N]
    const int* poly_id;           // [N] global instance index (the matched GT)
    const double* poly_xy;        // flat vertex list
    const int* poly_off;          // [polygons+1] offsets into poly_xy (in doubles)
    const int* inst_poly_off;     // [instances+1] polygons of an instance (null: exactly one polygon per instance, polygon index = instance index)
    const unsigned long long* rle_off;   // [instances+1] (null: no bitmask ground truth): an instance with runs takes its target from target_in
    const unsigned char* target_in;      // [N][784] row-major targets of the bitmask instances (mask_target_bitmask_kernel)
    int N, K;
    float inv_count;              // 1 / (N * 784)
    float* partial;               // [N] per-RoI BCE sums
    unsigned char* target_out;    // [N][784] (may be null): the rasterised targets, row-major, for the parity tests
};

constexpr int MS = 28;
constexpr int NBITS = MS * MS + 1;          // positions 0..784 (a = x*h + y with y <= h)
constexpr int NWORDS = (NBITS + 31) / 32;   // 25

__global__ __launch_bounds__(64) void mask_target_loss_kernel(const MaskLossArgs a) {
    __shared__ unsigned int tog[NWORDS];
    __shared__ unsigned char bits[MS * MS];   // column-major: bits[x*28 + y]
    __shared__ float s_red[64];
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* r = a.rois + (size_t)n * 4;
    const int pid = a.poly_id[n];
    const bool from_bitmask = a.rle_off && a.rle_off[pid + 1] > a.rle_off[pid];
    if (from_bitmask) {      // BitMasks.crop_and_resize ran in mask_target_bitmask_kernel: row-major bytes -> this kernel's column-major bits
        for (int i = lane; i < MS * MS; i += 64) { const int y = i / MS, x = i - y * MS; bits[x * MS + y] = a.target_in[(size_t)n * MS * MS + i]; }
    } else {
        for (int i = lane; i < MS * MS; i += 64) bits[i] = 0;
    }
    // detectron2 rasterize_polygons_within_box: box arithmetic in fp32, polygon in fp64
    const float w32 = __fsub_rn(r[2], r[0]), h32 = __fsub_rn(r[3], r[1]);
    const double rw = (double)__fdiv_rn((float)MS, fmaxf(w32, 0.1f)), rh = (double)__fdiv_rn((float)MS, fmaxf(h32, 0.1f));
    const double bx = (double)r[0], by = (double)r[1];
    // polygons_to_bitmask: frPyObjects of every polygon of the instance, merged (union), decoded
    const int q0 = from_bitmask ? 0 : (a.inst_poly_off ? a.inst_poly_off[pid] : pid);
    const int q1 = from_bitmask ? 0 : (a.inst_poly_off ? a.inst_poly_off[pid + 1] : pid + 1);
    for (int q = q0; q < q1; ++q) {
    __syncthreads();
    for (int i = lane; i < NWORDS; i += 64) tog[i] = 0u;
    __syncthreads();
    const double* xy = a.poly_xy + a.poly_off[q];
    const int k = (a.poly_off[q + 1] - a.poly_off[q]) / 2;
    const double scale = 5.0;
    for (int j = lane; j < k; j += 64) {
        const int j2 = (j + 1 == k) ? 0 : j + 1;
        int xs = (int)(scale * ((xy[2 * j] - bx) * rw) + 0.5), ys = (int)(scale * ((xy[2 * j + 1] - by) * rh) + 0.5);
        int xe = (int)(scale * ((xy[2 * j2] - bx) * rw) + 0.5), ye = (int)(scale * ((xy[2 * j2 + 1] - by) * rh) + 0.5);
        const int dx = abs(xe - xs), dy = abs(ys - ye);
        const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
        if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
        const int len = dx >= dy ? dx : dy;
        const double s = dx >= dy ? (dx ? (double)(ye - ys) / dx : 0.0) : (double)(xe - xs) / dy;
        int pu = 0, pv = 0;
        for (int d = 0; d <= len; ++d) {
            const int t = flip ? len - d : d;
            int u, v;
            if (dx >= dy) { u = t + xs; v = (int)(ys + s * t + 0.5); } else { v = t + ys; u = (int)(xs + s * t + 0.5); }
            if (d > 0 && u != pu) {
                double xd = (double)(u < pu ? u : u - 1);
                xd = (xd + 0.5) / scale - 0.5;
                if (floor(xd) == xd && xd >= 0 && xd <= MS - 1) {
                    double yd = (double)(v < pv ? v : pv);
                    yd = (yd + 0.5) / scale - 0.5;
                    if (yd < 0) yd = 0; else if (yd > MS) yd = MS;
                    yd = ceil(yd);
                    const int pos = (int)xd * MS + (int)yd;      // run boundary; runs alternate -> parity toggles
                    atomicXor(&tog[pos >> 5], 1u << (pos & 31));
                }
            }
            pu = u; pv = v;
        }
    }
    __syncthreads();
    if (lane == 0) {   // prefix parity over the 785 positions (column-major); 25 words
        unsigned int carry = 0;
        for (int wv = 0; wv < NWORDS; ++wv) {
            unsigned int x = tog[wv];
            x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
            if (carry) x = ~x;
            carry = x >> 31;
            tog[wv] = x;
        }
    }
    __syncthreads();
    for (int i = lane; i < MS * MS; i += 64) bits[i] |= (unsigned char)((tog[i >> 5] >> (i & 31)) & 1u);
    }
    __syncthreads();
    const int c = a.cls[n];
    float acc = 0.f;
    for (int i = lane; i < MS * MS; i += 64) {
        const int y = i / MS, x = i - y * MS;           // row-major pixel of the logits
        const float t = (float)bits[x * MS + y];
        const size_t off = ((size_t)n * MS * MS + i) * a.K;
        const float z = a.logits[off + c];
        acc = __fadd_rn(acc, __fadd_rn(__fsub_rn(fmaxf(z, 0.f), __fmul_rn(z, t)), log1pf(expf(-fabsf(z)))));
        if (a.dlogits) {
            for (int q = 0; q < a.K; ++q) a.dlogits[off + q] = 0.f;
            a.dlogits[off + c] = __fmul_rn(__fsub_rn(__fdiv_rn(1.f, __fadd_rn(1.f, expf(-z))), t), a.inv_count);
        }
        if (a.target_out) a.target_out[(size_t)n * MS * MS + i] = (unsigned char)t;
    }
    s_red[lane] = acc;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (lane < o) s_red[lane] = __fadd_rn(s_red[lane], s_red[lane + o]);
        __syncthreads();
    }
    if (lane == 0) a.partial[n] = s_red[0];
}

// ---- mask targets from BITMASK ground truth: detectron2 BitMasks.crop_and_resize = torchvision roi_align(mask as a 1-channel fp32 map,
// spatial_scale 1, sampling_ratio 0, aligned) >= 0.5 -------------------------------------------------------------------------------
// The instance's full-image mask never exists densely: it arrives as COCO run lengths (column-major).  One workgroup per RoI
//   1. finds the window of pixels the RoI's samples can touch,
//   2. decodes the runs that cross it into a bit map of the window (column-major like the runs: a run is a few word-wide ORs) in a
//      scratch slot of its own,
//   3. evaluates the 28 x 28 bins with roi_align's arithmetic, operation for operation (roi_align_lanes_kernel above; taps are 0 / 1).
struct BitmaskTargetArgs {
    const float* rois;                    // [N][4]
    const int* inst;                      // [N] instance of each RoI
    const unsigned long long* rle_off;    // [instances+1] into rle_counts
    const unsigned int* rle_counts;       // run lengths, the first run counts zeros
    const int* rle_hw;                    // [instances][2] height, width of the instance's mask
    unsigned int* scratch;                // [gridDim.x][slot_words]
    size_t slot_words;
    int N;
    unsigned char* target;                // [N][784] row-major
    int* overflow;                        // set when a window does not fit its slot (host sizes slots for the whole frame: never)
};

__global__ __launch_bounds__(256) void mask_target_bitmask_kernel(const BitmaskTargetArgs a) {
    __shared__ unsigned long long s_scan[256];
    const int tid = threadIdx.x;
    unsigned int* bm = a.scratch + (size_t)blockIdx.x * a.slot_words;
    for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
        const int inst = a.inst[n];
        const unsigned long long r0 = a.rle_off[inst], r1 = a.rle_off[inst + 1];
        if (r1 <= r0) continue;                                  // polygon instance: mask_target_loss_kernel rasterises it (block-uniform)
        const int nr = (int)(r1 - r0);
        const unsigned int* cnt = a.rle_counts + r0;
        const int H = a.rle_hw[2 * inst], W = a.rle_hw[2 * inst + 1];
        const float x1 = a.rois[4 * n + 0], y1 = a.rois[4 * n + 1], x2 = a.rois[4 * n + 2], y2 = a.rois[4 * n + 3];
        const float sw = __fsub_rn(x1, 0.5f), sh = __fsub_rn(y1, 0.5f), ew = __fsub_rn(x2, 0.5f), eh = __fsub_rn(y2, 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)MS), bw = __fdiv_rn(rw, (float)MS);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)MS)), gw = (int)ceilf(__fdiv_rn(rw, (float)MS));
        const float count = (float)max(gh * gw, 1);
        // pixels a sample can read: columns floor(max(x, 0)) and +1 for x in [sw, ew] (a little slack for fp32 rounding of the positions)
        const int wx0 = max(0, (int)floorf(fmaxf(sw, 0.f)) - 1), wx1 = min(W - 1, (int)floorf(fmaxf(ew, 0.f)) + 2);
        const int wy0 = max(0, (int)floorf(fmaxf(sh, 0.f)) - 1), wy1 = min(H - 1, (int)floorf(fmaxf(eh, 0.f)) + 2);
        const int ncols = max(wx1 - wx0 + 1, 0), nrows = max(wy1 - wy0 + 1, 0);
        const int pitch = (nrows + 31) >> 5;                     // words per column
        const size_t nwords = (size_t)ncols * pitch;
        if (nwords > a.slot_words) { if (tid == 0) *a.overflow = 1; continue; }
        __syncthreads();                                         // the previous RoI of this block has finished reading the slot
        for (size_t i = tid; i < nwords; i += 256) bm[i] = 0u;
        // start position of every run: each thread owns a contiguous chunk of runs; block-wide exclusive scan of the chunk sums
        const int per = (nr + 255) / 256;
        const int j0 = min(tid * per, nr), j1 = min(j0 + per, nr);
        unsigned long long sum = 0;
        for (int j = j0; j < j1; ++j) sum += cnt[j];
        s_scan[tid] = sum;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned long long v = tid >= o ? s_scan[tid - o] : 0ull;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        unsigned long long pos = s_scan[tid] - sum;              // exclusive
        for (int j = j0; j < j1; ++j) {
            const unsigned long long s0 = pos, e0 = pos + cnt[j];
            pos = e0;
            if (!(j & 1) || e0 == s0) continue;                  // even runs are zeros
            const int c_first = (int)(s0 / (unsigned)H), c_last = (int)((e0 - 1) / (unsigned)H);
            for (int c = max(c_first, wx0); c <= min(c_last, wx1); ++c) {
                const unsigned long long cb = (unsigned long long)c * H;
                int ya = (int)((s0 > cb ? s0 : cb) - cb), yb = (int)((e0 < cb + H ? e0 : cb + H) - cb);     // [ya, yb) inside column c
                ya = max(ya, wy0) - wy0; yb = min(yb, wy1 + 1) - wy0;
                if (yb <= ya) continue;
                unsigned int* col = bm + (size_t)(c - wx0) * pitch;
                for (int wv = ya >> 5; wv <= (yb - 1) >> 5; ++wv) {
                    const int lo = max(ya - (wv << 5), 0), hi = min(yb - (wv << 5), 32);                     // bits [lo, hi) of word wv
                    const unsigned int mask = (hi == 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
                    atomicOr(&col[wv], mask);
                }
            }
        }
        __syncthreads();
        auto tap = [&](int y, int x) -> float {                  // mask value of pixel (y, x); outside the window the mask is not sampled
            const int yy = y - wy0, xx = x - wx0;
            if (yy < 0 || yy >= nrows || xx < 0 || xx >= ncols) return 0.f;
            return (float)((bm[(size_t)xx * pitch + (yy >> 5)] >> (yy & 31)) & 1u);
        };
        for (int i = tid; i < MS * MS; i += 256) {
            const int ph = i / MS, pw = i - ph * MS;
            float acc = 0.f;
            for (int iy = 0; iy < gh; ++iy) {
                float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bh), (float)gh));
                if (y < -1.0f || y > (float)H) continue;
                if (y <= 0.f) y = 0.f;
                int ylo = (int)y, yhi;
                if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
                const float ly = __fsub_rn(y, (float)ylo), hy = __fsub_rn(1.0f, ly);
                for (int ix = 0; ix < gw; ++ix) {
                    float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                    if (x < -1.0f || x > (float)W) continue;
                    if (x <= 0.f) x = 0.f;
                    int xlo = (int)x, xhi;
                    if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
                    const float lx = __fsub_rn(x, (float)xlo), hx = __fsub_rn(1.0f, lx);
                    const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                    const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, tap(ylo, xlo)), __fmul_rn(w2, tap(ylo, xhi))), __fmul_rn(w3, tap(yhi, xlo))),
                                              __fmul_rn(w4, tap(yhi, xhi)));
                    acc = __fadd_rn(acc, t);
                }
            }
            a.target[(size_t)n * MS * MS + i] = (unsigned char)(__fdiv_rn(acc, count) >= 0.5f);
        }
    }
}

// ---- sigmoid focal loss (fvcore sigmoid_focal_loss as detectron2's dense heads use it), forward + gradient in one pass -----------------
// Not on the Mask R-CNN path of the reference; built because the north star lists it among the loss kernels.
//   p = sigmoid(x); ce = max(x, 0) - x t + log1p(exp(-|x|)); p_t = p t + (1 - p)(1 - t); loss = alpha_t ce (1 - p_t)^gamma
//   alpha_t = alpha t + (1 - alpha)(1 - t) for alpha >= 0, else 1;  t = one_hot(label)[:K] (label K = background, < 0 = ignored row)
//   d loss / d x = alpha_t [ (p - t)(1 - p_t)^gamma - ce gamma (1 - p_t)^(gamma-1) p (1 - p)(2t - 1) ]
// Sums: each workgroup adds its 256 lanes in a fixed tree, amp_sigmoid_focal_loss adds the workgroups' partials in index order on the
// host side of the same call (a second tiny kernel): the loss is bitwise reproducible.
struct FocalArgs {
    const float* logits;          // [N][K]
    const int* labels;            // [N]
    float* dlogits;               // [N][K] or null
    float* partial;               // [gridDim.x]
    long long total;              // N * K
    int K;
    float alpha, gamma, scale;    // scale multiplies loss and gradient (1 / normaliser)
};

__global__ __launch_bounds__(256) void sigmoid_focal_loss_kernel(const FocalArgs a) {
    __shared__ float s_red[256];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long long)gridDim.x * 256) {
        const long long n = i / a.K;
        const int k = (int)(i - n * a.K);
        const int lab = a.labels[n];
        float g = 0.f;
        if (lab >= 0) {
            const float x = a.logits[i];
            const float t = (lab == k) ? 1.f : 0.f;
            const float p = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-x)));
            const float ce = __fadd_rn(__fsub_rn(fmaxf(x, 0.f), __fmul_rn(x, t)), log1pf(expf(-fabsf(x))));
            const float pt = __fadd_rn(__fmul_rn(p, t), __fmul_rn(__fsub_rn(1.f, p), __fsub_rn(1.f, t)));
            const float q = __fsub_rn(1.f, pt);                                   // 1 - p_t
            const float mod = a.gamma == 0.f ? 1.f : powf(q, a.gamma);
            const float at = a.alpha >= 0.f ? __fadd_rn(__fmul_rn(a.alpha, t), __fmul_rn(__fsub_rn(1.f, a.alpha), __fsub_rn(1.f, t))) : 1.f;
            acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(at, ce), mod));
            if (a.dlogits) {
                const float dpt = __fmul_rn(__fmul_rn(p, __fsub_rn(1.f, p)), __fsub_rn(__fmul_rn(2.f, t), 1.f));
                const float dmod = a.gamma == 0.f ? 0.f : __fmul_rn(__fmul_rn(-a.gamma, powf(q, a.gamma - 1.f)), dpt);
                g = __fmul_rn(__fmul_rn(at, __fadd_rn(__fmul_rn(__fsub_rn(p, t), mod), __fmul_rn(ce, dmod))), a.scale);
            }
        }
        if (a.dlogits) a.dlogits[i] = g;
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] = __fadd_rn(s_red[threadIdx.x], s_red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[blockIdx.x] = s_red[0];
}

__global__ void focal_finish_kernel(const float* partial, int n, float scale, float* loss) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = __fadd_rn(s, partial[i]);
    *loss = __fmul_rn(s, scale);
}

void fill_geom(AnchorGeom& g, const amp_rpn_levels* lv) {
    g.off[0] = 0;
    for (int l = 0; l < NL; ++l) {
        g.hw[l] = lv->h[l] * lv->w[l];
        g.fw[l] = lv->w[l];
        g.stride[l] = lv->stride[l];
        g.off[l + 1] = g.off[l] + g.hw[l] * 3;
        for (int r = 0; r < 3; ++r) {
            const double ratio = (r == 0) ? 0.5 : (r == 1 ? 1.0 : 2.0);
            const double area = (double)lv->anchor_size[l] * (double)lv->anchor_size[l];
            const double w = sqrt(area / ratio), h = ratio * w;
            g.cell[l][r][0] = (float)(-w / 2.0); g.cell[l][r][1] = (float)(-h / 2.0);
            g.cell[l][r][2] = (float)(w / 2.0);  g.cell[l][r][3] = (float)(h / 2.0);
        }
    }
    g.total = g.off[NL];
}

}  // namespace

extern "C" {

int amp_anchor_labels(amp_ctx* ctx, const amp_rpn_levels* lv, int B, const float* gt_boxes, const int* gt_off, int total_gt,
                      float iou_lo, float iou_hi, float* match_val, int* match_idx, unsigned int* gt_best, signed char* label) {
    AMP_REQUIRE(ctx && lv && gt_boxes && gt_off && match_val && match_idx && gt_best && label, "amp_anchor_labels: null argument");
    AMP_REQUIRE(lv->nlevels == NL && lv->A == 3, "amp_anchor_labels: need 5 levels, 3 anchors");
    AnchorGeom g;
    fill_geom(g, lv);
    AMP_HIP_CHECK(hipMemsetAsync(gt_best, 0, (size_t)(total_gt > 0 ? total_gt : 1) * sizeof(unsigned int), ctx->stream));
    const dim3 grid(amp::cdiv(g.total, 256), B);
    hipLaunchKernelGGL(anchor_match_kernel, grid, dim3(256), 0, ctx->stream, g, gt_boxes, gt_off, match_val, match_idx, gt_best);
    hipLaunchKernelGGL(anchor_label_kernel, grid, dim3(256), 0, ctx->stream, g, gt_boxes, gt_off, match_val, gt_best, iou_lo, iou_hi, label);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_rpn_sample_loss(amp_ctx* ctx, const amp_rpn_levels* lv, float* const dpred[5], int B, const float* gt_boxes,
                        const int* gt_off, const signed char* label, const int* match_idx, uint32_t* keys_scratch, int batch,
                        float pos_frac, unsigned int seed, int* sampled, int* counts, float* partial) {
    AMP_REQUIRE(ctx && lv && gt_boxes && gt_off && label && match_idx && keys_scratch && sampled && counts && partial, "amp_rpn_sample_loss: null argument");
    AMP_REQUIRE(batch >= 1 && batch <= 512, "amp_rpn_sample_loss: batch must be in [1,512]");
    RpnLossArgs a;
    fill_geom(a.g, lv);
    for (int l = 0; l < NL; ++l) { a.pred[l] = lv->pred[l]; a.dpred[l] = dpred ? dpred[l] : nullptr; }
    a.ld = lv->ld;
    a.gt_boxes = gt_boxes; a.gt_off = gt_off; a.label = label; a.match_idx = match_idx; a.keys_scratch = keys_scratch;
    a.batch = batch; a.num_pos_max = (int)(batch * pos_frac); a.seed = seed; a.inv_norm = 1.0f / (float)(batch * B);
    a.sampled = sampled; a.counts = counts; a.partial = partial;
    // chunked selection when the chunks' candidate lists fit the caller's scratch ([B][total] uint32) and one LDS sort (nch * batch <= 2048)
    a.cand = nullptr; a.cand_count = nullptr; a.nch = amp::cdiv(a.g.total, SAMPLE_CHUNK);
    static const bool one_wg = getenv("AMP_SAMPLE_ONE_WG") != nullptr;      // EXPERIMENT switch: the one-workgroup-per-image selection
    const size_t cand_words = (size_t)B * 2 * a.nch * batch;
    const size_t need_bytes = cand_words * 8 + (size_t)B * 2 * a.nch * 4;
    if (!one_wg && a.nch * batch <= amp::SELECT_MAX_K && need_bytes <= (size_t)B * a.g.total * 4 && batch <= amp::SELECT_MAX_K) {
        a.cand = reinterpret_cast<unsigned long long*>(keys_scratch);
        a.cand_count = reinterpret_cast<int*>(a.cand + cand_words);
        hipLaunchKernelGGL(rpn_sample_select_kernel, dim3(B * 2 * a.nch), dim3(1024), 0, ctx->stream, a);
    }
    hipLaunchKernelGGL(rpn_sample_loss_kernel, dim3(B), dim3(1024), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_roi_sample(amp_ctx* ctx, int B, const float* prop_boxes, const int* prop_count, int Pcap, const float* gt_boxes,
                   const int* gt_classes, const int* gt_off, int K, int batch, float fg_frac, float iou_thresh, unsigned int seed,
                   uint32_t* keys_scratch, int* cls_scratch, int* gti_scratch, int ncap, float* rois, int* roi_cls, int* roi_gti,
                   int* counts, const int* prop_anchor, int num_anchors) {
    AMP_REQUIRE(ctx && prop_boxes && prop_count && gt_boxes && gt_classes && gt_off && keys_scratch && cls_scratch && gti_scratch && rois &&
                roi_cls && roi_gti && counts && prop_anchor, "amp_roi_sample: null argument");
    AMP_REQUIRE(batch >= 1 && batch <= amp::SELECT_MAX_K, "amp_roi_sample: batch out of range");
    RoiSampleArgs a;
    a.prop_boxes = prop_boxes; a.prop_count = prop_count; a.gt_boxes = gt_boxes; a.gt_classes = gt_classes; a.gt_off = gt_off;
    a.Pcap = Pcap; a.K = K; a.batch = batch; a.num_fg_max = (int)(batch * fg_frac); a.iou_thresh = iou_thresh; a.seed = seed;
    a.keys_scratch = keys_scratch; a.cls_scratch = cls_scratch; a.gti_scratch = gti_scratch; a.ncap = ncap;
    a.rois = rois; a.roi_cls = roi_cls; a.roi_gti = roi_gti; a.counts = counts; a.prop_anchor = prop_anchor; a.num_anchors = num_anchors;
    hipLaunchKernelGGL(roi_sample_kernel, dim3(B), dim3(1024), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_box_loss(amp_ctx* ctx, int B, int batch, int K, const float* pred, int ld, float* dpred, const float* rois, const int* roi_cls,
                 const int* roi_gti, const float* gt_boxes, const int* gt_off, const float reg_weights[4], int total_rois, float* partial) {
    AMP_REQUIRE(ctx && pred && rois && roi_cls && roi_gti && gt_boxes && gt_off && reg_weights && partial, "amp_box_loss: null argument");
    BoxLossArgs a;
    a.pred = pred; a.dpred = dpred; a.rois = rois; a.roi_cls = roi_cls; a.roi_gti = roi_gti; a.gt_boxes = gt_boxes; a.gt_off = gt_off;
    a.batch = batch; a.K = K; a.ld = ld; a.wx = reg_weights[0]; a.wy = reg_weights[1]; a.ww = reg_weights[2]; a.wh = reg_weights[3];
    a.inv_total = 1.0f / (float)(total_rois > 0 ? total_rois : 1);
    a.partial = partial;
    hipLaunchKernelGGL(box_loss_kernel, dim3(B), dim3(512), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_sigmoid_focal_loss(amp_ctx* ctx, long long N, int K, const float* logits, const int* labels, float alpha, float gamma, float scale,
                           float* dlogits, float* partial, int partial_cap, float* loss) {
    AMP_REQUIRE(ctx && logits && labels && partial && loss && N >= 0 && K >= 1 && partial_cap >= 1 && gamma >= 0.f, "amp_sigmoid_focal_loss: bad argument");
    FocalArgs a;
    a.logits = logits; a.labels = labels; a.dlogits = dlogits; a.partial = partial; a.total = N * K; a.K = K; a.alpha = alpha; a.gamma = gamma; a.scale = scale;
    const int grid = (int)std::max<long long>(1, std::min<long long>({(long long)partial_cap, 2048ll, (a.total + 255) / 256}));
    hipLaunchKernelGGL(sigmoid_focal_loss_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(1), 0, ctx->stream, partial, grid, scale, loss);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_mask_targets_bitmask(amp_ctx* ctx, int N, const float* rois, const int* inst, const unsigned long long* rle_off,
                             const uint32_t* rle_counts, const int* rle_hw, unsigned int* scratch, size_t slot_words, int nslots,
                             unsigned char* target, int* overflow_flag) {
    AMP_REQUIRE(ctx && rois && inst && rle_off && rle_counts && rle_hw && scratch && target && overflow_flag && slot_words > 0 && nslots > 0,
                "amp_mask_targets_bitmask: bad argument");
    if (N == 0) return AMP_OK;
    BitmaskTargetArgs a;
    a.rois = rois; a.inst = inst; a.rle_off = rle_off; a.rle_counts = rle_counts; a.rle_hw = rle_hw; a.scratch = scratch;
    a.slot_words = slot_words; a.N = N; a.target = target; a.overflow = overflow_flag;
    hipLaunchKernelGGL(mask_target_bitmask_kernel, dim3(std::min(N, nslots)), dim3(256), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_mask_target_loss_fmt(amp_ctx* ctx, int N, int K, const float* logits, float* dlogits, const float* rois, const int* cls,
                             const int* inst, const double* poly_xy, const int* poly_off, const int* inst_poly_off,
                             const unsigned long long* rle_off, const unsigned char* target_in, float* partial, unsigned char* target_out) {
    AMP_REQUIRE(ctx && logits && rois && cls && inst && poly_xy && poly_off && partial, "amp_mask_target_loss: null argument");
    AMP_REQUIRE(!rle_off || target_in, "amp_mask_target_loss: bitmask instances need the targets of amp_mask_targets_bitmask");
    if (N == 0) return AMP_OK;
    MaskLossArgs a;
    a.logits = logits; a.dlogits = dlogits; a.rois = rois; a.cls = cls; a.poly_id = inst; a.poly_xy = poly_xy; a.poly_off = poly_off;
    a.inst_poly_off = inst_poly_off; a.rle_off = rle_off; a.target_in = target_in;
    a.N = N; a.K = K; a.inv_count = 1.0f / ((float)N * MS * MS); a.partial = partial; a.target_out = target_out;
    hipLaunchKernelGGL(mask_target_loss_kernel, dim3(N), dim3(64), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_mask_target_loss(amp_ctx* ctx, int N, int K, const float* logits, float* dlogits, const float* rois, const int* cls,
                         const int* poly_id, const double* poly_xy, const int* poly_off, float* partial, unsigned char* target_out) {
    return amp_mask_target_loss_fmt(ctx, N, K, logits, dlogits, rois, cls, poly_id, poly_xy, poly_off, nullptr, nullptr, nullptr, partial, target_out);
}

}  // extern "C"
