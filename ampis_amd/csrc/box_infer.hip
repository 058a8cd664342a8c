// Box-head inference (detectron2 fast_rcnn.py FastRCNNOutputLayers.predict_* +
// fast_rcnn_inference_single_image): softmax over K+1 logits, class-specific decode (weights 10,10,5,5), clip,
// score > SCORE_THRESH_TEST -> candidates; they are then ordered (amp_sort_gather), suppressed per class (amp_nms) and the
// first DETECTIONS_PER_IMAGE survivors gathered here.  Decode / softmax follow the oracle's fp32 op order.
#include "common.h"

namespace {

struct CandArgs {
    const float* pred;        // [B*Rcap][ld]: cols [0,K] class logits (background last), cols [K+1, K+1+4K) deltas
    const float* proposals;   // [B][Rcap][4]
    const int* prop_count;    // [B]
    int B, Rcap, K, ld;
    float wx, wy, ww, wh, scale_clamp, score_thresh, img_h, img_w;
    const int* img_hw;        // optional device [B][2]: per-image (h, w) to clip to
    const float* thresh_img;  // optional device [B]: per-image score floor (>= score_thresh) instead of score_thresh
    float* dense_boxes;       // [B][Rcap*K][4] decoded + clipped boxes
    unsigned long long* keys; // [B][ccap] compacted candidate keys (zeroed by the caller)
    int* cand_count;          // [B] (zeroed by the caller)
    int ccap;
    int* overflow;            // set to 1 when an image has more than ccap candidates
};

__global__ void box_candidates_kernel(const CandArgs a) {
    const int total = a.B * a.Rcap;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int b = t / a.Rcap, r = t - b * a.Rcap;
        if (r >= a.prop_count[b]) continue;
        const float* row = a.pred + (size_t)t * a.ld;
        const float* pb = a.proposals + (size_t)t * 4;
        // softmax (torch: exp(x - max) / sum)
        float mx = row[0];
        for (int k = 1; k <= a.K; ++k) mx = fmaxf(mx, row[k]);
        float sum = 0.f;
        for (int k = 0; k <= a.K; ++k) sum = __fadd_rn(sum, expf(__fsub_rn(row[k], mx)));
        bool all_finite = isfinite(sum);
        const float w = __fsub_rn(pb[2], pb[0]), h = __fsub_rn(pb[3], pb[1]);
        const float cx = __fadd_rn(pb[0], __fmul_rn(0.5f, w)), cy = __fadd_rn(pb[1], __fmul_rn(0.5f, h));
        float* ob = a.dense_boxes + ((size_t)b * a.Rcap * a.K + (size_t)r * a.K) * 4;
        for (int k = 0; k < a.K; ++k) {
            const float* d = row + a.K + 1 + 4 * k;
            const float dx = __fdiv_rn(d[0], a.wx), dy = __fdiv_rn(d[1], a.wy);
            const float dw = fminf(__fdiv_rn(d[2], a.ww), a.scale_clamp), dh = fminf(__fdiv_rn(d[3], a.wh), a.scale_clamp);
            const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
            const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
            float x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw)), y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
            float x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw)), y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
            all_finite = all_finite && isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
            const float ch = a.img_hw ? (float)a.img_hw[2 * b] : a.img_h, cw = a.img_hw ? (float)a.img_hw[2 * b + 1] : a.img_w;
            x1 = fminf(fmaxf(x1, 0.f), cw); x2 = fminf(fmaxf(x2, 0.f), cw);
            y1 = fminf(fmaxf(y1, 0.f), ch); y2 = fminf(fmaxf(y2, 0.f), ch);
            ob[4 * k + 0] = x1; ob[4 * k + 1] = y1; ob[4 * k + 2] = x2; ob[4 * k + 3] = y2;
        }
        if (!all_finite) continue;
        for (int k = 0; k < a.K; ++k) {
            const float p = __fdiv_rn(expf(__fsub_rn(row[k], mx)), sum);
            bool take = p > (a.thresh_img ? a.thresh_img[b] : a.score_thresh);
            // one atomic per (wave, image) instead of one per candidate: thousands of adds to the same 8 counters were 4/5 of the
            // kernel.  Slot order is arbitrary either way (the candidates are sorted by key afterwards).
            while (true) {
                const unsigned long long pending = __ballot(take);
                if (!pending) break;
                const int leader = __ffsll((long long)pending) - 1;
                const int bsel = __shfl(b, leader, 64);
                const unsigned long long mine = __ballot(take && b == bsel);
                int base = 0;
                if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&a.cand_count[bsel], __popcll(mine));
                base = __shfl(base, leader, 64);
                if (take && b == bsel) {
                    const int lane = threadIdx.x & 63;
                    const int slot = base + __popcll(mine & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                    if (slot < a.ccap) a.keys[(size_t)b * a.ccap + slot] = amp::make_sortkey(amp::f2ord(p), r * a.K + k, k);
                    else *a.overflow = 1;
                    take = false;
                }
            }
        }
    }
}

__global__ void gather_dets_kernel(int B, int cap, int D, const float* sboxes, const float* sscores, const int* scats,
                                   const int* keep_idx, const int* keep_count, float* det_boxes, float* det_scores,
                                   int* det_classes, const int* payload_in, int* payload_out) {
    const int total = B * D;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int b = t / D, i = t - b * D;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, s = 0.f;
        int c = -1, pl = -1;
        if (i < keep_count[b]) {
            const int src = keep_idx[(size_t)b * D + i];
            const float* p = sboxes + ((size_t)b * cap + src) * 4;
            x1 = p[0]; y1 = p[1]; x2 = p[2]; y2 = p[3];
            s = sscores[(size_t)b * cap + src];
            c = scats[(size_t)b * cap + src];
            if (payload_in) pl = payload_in[(size_t)b * cap + src];
        }
        float* o = det_boxes + (size_t)t * 4;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        det_scores[t] = s;
        det_classes[t] = c;
        if (payload_out) payload_out[t] = pl;
    }
}

// compact list of the detections of a batch: row off[b] + i <- (image b, detection i), off = exclusive prefix of min(count, D)
__global__ void compact_dets_kernel(int B, int D, const int* __restrict__ det_count, const float* __restrict__ det_boxes,
                                    const float* __restrict__ det_scores, const int* __restrict__ det_classes, float* __restrict__ boxes,
                                    float* __restrict__ scores, int* __restrict__ classes, int* __restrict__ batch, int* __restrict__ n_total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && n_total) {          // the length of the compact list, for kernels that are queued before the host knows it
        int s = 0;
        for (int q = 0; q < B; ++q) s += min(det_count[q], D);
        *n_total = s;
    }
    if (t >= B * D) return;
    const int b = t / D, i = t - b * D;
    if (i >= min(det_count[b], D)) return;
    int off = 0;
    for (int q = 0; q < b; ++q) off += min(det_count[q], D);
    const int o = off + i;
    const float4 v = reinterpret_cast<const float4*>(det_boxes)[t];
    reinterpret_cast<float4*>(boxes)[o] = v;
    scores[o] = det_scores[t];
    classes[o] = det_classes[t];
    batch[o] = b;
}

}  // namespace

extern "C" {

int amp_compact_dets(amp_ctx* ctx, int B, int D, const int* det_count, const float* det_boxes, const float* det_scores, const int* det_classes,
                     float* boxes, float* scores, int* classes, int* batch) {
    return amp::compact_dets_run(ctx, B, D, det_count, det_boxes, det_scores, det_classes, boxes, scores, classes, batch, nullptr);
}


int amp_box_candidates(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B,
                       int Rcap, int K, const float reg_weights[4], float score_thresh, int img_h, int img_w,
                       float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow) {
    return amp_box_candidates_sized(ctx, pred, ld, proposals, prop_count, B, Rcap, K, reg_weights, score_thresh, img_h, img_w, nullptr,
                                    dense_boxes, keys, ccap, cand_count, overflow);
}

int amp_box_candidates_sized(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B,
                             int Rcap, int K, const float reg_weights[4], float score_thresh, int img_h, int img_w, const int* img_hw,
                             float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow) {
    return amp::box_candidates_run(ctx, pred, ld, proposals, prop_count, B, Rcap, K, reg_weights, score_thresh, img_h, img_w, img_hw, nullptr,
                                   dense_boxes, keys, ccap, cand_count, overflow);
}

}  // extern "C"

int amp::compact_dets_run(amp_ctx* ctx, int B, int D, const int* det_count, const float* det_boxes, const float* det_scores, const int* det_classes,
                          float* boxes, float* scores, int* classes, int* batch, int* n_total) {
    AMP_REQUIRE(ctx && det_count && det_boxes && det_scores && det_classes && boxes && scores && classes && batch && B >= 1 && D >= 1,
                "amp_compact_dets: bad argument");
    hipLaunchKernelGGL(compact_dets_kernel, dim3(amp::cdiv(B * D, 256)), dim3(256), 0, ctx->stream, B, D, det_count, det_boxes, det_scores,
                       det_classes, boxes, scores, classes, batch, n_total);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

// thresh_img: optional device [B] per-image score floors (model.hip raises them for an image with more than ccap candidates)
int amp::box_candidates_run(amp_ctx* ctx, const float* pred, int ld, const float* proposals, const int* prop_count, int B, int Rcap, int K,
                            const float reg_weights[4], float score_thresh, int img_h, int img_w, const int* img_hw, const float* thresh_img,
                            float* dense_boxes, unsigned long long* keys, int ccap, int* cand_count, int* overflow) {
    AMP_REQUIRE(ctx && pred && proposals && prop_count && reg_weights && dense_boxes && keys && cand_count && overflow,
                "amp_box_candidates: null argument");
    AMP_REQUIRE(B >= 1 && Rcap >= 1 && K >= 1 && K <= 255 && ld >= 5 * K + 1, "amp_box_candidates: bad shape");
    AMP_REQUIRE((long long)Rcap * K < (1 << 24), "amp_box_candidates: Rcap*K too large for the sort key");
    CandArgs a;
    a.pred = pred; a.proposals = proposals; a.prop_count = prop_count;
    a.B = B; a.Rcap = Rcap; a.K = K; a.ld = ld;
    a.wx = reg_weights[0]; a.wy = reg_weights[1]; a.ww = reg_weights[2]; a.wh = reg_weights[3];
    a.scale_clamp = (float)log(1000.0 / 16.0);
    a.score_thresh = score_thresh; a.img_h = (float)img_h; a.img_w = (float)img_w; a.img_hw = img_hw;
    a.thresh_img = thresh_img;
    a.dense_boxes = dense_boxes; a.keys = keys; a.cand_count = cand_count; a.ccap = ccap; a.overflow = overflow;
    AMP_HIP_CHECK(hipMemsetAsync(keys, 0, (size_t)B * ccap * sizeof(unsigned long long), ctx->stream));
    AMP_HIP_CHECK(hipMemsetAsync(cand_count, 0, (size_t)B * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(box_candidates_kernel, dim3(amp::cdiv(B * Rcap, 128)), dim3(128), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

extern "C" {

int amp_gather_dets(amp_ctx* ctx, int B, int cap, int D, const float* sboxes, const float* sscores, const int* scats,
                    const int* keep_idx, const int* keep_count, float* det_boxes, float* det_scores, int* det_classes,
                    const int* payload_in, int* payload_out) {
    AMP_REQUIRE(ctx && sboxes && sscores && scats && keep_idx && keep_count && det_boxes && det_scores && det_classes,
                "amp_gather_dets: null argument");
    hipLaunchKernelGGL(gather_dets_kernel, dim3(amp::cdiv(B * D, 256)), dim3(256), 0, ctx->stream, B, cap, D, sboxes,
                       sscores, scats, keep_idx, keep_count, det_boxes, det_scores, det_classes, payload_in, payload_out);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
