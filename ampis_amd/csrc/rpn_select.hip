// RPN proposal candidate selection (detectron2 proposal_utils.find_top_rpn_proposals, first half):
//   1. per (image, level): top-k anchors by objectness logit (k = min(PRE_NMS_TOPK, H*W*A)), order = logit descending,
//      ties by ascending anchor index (the order oracle/maskrcnn.py fixes; torch.topk leaves ties unspecified)
//   2. decode the selected anchors (Box2BoxTransform.apply_deltas, weights 1,1,1,1, scale clamp ln(1000/16)), clip to the
//      image, mark non-finite / empty (w <= 0 or h <= 0) boxes invalid
//   3. per image: order all levels' candidates by (logit desc, concatenated position asc), invalid last
// Integer / ordering work is exact; the decode arithmetic follows the oracle's fp32 op order (no contraction).
//
// Kernel 1 is a radix select (4 x 8-bit digits on the order-preserving uint32 image of the logit) done by one 1024-thread
// workgroup per segment with LDS histograms, then a wave-ballot compaction in index order and an LDS bitonic sort of the
// <= 2048 survivors.  Kernel 3 is an LDS bitonic sort of 64-bit (key, ~position) words, <= 16384 per image.
#include "common.h"

namespace {

constexpr int TOPK_THREADS = 1024;
constexpr int MAX_LEVELS = 5;

using amp::f2ord;
using amp::ord2f;

struct TopkArgs {
    const float* pred[MAX_LEVELS];   // RPN predictor output per level: [B, H*W, ld] with logits in columns [0, A)
    int hw[MAX_LEVELS];
    int nlevels, A, ld, k;
    uint32_t* keys_scratch;          // [B*nlevels][max_n] ordered keys (coalesced re-reads)
    int max_n;
    int* sel_idx;                    // [B][nlevels][k] anchor index within the level, sorted
    float* sel_logit;                // [B][nlevels][k]
    int* sel_count;                  // [B][nlevels]
    // A level with more than TOPK_CHUNK anchors is cut into nch[l] chunks of len[l] anchors, one workgroup each: the top-k of the
    // level is contained in the union of the chunks' top-k lists, which topk_merge_kernel orders exactly (64-bit words
    // (key << 32) | ~anchor index: logit descending, ties by ascending anchor index, as in the one-workgroup path).
    int nch[MAX_LEVELS], len[MAX_LEVELS], seg0[MAX_LEVELS];   // seg0[l]: first chunk-segment of level l inside an image
    int segs_per_img;
    unsigned long long* cand;        // [B][segs_per_img][k] sorted words of the chunk segments (levels with nch > 1)
    int* cand_count;                 // [B][segs_per_img]
};

}  // namespace
#include "select.h"
namespace {

constexpr int TOPK_CHUNK = 49152;    // anchors per workgroup: the 196 608 anchors of p2 of a 1024 x 1024 image become 4 chunks (8 chunks of 24 576: the select halves, 87 -> 46 us, and the merge of eight lists takes it back, 18 -> 55 us; round 2: 8 chunks were
                                     // slower: the select has ~100 us of fixed cost and the merge sorts twice as many words)

__global__ __launch_bounds__(TOPK_THREADS) void rpn_topk_kernel(const TopkArgs a) {
    __shared__ amp::SelectSmem sm;
    const int b = blockIdx.x / a.segs_per_img, s_in = blockIdx.x % a.segs_per_img;
    int lvl = 0;
    while (lvl + 1 < a.nlevels && s_in >= a.seg0[lvl + 1]) ++lvl;
    const int c = s_in - a.seg0[lvl];
    const int seg = b * a.nlevels + lvl;
    const int n_lvl = a.hw[lvl] * a.A;
    const int i0 = c * a.len[lvl];
    const int n = min(a.len[lvl], n_lvl - i0);
    const float* pred = a.pred[lvl] + (size_t)b * a.hw[lvl] * a.ld;
    uint32_t* keys = a.keys_scratch + (size_t)seg * a.max_n + i0;
    const int A = a.A, ld = a.ld;
    // key = order-preserving image of the logit; (logit desc, index asc) == (key desc, index asc)
    auto key_fn = [&](int i) {
        const int g = i0 + i, pix = g / A, an = g - pix * A;
        uint32_t key = f2ord(pred[(size_t)pix * ld + an]);
        return key ? key : 1u;   // 0 is the "not a candidate" marker of select_topk (only a negative NaN maps there)
    };
    // chunks are at most TOPK_CHUNK = 48 * 1024 keys: they live in registers; anything larger (nch capped by the merge) goes
    // through the scratch copy
    const int k = (n <= TOPK_CHUNK) ? amp::select_topk_reg<TOPK_CHUNK / 1024>(sm, n, a.k, key_fn) : amp::select_topk(sm, n, a.k, keys, key_fn);
    if (a.nch[lvl] == 1) {
        for (int i = threadIdx.x; i < k; i += TOPK_THREADS) {
            const unsigned long long wv = sm.sorted[i];
            a.sel_idx[(size_t)seg * a.k + i] = (int)(0xffffffffu - (uint32_t)(wv & 0xffffffffu));
            a.sel_logit[(size_t)seg * a.k + i] = ord2f((uint32_t)(wv >> 32));
        }
        if (threadIdx.x == 0) a.sel_count[seg] = k;
    } else {
        // chunk-local index -> anchor index of the level: ~(i0 + i) = ~i - i0 in the low word
        unsigned long long* out = a.cand + (size_t)blockIdx.x * a.k;
        for (int i = threadIdx.x; i < k; i += TOPK_THREADS) out[i] = sm.sorted[i] - (unsigned long long)(uint32_t)i0;
        if (threadIdx.x == 0) a.cand_count[blockIdx.x] = k;
    }
}

// one workgroup per (image, level with chunks): exact order of the chunks' candidates, first k out.  The chunks' lists are sorted and
// their words unique ((key, ~anchor index)): the rank of a word in the merge is its own index plus, per other list, the number of larger
// words (binary search in LDS) -- no sort.
__global__ __launch_bounds__(TOPK_THREADS) void topk_merge_kernel(const TopkArgs a, int lvl_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long mkeys[];        // [nch][k]
    __shared__ int ccount[8];
    const int b = blockIdx.x;
    int lvl = 0;                        // blockIdx.y-th level among those cut into chunks (bits of lvl_mask): all of them in one launch
    for (int seen = -1; lvl < MAX_LEVELS; ++lvl)
        if ((lvl_mask >> lvl) & 1) { if (++seen == (int)blockIdx.y) break; }
    const int nch = a.nch[lvl];
    if (threadIdx.x < nch) ccount[threadIdx.x] = min(a.cand_count[b * a.segs_per_img + a.seg0[lvl] + threadIdx.x], a.k);
    __syncthreads();
    int total = 0;
    for (int c = 0; c < nch; ++c) total += ccount[c];
    for (int i = threadIdx.x; i < nch * a.k; i += TOPK_THREADS) {
        const int c = i / a.k, j = i - c * a.k;
        mkeys[i] = (j < ccount[c]) ? a.cand[(size_t)(b * a.segs_per_img + a.seg0[lvl] + c) * a.k + j] : 0ull;
    }
    __syncthreads();
    const int k = min(a.k, total);
    const int seg = b * a.nlevels + lvl;
    for (int i = threadIdx.x; i < nch * a.k; i += TOPK_THREADS) {
        const int c = i / a.k, j = i - c * a.k;
        if (j >= ccount[c]) continue;
        const unsigned long long wv = mkeys[i];
        int rank = j;
        for (int c2 = 0; c2 < nch; ++c2) {
            if (c2 == c) continue;
            const unsigned long long* s = mkeys + c2 * a.k;
            int lo = 0, hi = ccount[c2];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s[mid] > wv) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            a.sel_idx[(size_t)seg * a.k + rank] = (int)(0xffffffffu - (uint32_t)(wv & 0xffffffffu));
            a.sel_logit[(size_t)seg * a.k + rank] = ord2f((uint32_t)(wv >> 32));
        }
    }
    if (threadIdx.x == 0) a.sel_count[seg] = k;
}

struct DecodeArgs {
    const float* pred[MAX_LEVELS];
    int hw[MAX_LEVELS], fw[MAX_LEVELS], stride[MAX_LEVELS];
    float cell[MAX_LEVELS][3][4];    // cell anchors per level (A = 3)
    int nlevels, A, ld, k;
    const int* sel_idx;
    const float* sel_logit;
    const int* sel_count;
    float img_h, img_w;
    const int* img_hw;               // optional device [B][2]: per-image (h, w) to clip to instead of img_h / img_w
    float scale_clamp;
    int cap;                         // capacity per image of the outputs (>= nlevels*k)
    float* boxes;                    // [B][cap][4] clipped boxes
    unsigned long long* sortkey;     // [B][cap]: (ord(logit) << 32) | ~pos, 0 when invalid / unused
    int* anchor_id;                  // [B][cap] global anchor index (level offset + index), -1 for unused slots; may be null
    int lvl_off[MAX_LEVELS];
};

__global__ void rpn_decode_kernel(const DecodeArgs a, int B) {
    const int per_img = a.nlevels * a.k;
    const int total = B * a.cap;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int b = t / a.cap, pos = t - b * a.cap;
        unsigned long long key = 0ull;
        int aid = -1;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
        // position in the level-major concatenation: level lvl contributes sel_count[b][lvl] entries
        if (pos < per_img) {
            int lvl = 0, off = pos;
            while (lvl < a.nlevels && off >= a.sel_count[b * a.nlevels + lvl]) { off -= a.sel_count[b * a.nlevels + lvl]; ++lvl; }
            if (lvl < a.nlevels) {
                const int seg = b * a.nlevels + lvl;
                const int idx = a.sel_idx[(size_t)seg * a.k + off];
                const float logit = a.sel_logit[(size_t)seg * a.k + off];
                aid = a.lvl_off[lvl] + idx;
                const int pix = idx / a.A, an = idx - pix * a.A;
                const int py = pix / a.fw[lvl], px = pix - py * a.fw[lvl];
                const float sx = (float)(px * a.stride[lvl]), sy = (float)(py * a.stride[lvl]);
                const float ax1 = __fadd_rn(sx, a.cell[lvl][an][0]), ay1 = __fadd_rn(sy, a.cell[lvl][an][1]);
                const float ax2 = __fadd_rn(sx, a.cell[lvl][an][2]), ay2 = __fadd_rn(sy, a.cell[lvl][an][3]);
                const float* d = a.pred[lvl] + ((size_t)b * a.hw[lvl] + pix) * a.ld + a.A + an * 4;
                const float w = __fsub_rn(ax2, ax1), h = __fsub_rn(ay2, ay1);
                const float cx = __fadd_rn(ax1, __fmul_rn(0.5f, w)), cy = __fadd_rn(ay1, __fmul_rn(0.5f, h));
                const float dx = d[0], dy = d[1];
                const float dw = fminf(d[2], a.scale_clamp), dh = fminf(d[3], a.scale_clamp);
                const float pcx = __fadd_rn(__fmul_rn(dx, w), cx), pcy = __fadd_rn(__fmul_rn(dy, h), cy);
                const float pw = __fmul_rn(expf(dw), w), ph = __fmul_rn(expf(dh), h);
                x1 = __fsub_rn(pcx, __fmul_rn(0.5f, pw));
                y1 = __fsub_rn(pcy, __fmul_rn(0.5f, ph));
                x2 = __fadd_rn(pcx, __fmul_rn(0.5f, pw));
                y2 = __fadd_rn(pcy, __fmul_rn(0.5f, ph));
                const bool finite = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(logit);
                const float ch = a.img_hw ? (float)a.img_hw[2 * b] : a.img_h, cw = a.img_hw ? (float)a.img_hw[2 * b + 1] : a.img_w;
                x1 = fminf(fmaxf(x1, 0.f), cw); x2 = fminf(fmaxf(x2, 0.f), cw);
                y1 = fminf(fmaxf(y1, 0.f), ch); y2 = fminf(fmaxf(y2, 0.f), ch);
                const bool nonempty = (__fsub_rn(x2, x1) > 0.f) && (__fsub_rn(y2, y1) > 0.f);
                if (finite && nonempty) {
                    key = amp::make_sortkey(f2ord(logit), pos, lvl);
                }
            }
        }
        float* o = a.boxes + (size_t)t * 4;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        a.sortkey[t] = key;
        if (a.anchor_id) a.anchor_id[t] = aid;
    }
}

// Per image: sort cap (<= 8192) 64-bit keys descending; emit sorted boxes / scores / categories and the valid count.
constexpr int SORT_THREADS = 1024;
__global__ __launch_bounds__(SORT_THREADS) void sort_gather_kernel(const unsigned long long* sortkey, const float* boxes_in,
                                                                  int cap, int box_stride, int Nmax, float* boxes_out,
                                                                  float* score_out, int* cat_out, int* count_out,
                                                                  int* pos_out, const int* payload_in, int* payload_out, const int* n_used) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int b = blockIdx.x;
    int N = Nmax;
    // n_used[b]: the words of image b sit in its first n_used slots and the rest are zero (amp_box_candidates' compacted list): sort the
    // smallest power of two that holds them instead of all `cap` slots
    if (n_used) {
        const int nv = min(n_used[b], cap);
        int nn = 64;
        while (nn < nv) nn <<= 1;
        N = min(N, nn);
    }
    for (int i = threadIdx.x; i < N; i += SORT_THREADS) skeys[i] = (i < cap) ? sortkey[(size_t)b * cap + i] : 0ull;
    __syncthreads();
    amp::bitonic_desc<SORT_THREADS>(skeys, N);
    int cnt = 0;
    for (int i = threadIdx.x; i < cap; i += SORT_THREADS) {
        const unsigned long long kv = (i < N) ? skeys[i] : 0ull;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, sc = 0.f;
        int cat = -1, pos = -1;
        if (kv != 0ull) {
            pos = amp::sortkey_pos(kv);
            cat = amp::sortkey_cat(kv);
            sc = ord2f((uint32_t)(kv >> 32));
            const float* s = boxes_in + ((size_t)b * box_stride + pos) * 4;
            x1 = s[0]; y1 = s[1]; x2 = s[2]; y2 = s[3];
            ++cnt;
        }
        float* o = boxes_out + ((size_t)b * cap + i) * 4;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        score_out[(size_t)b * cap + i] = sc;
        cat_out[(size_t)b * cap + i] = cat;
        if (pos_out) pos_out[(size_t)b * cap + i] = pos;
        if (payload_out) payload_out[(size_t)b * cap + i] = (pos >= 0 && payload_in) ? payload_in[(size_t)b * box_stride + pos] : -1;
    }
    // block reduce of the valid count (counter lives behind the keys: keep all LDS in the one dynamic region)
    int* s_cnt = reinterpret_cast<int*>(skeys + Nmax);
    __syncthreads();
    if (threadIdx.x == 0) *s_cnt = 0;
    __syncthreads();
    atomicAdd(s_cnt, cnt);
    __syncthreads();
    if (threadIdx.x == 0) count_out[b] = *s_cnt;
}

}  // namespace

extern "C" {

int amp_rpn_topk(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, uint32_t* keys_scratch, int max_n, int* sel_idx,
                 float* sel_logit, int* sel_count) {
    AMP_REQUIRE(ctx && lv && keys_scratch && sel_idx && sel_logit && sel_count, "amp_rpn_topk: null argument");
    AMP_REQUIRE(lv->nlevels >= 1 && lv->nlevels <= MAX_LEVELS && lv->A == 3, "amp_rpn_topk: need 1..5 levels, A == 3");
    AMP_REQUIRE(k >= 1 && k <= amp::SELECT_MAX_K, "amp_rpn_topk: k=%d out of range [1,%d]", k, amp::SELECT_MAX_K);
    TopkArgs a;
    for (int l = 0; l < lv->nlevels; ++l) {
        AMP_REQUIRE(lv->pred[l] && lv->h[l] > 0 && lv->w[l] > 0, "amp_rpn_topk: missing level %d", l);
        AMP_REQUIRE(lv->h[l] * lv->w[l] * lv->A <= max_n, "amp_rpn_topk: max_n=%d too small for level %d", max_n, l);
        a.pred[l] = lv->pred[l];
        a.hw[l] = lv->h[l] * lv->w[l];
    }
    a.nlevels = lv->nlevels; a.A = lv->A; a.ld = lv->ld; a.k = k;
    a.keys_scratch = keys_scratch; a.max_n = max_n;
    a.sel_idx = sel_idx; a.sel_logit = sel_logit; a.sel_count = sel_count;
    int segs = 0;
    bool chunked = false;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        a.seg0[l] = segs;
        if (l >= lv->nlevels) { a.nch[l] = 1; a.len[l] = 0; continue; }
        const int n = a.hw[l] * a.A;
        a.nch[l] = std::min(8, std::max(1, amp::cdiv(n, TOPK_CHUNK)));
        if ((long long)a.nch[l] * k > 8192) a.nch[l] = std::max(1, 8192 / k);     // the merge sorts nch * k words in LDS
        a.len[l] = amp::cdiv(amp::cdiv(n, a.nch[l]), 64) * 64;
        a.nch[l] = amp::cdiv(n, a.len[l]);
        chunked = chunked || a.nch[l] > 1;
        segs += a.nch[l];
    }
    a.segs_per_img = segs;
    a.cand = nullptr; a.cand_count = nullptr;
    if (chunked) {
        const size_t need = (size_t)B * segs * ((size_t)k * sizeof(unsigned long long) + sizeof(int));
        if (ctx->topk_bytes < need) {      // grown on demand, once per shape class
            AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (ctx->topk_scratch) AMP_HIP_CHECK(hipFree(ctx->topk_scratch));
            ctx->topk_scratch = nullptr; ctx->topk_bytes = 0;
            AMP_HIP_CHECK(hipMalloc(&ctx->topk_scratch, need));
            ctx->topk_bytes = need;
        }
        a.cand = reinterpret_cast<unsigned long long*>(ctx->topk_scratch);
        a.cand_count = reinterpret_cast<int*>(a.cand + (size_t)B * segs * k);
    }
    hipLaunchKernelGGL(rpn_topk_kernel, dim3(B * segs), dim3(TOPK_THREADS), 0, ctx->stream, a);
    int lvl_mask = 0, nmerge = 0, nmax = 0;
    for (int l = 0; l < lv->nlevels; ++l)
        if (a.nch[l] > 1) { lvl_mask |= 1 << l; ++nmerge; nmax = std::max(nmax, a.nch[l] * k); }
    if (nmerge) {
        AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(nmax * sizeof(unsigned long long))));
        hipLaunchKernelGGL(topk_merge_kernel, dim3(B, nmerge), dim3(TOPK_THREADS), (size_t)nmax * sizeof(unsigned long long), ctx->stream, a, lvl_mask);
    }
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_rpn_decode(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, const int* sel_idx, const float* sel_logit,
                   const int* sel_count, int img_h, int img_w, int cap, float* boxes, unsigned long long* sortkey, int* anchor_id) {
    return amp_rpn_decode_sized(ctx, lv, B, k, sel_idx, sel_logit, sel_count, img_h, img_w, nullptr, cap, boxes, sortkey, anchor_id);
}

int amp_rpn_decode_sized(amp_ctx* ctx, const amp_rpn_levels* lv, int B, int k, const int* sel_idx, const float* sel_logit,
                         const int* sel_count, int img_h, int img_w, const int* img_hw, int cap, float* boxes,
                         unsigned long long* sortkey, int* anchor_id) {
    AMP_REQUIRE(ctx && lv && sel_idx && sel_logit && sel_count && boxes && sortkey, "amp_rpn_decode: null argument");
    AMP_REQUIRE(lv->nlevels >= 1 && lv->nlevels <= MAX_LEVELS && lv->A == 3, "amp_rpn_decode: need 1..5 levels, A == 3");
    AMP_REQUIRE(cap >= lv->nlevels * k, "amp_rpn_decode: cap=%d < nlevels*k", cap);
    DecodeArgs a;
    for (int l = 0; l < lv->nlevels; ++l) {
        a.pred[l] = lv->pred[l];
        a.hw[l] = lv->h[l] * lv->w[l];
        a.fw[l] = lv->w[l];
        a.stride[l] = lv->stride[l];
        for (int r = 0; r < 3; ++r) {
            // detectron2 generate_cell_anchors: python-float math, stored fp32
            const double ratio = (r == 0) ? 0.5 : (r == 1 ? 1.0 : 2.0);
            const double area = (double)lv->anchor_size[l] * (double)lv->anchor_size[l];
            const double w = sqrt(area / ratio), h = ratio * w;
            a.cell[l][r][0] = (float)(-w / 2.0); a.cell[l][r][1] = (float)(-h / 2.0);
            a.cell[l][r][2] = (float)(w / 2.0);  a.cell[l][r][3] = (float)(h / 2.0);
        }
    }
    a.nlevels = lv->nlevels; a.A = lv->A; a.ld = lv->ld; a.k = k;
    a.sel_idx = sel_idx; a.sel_logit = sel_logit; a.sel_count = sel_count;
    a.img_h = (float)img_h; a.img_w = (float)img_w; a.img_hw = img_hw;
    a.scale_clamp = (float)log(1000.0 / 16.0);
    a.cap = cap; a.boxes = boxes; a.sortkey = sortkey; a.anchor_id = anchor_id;
    a.lvl_off[0] = 0;
    for (int l = 1; l < MAX_LEVELS; ++l) a.lvl_off[l] = (l <= lv->nlevels) ? a.lvl_off[l - 1] + lv->h[l - 1] * lv->w[l - 1] * lv->A : 0;
    const int total = B * cap;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(amp::cdiv(total, 256)), dim3(256), 0, ctx->stream, a, B);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_sort_gather(amp_ctx* ctx, int B, int cap, int box_stride, const unsigned long long* sortkey, const float* boxes_in,
                    float* boxes_out, float* score_out, int* cat_out, int* count_out, int* pos_out, const int* payload_in,
                    int* payload_out) {
    return amp_sort_gather_n(ctx, B, cap, box_stride, sortkey, boxes_in, boxes_out, score_out, cat_out, count_out, pos_out, payload_in,
                             payload_out, nullptr);
}

int amp_sort_gather_n(amp_ctx* ctx, int B, int cap, int box_stride, const unsigned long long* sortkey, const float* boxes_in,
                      float* boxes_out, float* score_out, int* cat_out, int* count_out, int* pos_out, const int* payload_in,
                      int* payload_out, const int* n_used) {
    AMP_REQUIRE(ctx && sortkey && boxes_in && boxes_out && score_out && cat_out && count_out, "amp_sort_gather: null argument");
    AMP_REQUIRE(B >= 1 && cap >= 1 && cap <= 16384, "amp_sort_gather: cap=%d out of range [1,16384]", cap);
    int N = 64;
    while (N < cap) N <<= 1;
    const size_t smem = (size_t)N * 8 + 16;
    AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sort_gather_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(sort_gather_kernel, dim3(B), dim3(SORT_THREADS), smem, ctx->stream, sortkey, boxes_in, cap, box_stride, N,
                       boxes_out, score_out, cat_out, count_out, pos_out, payload_in, payload_out, n_used);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
