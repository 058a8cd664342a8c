// Batched greedy NMS on score-sorted boxes (torchvision.ops.batched_nms, vanilla semantics:
// boxes of different categories never suppress each other, exact coordinates, suppress when IoU > thresh).
//
// Two kernels per call, all images of the batch at once:
//   nms_mask_kernel : 64x64 tiles of the upper-triangular suppression matrix, one wave per tile; lane i of a wave owns box i of the row tile
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

// (grid: NMS_MASK_BLOCKS workgroups of four waves per image walk the tiles the image's count needs -- the count is only known on the
//  device, and a grid for the capacity, 128 x 128 tiles of which a 1500-box image uses 24 x 24, spent the kernel's time starting empty
//  workgroups)
constexpr int NMS_MASK_BLOCKS = 128;
__global__ __launch_bounds__(256) void nms_mask_kernel(const NmsArgs a) {
    __shared__ float sb_all[4][64][4];
    __shared__ int sc_all[4][64];
    const int b = blockIdx.y;
    const int n = min(a.counts[b], a.cap);
    const int nW = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float (*sb)[4] = sb_all[wave];
    int* sc = sc_all[wave];
    for (int t = blockIdx.x * 4 + wave; t < nW * nW; t += NMS_MASK_BLOCKS * 4) {
        const int rb = t / nW, cb = t - rb * nW;
        const int i = rb * 64 + lane;
        unsigned long long* mrow = a.mask + ((size_t)b * a.cap + i) * a.W + cb;
        if (cb < rb) {
            if (i < n) *mrow = 0ull;
            continue;
        }
        const int j0 = cb * 64;
        __builtin_amdgcn_wave_barrier();          // the previous tile's readers are done (same wave: LDS accesses stay in order)
        {
            const int j = j0 + lane;
            if (j < n) {
                const float* p = a.boxes + ((size_t)b * a.cap + j) * 4;
                sb[lane][0] = p[0]; sb[lane][1] = p[1]; sb[lane][2] = p[2]; sb[lane][3] = p[3];
                sc[lane] = a.cats[(size_t)b * a.cap + j];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (i >= n) continue;
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
        // wave-uniform scalar loop, in score order, over the alive boxes that overlap another alive box of this chunk: one still alive at
        // its turn is kept and removes its row; the other alive boxes are kept whatever the order
        unsigned long long cur = ~removed & valid;
        unsigned long long work = __ballot((diag & cur) != 0ull) & cur;
        while (work != 0ull) {
            const int bit = __builtin_ctzll(work);
            work &= work - 1ull;
            if ((cur >> bit) & 1ull) {
                // readlane returns a signed int: cast both halves to unsigned before widening (no sign extension)
                const unsigned int dm_hi = (unsigned int)__builtin_amdgcn_readlane((int)dhi, bit);
                const unsigned int dm_lo = (unsigned int)__builtin_amdgcn_readlane((int)dlo, bit);
                cur &= ~(((unsigned long long)dm_hi << 32) | (unsigned long long)dm_lo);
            }
        }
        const unsigned long long keep = cur;
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


// ---------------------------------------------------------------------------------------------------------------------------------
// RPN proposals: NMS per (image, level) segment + merge.  batched_nms uses the level as category, so boxes of different levels never
// meet: instead of ONE score-ordered list of 5 k candidates per image (79 chunks walked by one wave, a 79 x 79-tile matrix, a
// 8192-word sort in front) each level's top-k list -- already in (logit desc, anchor asc) order -- is its own problem of <= 16 (32)
// chunks, and the first post_nms_topk of the image are a merge of L sorted lists of survivors.  Same decisions as the one-list path
// (greedy NMS restricted to a category IS the category's greedy NMS), same order (the sort word's position breaks ties between levels).
struct LvlArgs {
    const float* boxes;                 // [B][cap][4] decoded candidates: level l of image b at [off, off + sel_count[b][l]), off = counts before
    const unsigned long long* keys;     // [B][cap] sort words of those candidates, 0 = invalid (non-finite / empty box)
    const int* sel_count;               // [B][L]
    int L, cap, k, nchk;                // nchk = ceil(k / 64): chunks per segment at most
    float thresh;
    unsigned long long* mt;             // [B*L][nchk][nchk*64] suppression words, see lvl_mask_kernel
    unsigned long long* kept;           // [B*L][k] sort words of the survivors, in order
    int* kept_count;                    // [B*L]
    int max_keep;
};

__device__ __forceinline__ int lvl_offset(const LvlArgs& a, int b, int lvl) {
    int off = 0;
    for (int l = 0; l < lvl; ++l) off += a.sel_count[b * a.L + l];
    return off;
}

// One wave per 64 x 64 tile (rb <= cb) of a segment.  Lane = box cb*64 + lane; it meets the 64 boxes of chunk rb (LDS).
//   rb <  cb: word [rb][cb*64 + lane], bit ii = "box rb*64+ii (higher score) suppresses this box"     (column-oriented)
//   rb == cb: word [cb][cb*64 + lane], bit jj = "this box suppresses box cb*64+jj", jj > lane         (row-oriented: the scan's scalar loop)
// IoU in torchvision's op order, fp32, no contraction; max / min / the sum of the two areas are commutative, so the value does not depend
// on which of the two boxes the lane holds.  A trip whose 64 pairs are all disjoint skips the division (inter == 0 -> iou 0 or NaN, never
// above a threshold >= 0).
__global__ __launch_bounds__(64) void lvl_mask_kernel(const LvlArgs a) {
    const int seg = blockIdx.z;
    const int n = min(a.sel_count[seg], a.k);
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || cb * 64 >= n) return;
    const int b = seg / a.L, lvl = seg - b * a.L;
    const float4* boxes = reinterpret_cast<const float4*>(a.boxes) + (size_t)b * a.cap + lvl_offset(a, b, lvl);
    const int lane = threadIdx.x;
    __shared__ float4 sb[64];
    {
        const int i = rb * 64 + lane;
        sb[lane] = (i < n) ? boxes[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int j = cb * 64 + lane;
    const float4 me = (j < n) ? boxes[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float area_me = __fmul_rn(__fsub_rn(me.z, me.x), __fsub_rn(me.w, me.y));
    const bool diag = rb == cb;
    const int in = min(64, n - rb * 64);
    unsigned long long bits = 0ull;
    for (int ii = 0; ii < in; ++ii) {
        const float4 o = sb[ii];
        const float xx1 = fmaxf(me.x, o.x), yy1 = fmaxf(me.y, o.y);
        const float xx2 = fminf(me.z, o.z), yy2 = fminf(me.w, o.w);
        const float w = fmaxf(__fsub_rn(xx2, xx1), 0.f), h = fmaxf(__fsub_rn(yy2, yy1), 0.f);
        const float inter = __fmul_rn(w, h);
        const bool pair = (j < n) && (!diag || ii > lane);
        if (__ballot(pair && inter > 0.f) == 0ull) continue;
        const float area_o = __fmul_rn(__fsub_rn(o.z, o.x), __fsub_rn(o.w, o.y));
        const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_me, area_o), inter));
        if (pair && iou > a.thresh) bits |= (1ull << ii);
    }
    const int rowlen = a.nchk * 64;
    a.mt[((size_t)seg * a.nchk + rb) * rowlen + j] = bits;
}

// One 1024-thread workgroup per segment.  Super-blocks of 16 chunks (1024 boxes): every wave folds the decisions of earlier super-blocks
// into "removed" bits of its chunk (coalesced reads of the words, ANDed with the kept words) and copies the super-block's own triangle of
// words into LDS; then wave 0 walks the 16 chunks out of LDS: removed = OR of (word & kept word of the earlier chunk), the chunk's own
// order by the scalar loop over its row-oriented diagonal words.
constexpr int LVL_SB = 16;
__global__ __launch_bounds__(1024) void lvl_scan_kernel(const LvlArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long smt[];     // [LVL_SB][LVL_SB * 64]
    __shared__ unsigned long long skeep[64];                                       // kept word per chunk (nchk <= 64)
    __shared__ unsigned long long spre[LVL_SB];                                    // removed before the super-block's own scan
    __shared__ unsigned long long skey[LVL_SB * 64];                               // the super-block's sort words (the scan must not wait on global loads)
    const int seg = blockIdx.x;
    const int b = seg / a.L, lvl = seg - b * a.L;
    const int n = min(a.sel_count[seg], a.k);
    const int nch = (n + 63) >> 6;
    const int off = lvl_offset(a, b, lvl);
    const unsigned long long* keys = a.keys + (size_t)b * a.cap + off;
    const int rowlen = a.nchk * 64;
    const unsigned long long* mt = a.mt + (size_t)seg * a.nchk * rowlen;
    unsigned long long* out = a.kept + (size_t)seg * a.k;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ int s_nkept;
    if (threadIdx.x == 0) s_nkept = 0;
    for (int sb0 = 0; sb0 < nch; sb0 += LVL_SB) {
        __syncthreads();            // the previous super-block's kept words and count are visible, its LDS words are free
        int nkept = s_nkept;
        if (nkept >= a.max_keep) break;      // uniform: later boxes cannot change earlier decisions
        {
            const int c = sb0 + wave;
            if (c < nch) {
                const int j = c * 64 + lane;
                unsigned long long acc = 0ull;
                for (int rc = 0; rc < sb0; rc += 8) {         // sb0 is a multiple of 16
                    unsigned long long v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = mt[(size_t)(rc + u) * rowlen + j];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc |= v[u] & skeep[rc + u];
                }
                const unsigned long long key = (j < n) ? keys[j] : 0ull;
                const unsigned long long pre = __ballot(acc != 0ull || key == 0ull);
                if (lane == 0) spre[wave] = pre;
                skey[wave * 64 + lane] = key;
                unsigned long long tv[LVL_SB];              // all loads of the triangle's column in flight, then the LDS writes
#pragma unroll
                for (int u = 0; u < LVL_SB; ++u) tv[u] = (u <= wave) ? mt[(size_t)(sb0 + u) * rowlen + j] : 0ull;
#pragma unroll
                for (int u = 0; u < LVL_SB; ++u)
                    if (u <= wave) smt[u * (LVL_SB * 64) + wave * 64 + lane] = tv[u];
            }
        }
        __syncthreads();
        if (wave == 0) {
            const int cend = min(LVL_SB, nch - sb0);
            for (int cc = 0; cc < cend && nkept < a.max_keep; ++cc) {
                const int c = sb0 + cc;
                unsigned long long acc = 0ull;
                for (int r2 = 0; r2 < cc; ++r2) acc |= smt[r2 * (LVL_SB * 64) + cc * 64 + lane] & skeep[sb0 + r2];
                const unsigned long long removed = __ballot(acc != 0ull) | spre[cc];
                const unsigned long long diag = smt[cc * (LVL_SB * 64) + cc * 64 + lane];
                const unsigned int dlo = (unsigned int)(diag & 0xffffffffu), dhi = (unsigned int)(diag >> 32);
                // Only a box that overlaps an alive box of its own chunk can change anything: walk those in score order (a box still alive
                // when its turn comes is kept and removes its row); every other alive box is kept whatever the order.
                unsigned long long cur = ~removed;       // boxes beyond n and invalid ones are in spre
                unsigned long long work = __ballot((diag & cur) != 0ull) & cur;
                while (work != 0ull) {                   // wave-uniform
                    const int bit = __builtin_ctzll(work);
                    work &= work - 1ull;
                    if ((cur >> bit) & 1ull) {
                        const unsigned int dm_hi = (unsigned int)__builtin_amdgcn_readlane((int)dhi, bit);
                        const unsigned int dm_lo = (unsigned int)__builtin_amdgcn_readlane((int)dlo, bit);
                        cur &= ~(((unsigned long long)dm_hi << 32) | (unsigned long long)dm_lo);
                    }
                }
                const unsigned long long keep = cur;
                if (lane == 0) skeep[c] = keep;
                if ((keep >> lane) & 1ull) {
                    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                    out[nkept + __popcll(keep & below)] = skey[cc * 64 + lane];       // nkept <= n <= k
                }
                nkept += __popcll(keep);
            }
            if (lane == 0) s_nkept = nkept;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.kept_count[seg] = min(s_nkept, a.k);
}

// One workgroup per image: rank of every survivor in the merge of the L sorted lists (own index + binary searches in the others; the
// sort words are unique), the first max_keep written in order with their boxes / logits / levels / payloads, the rest of the rows zeroed
// as amp_gather_dets leaves them.
struct MergeArgs {
    const unsigned long long* kept;     // [B*L][k]
    const int* kept_count;              // [B*L]
    const float* boxes;                 // [B][cap][4], indexed by the sort word's position
    const int* payload_in;              // [B][cap] or null
    int L, cap, k, max_keep;
    float* out_boxes; float* out_scores; int* out_cat; int* out_count; int* payload_out;
};
__global__ __launch_bounds__(1024) void lvl_merge_kernel(const MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];        // [L][k]
    __shared__ int scnt[8];
    const int b = blockIdx.x;
    if (threadIdx.x < a.L) scnt[threadIdx.x] = min(a.kept_count[b * a.L + threadIdx.x], a.k);
    __syncthreads();
    int total = 0;
    for (int l = 0; l < a.L; ++l) total += scnt[l];
    for (int i = threadIdx.x; i < a.L * a.k; i += 1024) {
        const int l = i / a.k, j = i - l * a.k;
        sk[i] = (j < scnt[l]) ? a.kept[((size_t)b * a.L + l) * a.k + j] : 0ull;
    }
    __syncthreads();
    const int count = min(total, a.max_keep);
    for (int i = threadIdx.x; i < a.L * a.k; i += 1024) {
        const int l = i / a.k, j = i - l * a.k;
        if (j >= scnt[l]) continue;
        const unsigned long long kv = sk[i];
        int rank = j;
        for (int l2 = 0; l2 < a.L; ++l2) {
            if (l2 == l) continue;
            const unsigned long long* s = sk + l2 * a.k;
            int lo = 0, hi = scnt[l2];                  // number of words > kv in a descending list
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s[mid] > kv) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank >= a.max_keep) continue;
        const int pos = amp::sortkey_pos(kv);
        const size_t o = (size_t)b * a.max_keep + rank;
        reinterpret_cast<float4*>(a.out_boxes)[o] = reinterpret_cast<const float4*>(a.boxes)[(size_t)b * a.cap + pos];
        a.out_scores[o] = amp::ord2f((uint32_t)(kv >> 32));
        a.out_cat[o] = amp::sortkey_cat(kv);
        if (a.payload_out) a.payload_out[o] = a.payload_in ? a.payload_in[(size_t)b * a.cap + pos] : -1;
    }
    for (int r = count + threadIdx.x; r < a.max_keep; r += 1024) {
        const size_t o = (size_t)b * a.max_keep + r;
        reinterpret_cast<float4*>(a.out_boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        a.out_scores[o] = 0.f;
        a.out_cat[o] = -1;
        if (a.payload_out) a.payload_out[o] = -1;
    }
    if (threadIdx.x == 0) a.out_count[b] = count;
}

}  // namespace

extern "C" int amp_nms(amp_ctx* ctx, int B, int cap, const float* boxes, const int* cats, const int* counts, float thresh,
                       int max_keep, unsigned long long* mask_scratch, int* keep_idx, int* keep_count) {
    AMP_REQUIRE(ctx && boxes && cats && counts && mask_scratch && keep_idx && keep_count, "amp_nms: null argument");
    AMP_REQUIRE(B >= 1 && cap >= 1 && cap <= 16384 && max_keep >= 1, "amp_nms: cap=%d must be in [1,16384]", cap);
    NmsArgs a;
    a.boxes = boxes; a.cats = cats; a.counts = counts; a.mask = mask_scratch;
    a.cap = cap; a.W = amp::cdiv(cap, 64); a.thresh = thresh;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(NMS_MASK_BLOCKS, B), dim3(256), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, ctx->stream, a, max_keep, keep_idx, keep_count);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

extern "C" size_t amp_rpn_nms_scratch_words(int B, int L, int k) {
    const size_t nchk = (size_t)amp::cdiv(k, 64);
    return (size_t)B * L * (nchk * nchk * 64 + (size_t)k + 1);
}

extern "C" int amp_rpn_nms_levels(amp_ctx* ctx, int B, int L, int k, int cap, const float* cand_boxes, const unsigned long long* cand_keys,
                                  const int* sel_count, float thresh, int max_keep, unsigned long long* scratch, float* prop_boxes,
                                  float* prop_scores, int* prop_lvl, int* prop_count, const int* payload_in, int* payload_out) {
    AMP_REQUIRE(ctx && cand_boxes && cand_keys && sel_count && scratch && prop_boxes && prop_scores && prop_lvl && prop_count,
                "amp_rpn_nms_levels: null argument");
    AMP_REQUIRE(B >= 1 && L >= 1 && L <= 8 && k >= 1 && k <= 4096 && cap >= L * k && max_keep >= 1,
                "amp_rpn_nms_levels: need 1..8 levels, k=%d in [1,4096], cap >= L*k", k);
    AMP_REQUIRE(thresh >= 0.f, "amp_rpn_nms_levels: threshold must not be negative");
    LvlArgs a;
    a.boxes = cand_boxes; a.keys = cand_keys; a.sel_count = sel_count;
    a.L = L; a.cap = cap; a.k = k; a.nchk = amp::cdiv(k, 64); a.thresh = thresh; a.max_keep = max_keep;
    const size_t mt_words = (size_t)B * L * a.nchk * a.nchk * 64;
    a.mt = scratch;
    a.kept = scratch + mt_words;
    a.kept_count = reinterpret_cast<int*>(a.kept + (size_t)B * L * k);
    hipLaunchKernelGGL(lvl_mask_kernel, dim3(a.nchk, a.nchk, B * L), dim3(64), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    const size_t smem = (size_t)LVL_SB * LVL_SB * 64 * sizeof(unsigned long long);
    static bool attr_done = false;
    if (!attr_done) {
        AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lvl_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    hipLaunchKernelGGL(lvl_scan_kernel, dim3(B * L), dim3(1024), smem, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    MergeArgs g;
    g.kept = a.kept; g.kept_count = a.kept_count; g.boxes = cand_boxes; g.payload_in = payload_in;
    g.L = L; g.cap = cap; g.k = k; g.max_keep = max_keep;
    g.out_boxes = prop_boxes; g.out_scores = prop_scores; g.out_cat = prop_lvl; g.out_count = prop_count; g.payload_out = payload_out;
    const size_t msmem = (size_t)L * k * sizeof(unsigned long long);
    AMP_REQUIRE(msmem <= 150 * 1024, "amp_rpn_nms_levels: L*k=%d survivors do not fit the merge's LDS", L * k);
    AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lvl_merge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msmem));
    hipLaunchKernelGGL(lvl_merge_kernel, dim3(B), dim3(1024), msmem, ctx->stream, g);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}
