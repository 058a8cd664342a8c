// Mask post-processing fused down to COCO RLE counts (SURVEY.md §8a row a17 + a3; §8 f2):
//   detectron2 detector_postprocess (scale boxes to the output size, clip, drop empty) + paste_masks_in_image, CPU path
//   (_do_paste_mask(skip_empty=True): region [floor(min)-1, ceil(max)+1) clamped to the image, bilinear grid_sample with
//   align_corners=False and zero padding, >= 0.5) + pycocotools rleEncode (column-major runs, starting with a 0-run),
//   which is what ampis/data_utils.py:275 computes per mask after a D2H copy.
// The dense N x H x W bool tensor (210 MB / image at 200 detections of 1024^2) never exists: one workgroup per detection
// evaluates the pasted bit per pixel of the box region straight from the 28x28 probability tile in LDS, counts the 0/1
// transitions per image column, scans them, and writes the run lengths into a global pool.  Output: uint32 counts per
// mask (the host turns them into the compressed `counts` bytes, amp_rle_to_string).
#include "common.h"

namespace {

constexpr int MS = 28;              // mask side
constexpr int PT = 256;             // threads per workgroup

__global__ void mask_prob_kernel(const float* __restrict__ logits, const int* __restrict__ classes, int N, int K,
                                 float* __restrict__ prob) {
    const int total = N * MS * MS;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int n = t / (MS * MS);
        const int c = classes[n];
        const float x = logits[(size_t)t * K + ((c >= 0 && c < K) ? c : 0)];
        prob[t] = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x)));
    }
}

struct PasteArgs {
    const float* prob;        // [N][28][28]
    const float* det_boxes;   // [N][4] in network-input coordinates
    const int* det_batch;     // [N] image index of each detection
    const int* out_h;         // [B] output (original image) height / width
    const int* out_w;
    int N, in_h, in_w;
    const int* in_hw;     // optional device [B][2]: per-image network-input (h, w) instead of in_h / in_w
    float threshold;
    float* out_boxes;         // [N][4] rescaled + clipped
    int* valid;               // [N] 1 when the rescaled box is non-empty
    unsigned int* pool;       // RLE counts pool
    unsigned long long pool_cap;
    unsigned long long* pool_used;   // device counter
    unsigned long long* rle_off;     // [N]
    int* rle_len;             // [N]
    int* overflow;
    unsigned int* pos_pool;   // optional: transition positions go here (own counter) and `pool` holds run lengths only, densely
    unsigned long long pos_cap;
    unsigned long long* pos_used;
    int max_rows;             // capacity of the per-row LDS tables
};

// bit of the pasted mask at output pixel (x, y) given the precomputed column / row interpolation parameters
struct Axis { int i0; float w1; };   // i0 = floor(coord), w1 = coord - i0 (weight of tap i0+1)

__device__ __forceinline__ Axis axis_param(int pix, float lo, float hi) {
    // img = (pix + 0.5 - lo) / (hi - lo) * 2 - 1 ; unnormalize (align_corners=False): ((img + 1) * 28 - 1) / 2
    const float g = __fsub_rn(__fmul_rn(__fdiv_rn(__fsub_rn(__fadd_rn((float)pix, 0.5f), lo), __fsub_rn(hi, lo)), 2.0f), 1.0f);
    const float c = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(g, 1.0f), (float)MS), 1.0f), 2.0f);
    const float f = floorf(c);
    Axis a;
    a.i0 = (int)f;
    a.w1 = __fsub_rn(c, f);
    return a;
}

__device__ __forceinline__ float tap(const float* sp, int y, int x) {
    return ((unsigned)y < (unsigned)MS && (unsigned)x < (unsigned)MS) ? sp[y * MS + x] : 0.f;
}

// The four taps of a column stay the same while the source row does (a 28-row mask pasted into a 200-row box: ~7 output rows per
// source row), and all lanes of a wave walk the same output row: reload them only when ay.i0 changes (wave-uniform branch).
// Same products, same summation order as paste_bit.
struct ColTaps {
    int i0;
    float t00, t01, t10, t11;
};
__device__ __forceinline__ int paste_bit_cached(const float* sp, Axis ax, Axis ay, float thr, ColTaps& c) {
    if (ay.i0 != c.i0) {
        c.i0 = ay.i0;
        c.t00 = tap(sp, ay.i0, ax.i0); c.t01 = tap(sp, ay.i0, ax.i0 + 1);
        c.t10 = tap(sp, ay.i0 + 1, ax.i0); c.t11 = tap(sp, ay.i0 + 1, ax.i0 + 1);
    }
    const float w = ax.w1, e = __fsub_rn(1.0f, w), n = ay.w1, s = __fsub_rn(1.0f, n);
    const float nw = __fmul_rn(s, e), ne = __fmul_rn(s, w), sw = __fmul_rn(n, e), se = __fmul_rn(n, w);
    float v = __fmul_rn(c.t00, nw);
    v = __fadd_rn(v, __fmul_rn(c.t01, ne));
    v = __fadd_rn(v, __fmul_rn(c.t10, sw));
    v = __fadd_rn(v, __fmul_rn(c.t11, se));
    return v >= thr ? 1 : 0;
}

__device__ __forceinline__ int paste_bit(const float* sp, Axis ax, Axis ay, float thr) {
    const float w = ax.w1, e = __fsub_rn(1.0f, w), n = ay.w1, s = __fsub_rn(1.0f, n);
    const float nw = __fmul_rn(s, e), ne = __fmul_rn(s, w), sw = __fmul_rn(n, e), se = __fmul_rn(n, w);
    float v = __fmul_rn(tap(sp, ay.i0, ax.i0), nw);
    v = __fadd_rn(v, __fmul_rn(tap(sp, ay.i0, ax.i0 + 1), ne));
    v = __fadd_rn(v, __fmul_rn(tap(sp, ay.i0 + 1, ax.i0), sw));
    v = __fadd_rn(v, __fmul_rn(tap(sp, ay.i0 + 1, ax.i0 + 1), se));
    return v >= thr ? 1 : 0;
}

__global__ __launch_bounds__(PT) void paste_rle_kernel(const PasteArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sp = reinterpret_cast<float*>(smem);                          // [784]
    int* s_scan = reinterpret_cast<int*>(sp + MS * MS);                  // [PT]
    int* s_misc = s_scan + PT;                                           // [8]
    int* row_i0 = s_misc + 8;                                            // [max_rows]
    float* row_w1 = reinterpret_cast<float*>(row_i0 + a.max_rows);       // [max_rows]
    int* colcnt = reinterpret_cast<int*>(row_w1 + a.max_rows);           // [max_rows] (columns <= max_rows too)

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int b = a.det_batch[n];
    const int H = a.out_h[b], W = a.out_w[b];
    const int in_h = a.in_hw ? a.in_hw[2 * b] : a.in_h, in_w = a.in_hw ? a.in_hw[2 * b + 1] : a.in_w;
    const float sx = (float)((double)W / (double)in_w), sy = (float)((double)H / (double)in_h);
    const float* db = a.det_boxes + (size_t)n * 4;
    float x0 = __fmul_rn(db[0], sx), y0 = __fmul_rn(db[1], sy), x1 = __fmul_rn(db[2], sx), y1 = __fmul_rn(db[3], sy);
    x0 = fminf(fmaxf(x0, 0.f), (float)W); x1 = fminf(fmaxf(x1, 0.f), (float)W);
    y0 = fminf(fmaxf(y0, 0.f), (float)H); y1 = fminf(fmaxf(y1, 0.f), (float)H);
    const bool nonempty = (__fsub_rn(x1, x0) > 0.f) && (__fsub_rn(y1, y0) > 0.f);
    if (tid == 0) {
        float* ob = a.out_boxes + (size_t)n * 4;
        ob[0] = x0; ob[1] = y0; ob[2] = x1; ob[3] = y1;
        a.valid[n] = nonempty ? 1 : 0;
        if (!nonempty) { a.rle_len[n] = 0; a.rle_off[n] = 0ull; }
    }
    if (!nonempty) return;
    const int x0i = max((int)floorf(x0) - 1, 0), y0i = max((int)floorf(y0) - 1, 0);
    const int x1i = min((int)ceilf(x1) + 1, W), y1i = min((int)ceilf(y1) + 1, H);
    const int nx = x1i - x0i, ny = y1i - y0i;

    for (int i = tid; i < MS * MS; i += PT) sp[i] = a.prob[(size_t)n * MS * MS + i];
    for (int i = tid; i < ny; i += PT) {
        const Axis ay = axis_param(y0i + i, y0, y1);
        row_i0[i] = ay.i0;
        row_w1[i] = ay.w1;
    }
    __syncthreads();

    const bool wrap = (y0i == 0 && y1i == H);   // columns are contiguous in the column-major linear order
    const float thr = a.threshold;

    // ---- pass 1: transitions per column ----
    for (int cx = tid; cx < nx; cx += PT) {
        const int x = x0i + cx;
        const Axis ax = axis_param(x, x0, x1);
        int prev = 0;
        if (wrap && cx > 0) {
            Axis pax = axis_param(x - 1, x0, x1), pay;
            pay.i0 = row_i0[ny - 1]; pay.w1 = row_w1[ny - 1];
            prev = paste_bit(sp, pax, pay, thr);
        }
        int cnt = 0;
        ColTaps ct; ct.i0 = -(1 << 30);
        for (int iy = 0; iy < ny; ++iy) {
            Axis ay; ay.i0 = row_i0[iy]; ay.w1 = row_w1[iy];
            const int bit = paste_bit_cached(sp, ax, ay, thr, ct);
            cnt += bit ^ prev;          // (both are 0 / 1)
            prev = bit;
        }
        // closing transition back to 0 when the next pixel in linear order lies outside the region
        if (prev == 1) {
            if (y1i < H) cnt += 1;
            else if (!(wrap && cx + 1 < nx) && x + 1 < W) cnt += 1;
        }
        colcnt[cx] = cnt;
    }
    __syncthreads();

    // ---- exclusive scan of colcnt (thread t owns a contiguous segment) ----
    const int seg = (nx + PT - 1) / PT;
    const int sbeg = min(tid * seg, nx), send = min(sbeg + seg, nx);
    int part = 0;
    for (int i = sbeg; i < send; ++i) part += colcnt[i];
    s_scan[tid] = part;
    __syncthreads();
    for (int off = 1; off < PT; off <<= 1) {
        const int v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    const int T = s_scan[PT - 1];
    int run = s_scan[tid] - part;
    for (int i = sbeg; i < send; ++i) {
        const int c = colcnt[i];
        colcnt[i] = run;
        run += c;
    }
    if (tid == 0) {
        // run lengths (T + 1 words, what the host reads back) and the positions they are made from (T words of scratch): one
        // region when there is no separate scratch pool, else two pools so that the read-back is the run lengths only
        const unsigned long long need = a.pos_pool ? (unsigned long long)T + 1ull : 2ull * (unsigned long long)T + 1ull;
        const unsigned long long off = atomicAdd(a.pool_used, need);
        unsigned long long poff = off + (unsigned long long)T + 1ull;
        int ok = 1;
        if (off + need > a.pool_cap) { ok = 0; *a.overflow = 1; }
        if (a.pos_pool) {
            poff = atomicAdd(a.pos_used, (unsigned long long)T);
            if (poff + (unsigned long long)T > a.pos_cap) { ok = 0; *a.overflow = 1; }
        }
        s_misc[0] = ok;
        s_misc[1] = (int)(off & 0xffffffffull);
        s_misc[2] = (int)(off >> 32);
        s_misc[3] = (int)(poff & 0xffffffffull);
        s_misc[4] = (int)(poff >> 32);
        a.rle_off[n] = off;
        a.rle_len[n] = ok ? T + 1 : 0;
    }
    __syncthreads();
    if (!s_misc[0]) return;
    const unsigned long long off = ((unsigned long long)(unsigned)s_misc[2] << 32) | (unsigned)s_misc[1];
    const unsigned long long poff = ((unsigned long long)(unsigned)s_misc[4] << 32) | (unsigned)s_misc[3];
    unsigned int* counts = a.pool + off;
    unsigned int* posbuf = (a.pos_pool ? a.pos_pool : a.pool) + poff;

    // ---- pass 2: transition positions (column-major linear index p = x*H + y) ----
    for (int cx = tid; cx < nx; cx += PT) {
        const int x = x0i + cx;
        const Axis ax = axis_param(x, x0, x1);
        int prev = 0;
        if (wrap && cx > 0) {
            Axis pax = axis_param(x - 1, x0, x1), pay;
            pay.i0 = row_i0[ny - 1]; pay.w1 = row_w1[ny - 1];
            prev = paste_bit(sp, pax, pay, thr);
        }
        int w = colcnt[cx];
        const unsigned int base = (unsigned int)x * (unsigned int)H + (unsigned int)y0i;
        ColTaps ct; ct.i0 = -(1 << 30);
        for (int iy = 0; iy < ny; ++iy) {
            Axis ay; ay.i0 = row_i0[iy]; ay.w1 = row_w1[iy];
            const int bit = paste_bit_cached(sp, ax, ay, thr, ct);
            if (bit != prev) posbuf[w++] = base + (unsigned int)iy;
            prev = bit;
        }
        if (prev == 1) {
            if (y1i < H) posbuf[w++] = (unsigned int)x * (unsigned int)H + (unsigned int)y1i;
            else if (!(wrap && cx + 1 < nx) && x + 1 < W) posbuf[w++] = (unsigned int)(x + 1) * (unsigned int)H;
        }
    }
    __syncthreads();   // workgroup-scope release/acquire: posbuf written above is visible to every lane below

    // ---- pass 3: positions -> run lengths ----
    const unsigned int total_px = (unsigned int)H * (unsigned int)W;
    for (int i = tid; i <= T; i += PT) {
        const unsigned int lo = (i == 0) ? 0u : posbuf[i - 1];
        const unsigned int hi = (i == T) ? total_px : posbuf[i];
        counts[i] = hi - lo;
    }
}


// paste_rle_kernel with the rows of a column cut into segments.  There a thread owns a whole image column of the box region: a 100 x 800
// box keeps 100 of the 256 threads busy for 800 dependent trips, twice, and the kernel lasts as long as its tallest box (the bench's
// detections: median 75 x 85 px, 5 % taller than 485, the tallest 811: 0.31 ms for 39 Mpx).  Here the unit of work is (column, segment of SEG
// rows), PT2 threads take units in segment-major order (a wave = 64 neighbouring columns on the same rows: its row parameters and the
// reload of the taps stay wave-uniform), a unit starts from the bit of the row above it (one extra evaluation) and the transition counts
// are scanned in (column, segment) order -- the column-major order of the run lengths.  Same bits, same counts.
constexpr int PT2 = 512;
constexpr int MAX_UNITS = 4096;

__global__ __launch_bounds__(PT2) void paste_rle_seg_kernel(const PasteArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sp = reinterpret_cast<float*>(smem);                          // [784]
    int* s_scan = reinterpret_cast<int*>(sp + MS * MS);                  // [PT2]
    int* s_misc = s_scan + PT2;                                          // [8]
    int* row_i0 = s_misc + 8;                                            // [max_rows]
    float* row_w1 = reinterpret_cast<float*>(row_i0 + a.max_rows);       // [max_rows]
    int* ucnt = reinterpret_cast<int*>(row_w1 + a.max_rows);             // [max(MAX_UNITS, max_rows)]: transitions per unit, index cx * nseg + seg
    // the bits of a unit (segments of <= 64 rows): pass 2 walks the transitions of the stored word instead of evaluating every pixel again
    unsigned long long* ubits = reinterpret_cast<unsigned long long*>(ucnt + ((max(MAX_UNITS, a.max_rows) + 1) & ~1));   // [MAX_UNITS]

    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int b = a.det_batch[n];
    const int H = a.out_h[b], W = a.out_w[b];
    const int in_h = a.in_hw ? a.in_hw[2 * b] : a.in_h, in_w = a.in_hw ? a.in_hw[2 * b + 1] : a.in_w;
    const float sx = (float)((double)W / (double)in_w), sy = (float)((double)H / (double)in_h);
    const float* db = a.det_boxes + (size_t)n * 4;
    float x0 = __fmul_rn(db[0], sx), y0 = __fmul_rn(db[1], sy), x1 = __fmul_rn(db[2], sx), y1 = __fmul_rn(db[3], sy);
    x0 = fminf(fmaxf(x0, 0.f), (float)W); x1 = fminf(fmaxf(x1, 0.f), (float)W);
    y0 = fminf(fmaxf(y0, 0.f), (float)H); y1 = fminf(fmaxf(y1, 0.f), (float)H);
    const bool nonempty = (__fsub_rn(x1, x0) > 0.f) && (__fsub_rn(y1, y0) > 0.f);
    if (tid == 0) {
        float* ob = a.out_boxes + (size_t)n * 4;
        ob[0] = x0; ob[1] = y0; ob[2] = x1; ob[3] = y1;
        a.valid[n] = nonempty ? 1 : 0;
        if (!nonempty) { a.rle_len[n] = 0; a.rle_off[n] = 0ull; }
    }
    if (!nonempty) return;
    const int x0i = max((int)floorf(x0) - 1, 0), y0i = max((int)floorf(y0) - 1, 0);
    const int x1i = min((int)ceilf(x1) + 1, W), y1i = min((int)ceilf(y1) + 1, H);
    const int nx = x1i - x0i, ny = y1i - y0i;
    // segments: 32 rows each unless that makes more than MAX_UNITS units (nx <= max_rows <= MAX_UNITS * ... : one segment always fits)
    int nseg = (ny + 31) / 32;
    if (nseg * nx > MAX_UNITS) nseg = max(MAX_UNITS / nx, 1);
    const int SEG = (ny + nseg - 1) / nseg;
    nseg = (ny + SEG - 1) / SEG;
    const int nunits = nx * nseg;
    const bool keep_bits = SEG <= 64 && nunits <= MAX_UNITS;

    for (int i = tid; i < MS * MS; i += PT2) sp[i] = a.prob[(size_t)n * MS * MS + i];
    for (int i = tid; i < ny; i += PT2) {
        const Axis ay = axis_param(y0i + i, y0, y1);
        row_i0[i] = ay.i0;
        row_w1[i] = ay.w1;
    }
    __syncthreads();

    const bool wrap = (y0i == 0 && y1i == H);   // columns are contiguous in the column-major linear order
    const float thr = a.threshold;

    // the bit in front of a unit's first row, in the linear order of the runs
    auto bit_before = [&](int cx, int x, const Axis& ax, int iy0) -> int {
        if (iy0 > 0) { Axis ay; ay.i0 = row_i0[iy0 - 1]; ay.w1 = row_w1[iy0 - 1]; return paste_bit(sp, ax, ay, thr); }
        if (wrap && cx > 0) {
            Axis pax = axis_param(x - 1, x0, x1), pay;
            pay.i0 = row_i0[ny - 1]; pay.w1 = row_w1[ny - 1];
            return paste_bit(sp, pax, pay, thr);
        }
        return 0;
    };

    // ---- pass 1: transitions per unit ----
    for (int u = tid; u < nunits; u += PT2) {
        const int seg = u / nx, cx = u - seg * nx;
        const int x = x0i + cx;
        const Axis ax = axis_param(x, x0, x1);
        const int iy0 = seg * SEG, iy1 = min(iy0 + SEG, ny);
        const int first = bit_before(cx, x, ax, iy0);
        int prev = first;
        int cnt = 0;
        unsigned long long word = 0ull;
        ColTaps ct; ct.i0 = -(1 << 30);
        for (int iy = iy0; iy < iy1; ++iy) {
            Axis ay; ay.i0 = row_i0[iy]; ay.w1 = row_w1[iy];
            const int bit = paste_bit_cached(sp, ax, ay, thr, ct);
            cnt += bit ^ prev;          // (both are 0 / 1)
            word |= (unsigned long long)(bit ^ prev) << ((iy - iy0) & 63);      // bit r: a transition in front of row iy0 + r
            prev = bit;
        }
        // closing transition back to 0 when the next pixel in linear order lies outside the region
        if (iy1 == ny && prev == 1) {
            if (y1i < H) cnt += 1;
            else if (!(wrap && cx + 1 < nx) && x + 1 < W) cnt += 1;
        }
        ucnt[cx * nseg + seg] = cnt;
        if (keep_bits) ubits[u] = word;
    }
    __syncthreads();

    // ---- exclusive scan of ucnt (thread t owns a contiguous stretch; wave-level shuffles, one pass over the 8 wave totals) ----
    const int stretch = (nunits + PT2 - 1) / PT2;
    const int sbeg = min(tid * stretch, nunits), send = min(sbeg + stretch, nunits);
    int part = 0;
    for (int i = sbeg; i < send; ++i) part += ucnt[i];
    int incl = part;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_scan[wv] = incl;
    __syncthreads();
    int wbase = 0, T = 0;
#pragma unroll
    for (int w2 = 0; w2 < PT2 / 64; ++w2) {
        const int v = s_scan[w2];
        if (w2 < wv) wbase += v;
        T += v;
    }
    int run = wbase + incl - part;
    for (int i = sbeg; i < send; ++i) {
        const int c = ucnt[i];
        ucnt[i] = run;
        run += c;
    }
    if (tid == 0) {
        const unsigned long long need = a.pos_pool ? (unsigned long long)T + 1ull : 2ull * (unsigned long long)T + 1ull;
        const unsigned long long off = atomicAdd(a.pool_used, need);
        unsigned long long poff = off + (unsigned long long)T + 1ull;
        int ok = 1;
        if (off + need > a.pool_cap) { ok = 0; *a.overflow = 1; }
        if (a.pos_pool) {
            poff = atomicAdd(a.pos_used, (unsigned long long)T);
            if (poff + (unsigned long long)T > a.pos_cap) { ok = 0; *a.overflow = 1; }
        }
        s_misc[0] = ok;
        s_misc[1] = (int)(off & 0xffffffffull);
        s_misc[2] = (int)(off >> 32);
        s_misc[3] = (int)(poff & 0xffffffffull);
        s_misc[4] = (int)(poff >> 32);
        a.rle_off[n] = off;
        a.rle_len[n] = ok ? T + 1 : 0;
    }
    __syncthreads();
    if (!s_misc[0]) return;
    const unsigned long long off = ((unsigned long long)(unsigned)s_misc[2] << 32) | (unsigned)s_misc[1];
    const unsigned long long poff = ((unsigned long long)(unsigned)s_misc[4] << 32) | (unsigned)s_misc[3];
    unsigned int* counts = a.pool + off;
    unsigned int* posbuf = (a.pos_pool ? a.pos_pool : a.pool) + poff;

    // ---- pass 2: transition positions (column-major linear index p = x*H + y) ----
    for (int u = tid; u < nunits; u += PT2) {
        const int seg = u / nx, cx = u - seg * nx;
        const int x = x0i + cx;
        const Axis ax = axis_param(x, x0, x1);
        const int iy0 = seg * SEG, iy1 = min(iy0 + SEG, ny);
        int w = ucnt[cx * nseg + seg];
        const unsigned int base = (unsigned int)x * (unsigned int)H + (unsigned int)y0i;
        int prev;
        if (keep_bits) {
            // the transitions of the unit were recorded in pass 1: a few set bits instead of iy1 - iy0 evaluations
            unsigned long long word = ubits[u];
            const int ntr = __popcll(word);
            while (word) {
                const int r = __builtin_ctzll(word);
                word &= word - 1ull;
                posbuf[w++] = base + (unsigned int)(iy0 + r);
            }
            // the unit's last bit = the bit in front of it, flipped once per transition
            prev = (bit_before(cx, x, ax, iy0) + ntr) & 1;
        } else {
            prev = bit_before(cx, x, ax, iy0);
            ColTaps ct; ct.i0 = -(1 << 30);
            for (int iy = iy0; iy < iy1; ++iy) {
                Axis ay; ay.i0 = row_i0[iy]; ay.w1 = row_w1[iy];
                const int bit = paste_bit_cached(sp, ax, ay, thr, ct);
                if (bit != prev) posbuf[w++] = base + (unsigned int)iy;
                prev = bit;
            }
        }
        if (iy1 == ny && prev == 1) {
            if (y1i < H) posbuf[w++] = (unsigned int)x * (unsigned int)H + (unsigned int)y1i;
            else if (!(wrap && cx + 1 < nx) && x + 1 < W) posbuf[w++] = (unsigned int)(x + 1) * (unsigned int)H;
        }
    }
    __syncthreads();   // workgroup-scope release/acquire: posbuf written above is visible to every lane below

    // ---- pass 3: positions -> run lengths ----
    const unsigned int total_px = (unsigned int)H * (unsigned int)W;
    for (int i = tid; i <= T; i += PT2) {
        const unsigned int lo = (i == 0) ? 0u : posbuf[i - 1];
        const unsigned int hi = (i == T) ? total_px : posbuf[i];
        counts[i] = hi - lo;
    }
}

}  // namespace

extern "C" {

int amp_mask_prob(amp_ctx* ctx, const float* logits, const int* classes, int N, int K, float* prob) {
    AMP_REQUIRE(ctx && logits && classes && prob, "amp_mask_prob: null argument");
    if (N == 0) return AMP_OK;
    hipLaunchKernelGGL(mask_prob_kernel, dim3(amp::cdiv(N * MS * MS, 256)), dim3(256), 0, ctx->stream, logits, classes, N, K,
                       prob);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_paste_rle(amp_ctx* ctx, const float* prob, const float* det_boxes, const int* det_batch, int N, const int* out_h,
                  const int* out_w, int max_out_hw, int in_h, int in_w, float threshold, float* out_boxes, int* valid,
                  unsigned int* pool, unsigned long long pool_cap, unsigned long long* pool_used, unsigned long long* rle_off,
                  int* rle_len, int* overflow) {
    return amp_paste_rle_sized(ctx, prob, det_boxes, det_batch, N, out_h, out_w, max_out_hw, in_h, in_w, nullptr, threshold, out_boxes,
                               valid, pool, pool_cap, pool_used, rle_off, rle_len, overflow, nullptr, 0, nullptr);
}

int amp_paste_rle_sized(amp_ctx* ctx, const float* prob, const float* det_boxes, const int* det_batch, int N, const int* out_h,
                        const int* out_w, int max_out_hw, int in_h, int in_w, const int* in_hw, float threshold, float* out_boxes,
                        int* valid, unsigned int* pool, unsigned long long pool_cap, unsigned long long* pool_used,
                        unsigned long long* rle_off, int* rle_len, int* overflow, unsigned int* pos_scratch,
                        unsigned long long pos_cap, unsigned long long* pos_used) {
    AMP_REQUIRE(!pos_scratch || pos_used, "amp_paste_rle: a position scratch pool needs its counter");
    AMP_REQUIRE(ctx && prob && det_boxes && det_batch && out_h && out_w && out_boxes && valid && pool && pool_used && rle_off &&
                rle_len && overflow, "amp_paste_rle: null argument");
    AMP_REQUIRE(max_out_hw >= 1 && max_out_hw <= 8192, "amp_paste_rle: max_out_hw=%d out of range [1,8192]", max_out_hw);
    if (N == 0) return AMP_OK;
    PasteArgs a;
    a.prob = prob; a.det_boxes = det_boxes; a.det_batch = det_batch; a.out_h = out_h; a.out_w = out_w;
    a.N = N; a.in_h = in_h; a.in_w = in_w; a.in_hw = in_hw; a.threshold = threshold;
    a.pos_pool = pos_scratch; a.pos_cap = pos_cap; a.pos_used = pos_used;
    a.out_boxes = out_boxes; a.valid = valid; a.pool = pool; a.pool_cap = pool_cap; a.pool_used = pool_used;
    a.rle_off = rle_off; a.rle_len = rle_len; a.overflow = overflow;
    a.max_rows = max_out_hw + 2;
    static const bool v1 = getenv("AMP_PASTE_V1") != nullptr;      // EXPERIMENT switch: one thread per whole column (rounds 1-3)
    if (v1) {
        const size_t smem = (size_t)(MS * MS + PT + 8) * 4 + (size_t)a.max_rows * 12;
        AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(paste_rle_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(paste_rle_kernel, dim3(N), dim3(PT), smem, ctx->stream, a);
    } else {
        const size_t smem = (size_t)(MS * MS + PT2 + 8) * 4 + (size_t)a.max_rows * 8 + (size_t)((std::max(MAX_UNITS, a.max_rows) + 1) & ~1) * 4 + 8 +
                            (size_t)MAX_UNITS * 8;
        AMP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(paste_rle_seg_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(paste_rle_seg_kernel, dim3(N), dim3(PT2), smem, ctx->stream, a);
    }
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
