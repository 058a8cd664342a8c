// Host-side COCO RLE codec of the C ABI.  Replaces the pycocotools.mask calls AMPIS makes on the
// output of the hot path: encode (ampis/data_utils.py:275), decode/area (ampis/structures.py:465-468,568,752),
// iou (ampis/analyze.py:108,158), merge (ampis/analyze.py:315-321, ampis/applications/powder.py:82-83).
// pycocotools 2.0.4 (docker/env.yml:21) is not vendored in the reference; this follows its published format:
// column-major runs alternating 0/1 starting with a 0-run; the `counts` string stores each run as 5-bit groups, LSB first,
// char = group + 48, bit 0x20 = continuation, bit 0x10 of the last group = sign, runs i > 2 stored as a delta against
// run i-2.  Byte format pinned by the reference's five result pickles (tests/golden/rle_pickles.json).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

extern "C" {

int amp_rle_to_string(const uint32_t* cnts, int m, char* out, size_t cap, size_t* len) {
    AMP_REQUIRE((cnts || m == 0) && out && len && m >= 0, "amp_rle_to_string: bad argument");
    size_t p = 0;
    for (int i = 0; i < m; ++i) {
        long long x = (long long)cnts[i];
        if (i > 2) x -= (long long)cnts[i - 2];
        bool more = true;
        while (more) {
            int c = (int)(x & 0x1f);
            x >>= 5;   // arithmetic shift keeps the sign
            more = (c & 0x10) ? (x != -1) : (x != 0);
            if (more) c |= 0x20;
            AMP_REQUIRE(p + 1 < cap, "amp_rle_to_string: output buffer too small (cap=%zu)", cap);
            out[p++] = (char)(c + 48);
        }
    }
    out[p] = 0;
    *len = p;
    return AMP_OK;
}

/* n run-length lists out of one pool (list i = pool[off[i] .. off[i] + len[i])) -> n counts strings, back to back in `out`;
 * string i = out[str_off[i] .. str_off[i + 1]).  One call per image instead of one per mask (a ctypes round trip each). */
int amp_rle_to_strings(const uint32_t* pool, const unsigned long long* off, const int* len, int n, char* out, size_t cap,
                       size_t* str_off) {
    AMP_REQUIRE((pool || n == 0) && (off || n == 0) && (len || n == 0) && out && str_off && n >= 0, "amp_rle_to_strings: bad argument");
    size_t p = 0;
    str_off[0] = 0;
    for (int i = 0; i < n; ++i) {
        size_t l = 0;
        AMP_REQUIRE(p < cap, "amp_rle_to_strings: output buffer too small (cap=%zu)", cap);
        const int st = amp_rle_to_string(pool + off[i], len[i], out + p, cap - p, &l);
        if (st != AMP_OK) return st;
        p += l;
        str_off[i + 1] = p;
    }
    return AMP_OK;
}

int amp_rle_from_string(const char* s, size_t len, uint32_t* cnts, int cap, int* m_out) {
    AMP_REQUIRE((s || len == 0) && cnts && m_out, "amp_rle_from_string: null argument");
    int m = 0;
    size_t p = 0;
    while (p < len) {
        unsigned long long x = 0;
        int k = 0;
        bool more = true;
        while (more) {
            AMP_REQUIRE(p < len, "amp_rle_from_string: truncated counts string");
            const int c = (int)(unsigned char)s[p] - 48;
            AMP_REQUIRE(c >= 0 && c < 64, "amp_rle_from_string: byte 0x%02x at offset %zu is not a counts character", (unsigned)(unsigned char)s[p], p);
            // a 32-bit run (or its signed delta) needs at most 7 groups of 5 bits; hostile input must not shift past the word (UB)
            AMP_REQUIRE(k < 8, "amp_rle_from_string: a run of more than 8 groups at offset %zu", p);
            x |= (unsigned long long)(c & 0x1f) << (5 * k);
            more = (c & 0x20) != 0;
            ++p;
            ++k;
            if (!more && (c & 0x10)) x |= ~0ull << (5 * k);      // sign extension
        }
        long long v = (long long)x;
        if (m > 2) v += (long long)cnts[m - 2];
        AMP_REQUIRE(v >= 0 && v <= 0xffffffffll, "amp_rle_from_string: run %d decodes to %lld (not a 32-bit run length)", m, v);
        AMP_REQUIRE(m < cap, "amp_rle_from_string: more than cap=%d runs", cap);
        cnts[m++] = (uint32_t)v;
    }
    *m_out = m;
    return AMP_OK;
}

// mask: column-major (Fortran order) h*w bytes, non-zero = foreground.
int amp_rle_encode(const uint8_t* mask_colmajor, int h, int w, uint32_t* cnts, int cap, int* m_out) {
    AMP_REQUIRE(mask_colmajor && cnts && m_out && h >= 0 && w >= 0, "amp_rle_encode: bad argument");
    const size_t a = (size_t)h * w;
    int m = 0;
    uint32_t c = 0;
    uint8_t p = 0;
    for (size_t j = 0; j < a; ++j) {
        const uint8_t v = mask_colmajor[j] ? 1 : 0;
        if (v != p) {
            AMP_REQUIRE(m < cap, "amp_rle_encode: more than cap=%d runs", cap);
            cnts[m++] = c;
            c = 0;
            p = v;
        }
        ++c;
    }
    AMP_REQUIRE(m < cap, "amp_rle_encode: more than cap=%d runs", cap);
    cnts[m++] = c;
    *m_out = m;
    return AMP_OK;
}

int amp_rle_decode(const uint32_t* cnts, int m, int h, int w, uint8_t* mask_colmajor) {
    AMP_REQUIRE((cnts || m == 0) && mask_colmajor, "amp_rle_decode: null argument");
    const size_t a = (size_t)h * w;
    size_t pos = 0;
    uint8_t v = 0;
    for (int i = 0; i < m; ++i) {
        AMP_REQUIRE(pos + cnts[i] <= a, "amp_rle_decode: runs exceed h*w");
        std::fill(mask_colmajor + pos, mask_colmajor + pos + cnts[i], v);
        pos += cnts[i];
        v = !v;
    }
    AMP_REQUIRE(pos == a, "amp_rle_decode: runs sum to %zu, expected %zu", pos, a);
    return AMP_OK;
}

int amp_rle_area(const uint32_t* cnts, int m, unsigned long long* area) {
    AMP_REQUIRE((cnts || m == 0) && area, "amp_rle_area: null argument");
    unsigned long long s = 0;
    for (int i = 1; i < m; i += 2) s += cnts[i];
    *area = s;
    return AMP_OK;
}

}  // extern "C"

// Walk two run lists in lock step; fn(len, va, vb) for each maximal stretch where both values are constant.
template <class F>
static void rle_zip(const uint32_t* A, int ka, const uint32_t* B, int kb, F fn) {
    unsigned long long ca = ka ? A[0] : 0, cb = kb ? B[0] : 0;
    int a = 1, b = 1;
    bool va = false, vb = false;
    unsigned long long ct = 1;
    while (ct > 0) {
        const unsigned long long c = std::min(ca, cb);
        fn(c, va, vb);
        ct = 0;
        ca -= c;
        if (!ca && a < ka) { ca = A[a++]; va = !va; }
        ct += ca;
        cb -= c;
        if (!cb && b < kb) { cb = B[b++]; vb = !vb; }
        ct += cb;
    }
}

extern "C" {

// IoU of mask d (dt) against mask g (gt); iscrowd: union replaced by area(dt). Same values as pycocotools rleIou
// (0 when the intersection is empty).
int amp_rle_iou(const uint32_t* dt, int md, const uint32_t* gt, int mg, int iscrowd, double* iou) {
    AMP_REQUIRE(dt && gt && iou && md > 0 && mg > 0, "amp_rle_iou: bad argument");
    unsigned long long i = 0, u = 0;
    rle_zip(dt, md, gt, mg, [&](unsigned long long c, bool va, bool vb) {
        if (va || vb) {
            u += c;
            if (va && vb) i += c;
        }
    });
    if (i == 0) u = 1;
    else if (iscrowd) (void)amp_rle_area(dt, md, &u);
    *iou = (double)i / (double)u;
    return AMP_OK;
}

/* IoU of every pair out of two pools of run-length lists (list i = pool[off[i] .. off[i] + len[i])): out[d * ng + g], the matrix
 * pycocotools.mask.iou(dt, gt, iscrowd) returns.  Like rleIou it looks at the bounding boxes first (h = mask height, 0 = skip that
 * test): boxes that do not overlap mean IoU 0 without walking the runs -- for a few hundred instances per micrograph that is all but
 * a few pairs per row. */
int amp_rle_iou_matrix(const uint32_t* dpool, const unsigned long long* doff, const int* dlen, int nd, const uint32_t* gpool,
                       const unsigned long long* goff, const int* glen, int ng, const unsigned char* iscrowd, int h, double* out) {
    AMP_REQUIRE(nd >= 0 && ng >= 0 && h >= 0 && (nd == 0 || (dpool && doff && dlen)) && (ng == 0 || (gpool && goff && glen)) &&
                (out || nd == 0 || ng == 0), "amp_rle_iou_matrix: bad argument");
    struct Box { long long x0, x1, y0, y1; bool any; };
    auto bbox = [&](const uint32_t* c, int m) {
        Box b{1LL << 60, -1, 1LL << 60, -1, false};
        unsigned long long p = 0;
        for (int i = 0; i < m; ++i) {
            const unsigned long long l = c[i];
            if ((i & 1) && l > 0) {
                b.any = true;
                if (h > 0) {
                    const long long xa = (long long)(p / h), xb = (long long)((p + l - 1) / h);
                    b.x0 = std::min(b.x0, xa); b.x1 = std::max(b.x1, xb);
                    if (xa == xb) { b.y0 = std::min(b.y0, (long long)(p % h)); b.y1 = std::max(b.y1, (long long)((p + l - 1) % h)); }
                    else { b.y0 = 0; b.y1 = h - 1; }
                } else {
                    b.x0 = std::min(b.x0, (long long)p); b.x1 = std::max(b.x1, (long long)(p + l - 1));   // linear extent
                    b.y0 = 0; b.y1 = 0;
                }
            }
            p += l;
        }
        return b;
    };
    std::vector<Box> db((size_t)nd), gb((size_t)ng);
    for (int d = 0; d < nd; ++d) { AMP_REQUIRE(dlen[d] > 0, "amp_rle_iou_matrix: empty run list"); db[d] = bbox(dpool + doff[d], dlen[d]); }
    for (int g = 0; g < ng; ++g) { AMP_REQUIRE(glen[g] > 0, "amp_rle_iou_matrix: empty run list"); gb[g] = bbox(gpool + goff[g], glen[g]); }
    for (int d = 0; d < nd; ++d)
        for (int g = 0; g < ng; ++g) {
            const Box &a = db[d], &b = gb[g];
            double v = 0.0;
            if (a.any && b.any && a.x0 <= b.x1 && b.x0 <= a.x1 && a.y0 <= b.y1 && b.y0 <= a.y1) {
                const int st = amp_rle_iou(dpool + doff[d], dlen[d], gpool + goff[g], glen[g], iscrowd ? (int)iscrowd[g] : 0, &v);
                if (st != AMP_OK) return st;
            }
            out[(size_t)d * ng + g] = v;
        }
    return AMP_OK;
}

// out = A & B (intersect != 0) or A | B, both over the same h*w. Returns the number of runs in *m_out.
int amp_rle_pair_overlap(const uint32_t* apool, const unsigned long long* aoff, const int* alen, const uint32_t* bpool,
                         const unsigned long long* boff, const int* blen, const int* pair_a, const int* pair_b, int npairs,
                         unsigned long long* inter, unsigned long long* only_a, unsigned long long* only_b) {
    AMP_REQUIRE(npairs >= 0 && (npairs == 0 || (apool && aoff && alen && bpool && boff && blen && pair_a && pair_b && inter && only_a && only_b)),
                "amp_rle_pair_overlap: bad argument");
    for (int p = 0; p < npairs; ++p) {
        const int ia = pair_a[p], ib = pair_b[p];
        AMP_REQUIRE(ia >= 0 && ib >= 0 && alen[ia] > 0 && blen[ib] > 0, "amp_rle_pair_overlap: pair %d names an empty run list", p);
        unsigned long long both = 0, a = 0, b = 0;      // one pass over the two run lists: the three pixel classes of the pair
        rle_zip(apool + aoff[ia], alen[ia], bpool + boff[ib], blen[ib], [&](unsigned long long c, bool va, bool vb) {
            if (va && vb) both += c;
            else if (va) a += c;
            else if (vb) b += c;
        });
        inter[p] = both; only_a[p] = a; only_b[p] = b;
    }
    return AMP_OK;
}

int amp_rle_merge2(const uint32_t* A, int ka, const uint32_t* B, int kb, int intersect, uint32_t* out, int cap, int* m_out) {
    AMP_REQUIRE(A && B && out && m_out && ka > 0 && kb > 0, "amp_rle_merge2: bad argument");
    int m = 0;
    bool v = false;
    unsigned long long cc = 0;
    bool overflow = false;
    unsigned long long ca = A[0], cb = B[0];
    int a = 1, b = 1;
    bool va = false, vb = false;
    unsigned long long ct = 1;
    while (ct > 0) {
        const unsigned long long c = std::min(ca, cb);
        cc += c;
        ct = 0;
        ca -= c;
        if (!ca && a < ka) { ca = A[a++]; va = !va; }
        ct += ca;
        cb -= c;
        if (!cb && b < kb) { cb = B[b++]; vb = !vb; }
        ct += cb;
        const bool vp = v;
        v = intersect ? (va && vb) : (va || vb);
        if (v != vp || ct == 0) {
            if (m < cap) out[m++] = (uint32_t)cc; else overflow = true;
            cc = 0;
        }
    }
    AMP_REQUIRE(!overflow, "amp_rle_merge2: more than cap=%d runs", cap);
    *m_out = m;
    return AMP_OK;
}

}  // extern "C"

extern "C" {

// Polygon (flat x0,y0,x1,y1,... ; k vertices) -> run lengths of an h x w mask: pycocotools maskApi.c rleFrPoly, the routine behind
// RLE.frPyObjects (ampis/structures.py:677) and detectron2's polygons_to_bitmask.  Boundary is traced on a 5x upsampled grid,
// x-crossings become run boundaries, sorted, differenced, zero-length runs merged.
int amp_rle_from_polygon(const double* xy, int k, int h, int w, uint32_t* cnts, int cap, int* m_out) {
    AMP_REQUIRE(xy && cnts && m_out && k >= 1 && h > 0 && w > 0, "amp_rle_from_polygon: bad argument");
    const double scale = 5.0;
    std::vector<int> x(k + 1), y(k + 1);
    for (int j = 0; j < k; ++j) { x[j] = (int)(scale * xy[2 * j] + 0.5); y[j] = (int)(scale * xy[2 * j + 1] + 0.5); }
    x[k] = x[0]; y[k] = y[0];
    std::vector<unsigned long long> a;
    for (int j = 0; j < k; ++j) {
        int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1];
        const int dx = std::abs(xe - xs), dy = std::abs(ys - ye);
        const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
        if (flip) { std::swap(xs, xe); std::swap(ys, ye); }
        const int len = dx >= dy ? dx : dy;
        const double s = dx >= dy ? (dx ? (double)(ye - ys) / dx : 0.0) : (double)(xe - xs) / dy;
        int pu = 0, pv = 0;
        for (int d = 0; d <= len; ++d) {
            const int t = flip ? len - d : d;
            int u, v;
            if (dx >= dy) { u = t + xs; v = (int)(ys + s * t + 0.5); } else { v = t + ys; u = (int)(xs + s * t + 0.5); }
            if (d > 0 && u != pu) {   // consecutive edges share their vertex, so pairs across edges never differ
                double xd = (double)(u < pu ? u : u - 1);
                xd = (xd + 0.5) / scale - 0.5;
                if (std::floor(xd) == xd && xd >= 0 && xd <= w - 1) {
                    double yd = (double)(v < pv ? v : pv);
                    yd = (yd + 0.5) / scale - 0.5;
                    if (yd < 0) yd = 0; else if (yd > h) yd = h;
                    yd = std::ceil(yd);
                    a.push_back((unsigned long long)xd * (unsigned long long)h + (unsigned long long)yd);
                }
            }
            pu = u; pv = v;
        }
    }
    a.push_back((unsigned long long)h * (unsigned long long)w);
    std::sort(a.begin(), a.end());
    unsigned long long p = 0;
    for (auto& v : a) { const unsigned long long t = v; v -= p; p = t; }
    std::vector<unsigned long long> b;
    size_t j = 0;
    b.push_back(a[j++]);
    while (j < a.size()) {
        if (a[j] > 0) b.push_back(a[j++]);
        else { ++j; if (j < a.size()) b.back() += a[j++]; }
    }
    AMP_REQUIRE((int)b.size() <= cap, "amp_rle_from_polygon: more than cap=%d runs", cap);
    for (size_t i = 0; i < b.size(); ++i) cnts[i] = (uint32_t)b[i];
    *m_out = (int)b.size();
    return AMP_OK;
}

}  // extern "C"

// ---- nearest-neighbour resize (+ horizontal mirror) of a mask IN THE RUN-LENGTH DOMAIN -----------------------------------------------
// What detectron2 does to a bitmask annotation when the image is resized: ResizeTransform.apply_segmentation = PIL Image.resize(NEAREST)
// of the decoded mask (then HFlipTransform).  Decoding, resizing and re-encoding every instance of a micrograph costs seconds per image on
// the host (476 instances at 1024 x 1536: 2.1 s); the same result from the runs costs microseconds.  The pixel correspondence is Pillow's
// ImagingScaleAffine, restated with its double-precision ACCUMULATION (xo += a0 per step, COORD() = truncation): output column x reads source
// column xin[x], output row y reads source row yin[y]; both tables are non-decreasing, so a run boundary of a source column maps to the
// first output row whose source row reaches it.
extern "C" int amp_rle_resize_nearest(const uint32_t* cnts, int m, int h, int w, int nh, int nw, int flip, uint32_t* out, int cap, int* m_out) {
    AMP_REQUIRE(cnts && out && m_out && m > 0 && h > 0 && w > 0 && nh > 0 && nw > 0 && cap > 0, "amp_rle_resize_nearest: bad argument");
    auto table = [](int n_in, int n_out, std::vector<int>& tab) {
        tab.resize((size_t)n_out);
        const double a = (double)n_in / (double)n_out;
        double o = a * 0.5;
        for (int i = 0; i < n_out; ++i) {
            int v = o < 0.0 ? -1 : (int)o;
            tab[(size_t)i] = v < n_in ? v : n_in - 1;          // (Pillow skips coordinates beyond the image; they cannot occur for a pure scale)
            o += a;
        }
    };
    std::vector<int> xin, yin;
    table(w, nw, xin);
    table(h, nh, yin);
    // first output row that reads source row >= r, for r in [0, h]
    std::vector<int> first((size_t)h + 1);
    {
        int y = 0;
        for (int r = 0; r <= h; ++r) {
            while (y < nh && yin[(size_t)y] < r) ++y;
            first[(size_t)r] = y;
        }
    }
    // per source column: value at row 0 and the rows where the value changes (a zero-length run gives two changes at one row: they cancel)
    std::vector<int> col_start((size_t)w + 1, 0);
    std::vector<int> trans;          // transition rows, column after column
    std::vector<unsigned char> col_v0((size_t)w, 0);
    {
        unsigned long long total = 0;
        for (int j = 0; j < m; ++j) total += cnts[j];
        AMP_REQUIRE(total == (unsigned long long)h * w, "amp_rle_resize_nearest: the runs cover %llu pixels, the mask has %d x %d", total, h, w);
        int j = 0;
        unsigned long long run_end = cnts[0];
        unsigned char v = 0;
        for (int c = 0; c < w; ++c) {
            const unsigned long long top = (unsigned long long)c * h, bot = top + h;
            while (run_end <= top) { ++j; run_end += cnts[j]; v ^= 1; }
            col_v0[(size_t)c] = v;
            col_start[(size_t)c] = (int)trans.size();
            while (run_end < bot) { trans.push_back((int)(run_end - top)); ++j; run_end += cnts[j]; v ^= 1; }
        }
        col_start[(size_t)w] = (int)trans.size();
    }
    // emit the output runs column by column
    int mo = 0;
    unsigned long long run = 0;
    unsigned char cur = 0;           // COCO RLE starts with a run of zeros
    auto put = [&](unsigned char v, unsigned long long n) -> bool {
        if (n == 0) return true;
        if (v == cur) { run += n; return true; }
        if (mo >= cap) return false;
        out[mo++] = (uint32_t)run;
        cur = v; run = n;
        return true;
    };
    for (int ox = 0; ox < nw; ++ox) {
        const int sx = xin[(size_t)(flip ? nw - 1 - ox : ox)];
        unsigned char v = col_v0[(size_t)sx];
        int y0 = 0;
        for (int t = col_start[(size_t)sx]; t < col_start[(size_t)sx + 1]; ++t) {
            const int y1 = first[(size_t)trans[(size_t)t]];
            if (!put(v, (unsigned long long)(y1 - y0))) { amp::set_error("amp_rle_resize_nearest: output capacity %d too small", cap); return AMP_ERR_NOMEM; }
            y0 = y1;
            v ^= 1;
        }
        if (!put(v, (unsigned long long)(nh - y0))) { amp::set_error("amp_rle_resize_nearest: output capacity %d too small", cap); return AMP_ERR_NOMEM; }
    }
    if (mo >= cap) { amp::set_error("amp_rle_resize_nearest: output capacity %d too small", cap); return AMP_ERR_NOMEM; }
    out[mo++] = (uint32_t)run;
    *m_out = mo;
    return AMP_OK;
}
