// Backward-pass kernels that are not GEMMs: RoIAlign backward (scatter-add), the FPN top-down and p6
// gradients, the mask-predictor data gradient, ReLU masking, and the fused SGD-with-momentum update (torch.optim.SGD semantics:
// g' = g + wd*p; v = mu*v + g'; p -= lr*v).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RoiBwdArgs {
    float* dfeat[4];
    int fh[4], fw[4];
    float scale[4];
    const float* rois;
    const int* batch_idx;
    const float* dout;      // [R,P,P,C]
    int R, P, C;
};

__device__ __forceinline__ int assign_level_b(float x1, float y1, float x2, float y2) {
    const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    float lv = floorf(__fadd_rn(4.0f, log2f(__fadd_rn(__fdiv_rn(sqrtf(area), 224.0f), 1e-8f))));
    lv = fminf(fmaxf(lv, 2.0f), 5.0f);
    return (int)lv - 2;
}

// One wavefront per output bin, lanes own channels c = lane + 64*q so every atomic instruction adds 256 contiguous bytes
// (the shape the memory-side float atomics run at full rate: MI355X_MICROARCH "Global float atomics").  The sum order of
// overlapping RoIs is not fixed: gradients of the FPN features are reproducible to fp32 rounding, not bitwise.
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const RoiBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nbins = (long long)a.R * a.P * a.P;
    for (long long bin = (long long)blockIdx.x * 4 + wave; bin < nbins; bin += (long long)gridDim.x * 4) {
        const int pw = (int)(bin % a.P), ph = (int)((bin / a.P) % a.P), r = (int)(bin / (a.P * a.P));
        const float x1 = a.rois[4 * r], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level_b(x1, y1, x2, y2);
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f), sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float rw = __fsub_rn(__fsub_rn(__fmul_rn(x2, sc), 0.5f), sw), rh = __fsub_rn(__fsub_rn(__fmul_rn(y2, sc), 0.5f), sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(bh), gw = (int)ceilf(bw);
        if (gh <= 0 || gw <= 0) continue;
        const float inv = __fdiv_rn(1.0f, (float)(gh * gw));
        float* f = a.dfeat[lv] + (size_t)b * H * W * a.C;
        const float* g = a.dout + (size_t)bin * a.C;
        // The 4*gh*gw taps of a bin fall on at most (gh+1) x (gw+1) pixels and their weights are an outer product: pixel (r, c)
        // receives gv * WY[r] * WX[c] with WY[r] = sum of the hy / ly of the samples whose ylo / yhi is r (same for x).  One atomic
        // per touched pixel and channel instead of one per tap: 16 -> 9 at gh = gw = 2, 36 -> 16 at 3 (the kernel is bound by the
        // float-atomic rate).  Lane k accumulates WY of row ymin+k and WX of column xmin+k; the pixel loop broadcasts them.
        auto axis = [&](int n_samp, float start, float binsz, int pidx, int extent, int& vmin, int& vmax) -> float {
            vmin = 1 << 30; vmax = -1;
            float wsum = 0.f;
            // pass 1 (uniform): range of touched indices
            for (int i = 0; i < n_samp; ++i) {
                float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)pidx, binsz)), __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), binsz), (float)n_samp));
                if (v < -1.0f || v > (float)extent) continue;
                if (v <= 0.f) v = 0.f;
                int lo = (int)v, hi;
                if (lo >= extent - 1) { lo = hi = extent - 1; } else hi = lo + 1;
                vmin = min(vmin, lo); vmax = max(vmax, hi);
            }
            if (vmax < 0) return 0.f;
            const int mine = vmin + lane;          // this lane's row / column
            for (int i = 0; i < n_samp; ++i) {
                float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)pidx, binsz)), __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), binsz), (float)n_samp));
                if (v < -1.0f || v > (float)extent) continue;
                if (v <= 0.f) v = 0.f;
                int lo = (int)v, hi;
                if (lo >= extent - 1) { lo = hi = extent - 1; v = (float)lo; } else hi = lo + 1;
                const float l = __fsub_rn(v, (float)lo), h = __fsub_rn(1.0f, l);
                if (lo == mine) wsum = __fadd_rn(wsum, h);
                if (hi == mine) wsum = __fadd_rn(wsum, l);
            }
            return wsum;
        };
        int ymin, ymax, xmin, xmax;
        const float wy = axis(gh, sh, bh, ph, H, ymin, ymax);
        const float wx = axis(gw, sw, bw, pw, W, xmin, xmax);
        if (ymax < 0 || xmax < 0) continue;
        const int ny = ymax - ymin + 1, nx = xmax - xmin + 1;
        if (ny <= 64 && nx <= 64 && a.C <= 256) {
            float gv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[q] = (lane + 64 * q < a.C) ? __fmul_rn(g[lane + 64 * q], inv) : 0.f;
            for (int ry = 0; ry < ny; ++ry) {                    // uniform loops: every lane takes part in the shuffles
                const float wyr = __shfl(wy, ry, 64);
                if (wyr == 0.f) continue;
                float* row = f + ((size_t)(ymin + ry) * W + xmin) * a.C + lane;
                for (int rx = 0; rx < nx; ++rx) {
                    const float w = __fmul_rn(wyr, __shfl(wx, rx, 64));
                    if (w == 0.f) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (lane + 64 * q < a.C) atomicAdd(row + (size_t)rx * a.C + 64 * q, __fmul_rn(w, gv[q]));
                }
            }
            continue;
        }
        // very elongated RoI (more than 64 rows or columns under one bin) or more than 256 channels: tap by tap
        for (int c = lane; c < a.C; c += 64) {
            const float gv = __fmul_rn(g[c], inv);
            for (int iy = 0; iy < gh; ++iy) {
                float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bh), (float)gh));
                if (y < -1.0f || y > (float)H) continue;
                if (y <= 0.f) y = 0.f;
                int ylo = (int)y, yhi;
                if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else yhi = ylo + 1;
                const float ly = __fsub_rn(y, (float)ylo), hy = __fsub_rn(1.0f, ly);
                for (int ix = 0; ix < gw; ++ix) {
                    float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                    if (x < -1.0f || x > (float)W) continue;
                    if (x <= 0.f) x = 0.f;
                    int xlo = (int)x, xhi;
                    if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else xhi = xlo + 1;
                    const float lx = __fsub_rn(x, (float)xlo), hx = __fsub_rn(1.0f, lx);
                    atomicAdd(f + ((size_t)ylo * W + xlo) * a.C + c, __fmul_rn(__fmul_rn(hy, hx), gv));
                    atomicAdd(f + ((size_t)ylo * W + xhi) * a.C + c, __fmul_rn(__fmul_rn(hy, lx), gv));
                    atomicAdd(f + ((size_t)yhi * W + xlo) * a.C + c, __fmul_rn(__fmul_rn(ly, hx), gv));
                    atomicAdd(f + ((size_t)yhi * W + xhi) * a.C + c, __fmul_rn(__fmul_rn(ly, lx), gv));
                }
            }
        }
    }
}

// ---- deterministic RoIAlign backward: owner computes ------------------------------------------------------------------------
// roi_align_bwd_kernel scatters with float atomics: 4.8 ms of a 100 ms training step at the memory-side atomic rate, and the sum
// order of overlapping RoIs -- hence the low bits of every FPN / backbone gradient -- changes from run to run.  Here every 4 x 4
// tile of a gradient map belongs to ONE wave, which walks the RoIs in index order (those of its image and level whose footprint
// meets the tile), inside a RoI the bins in (ph, pw) order, and adds  g[bin] / count * WY[row] * WX[col]  (the separable tap weights
// of roi_align_bwd_kernel) into 16 x 4 accumulators per lane -- registers, statically indexed, lane = 4 consecutive channels --
// and finally adds them to the map with plain loads and stores.  No atomics, a fixed order per cell: bitwise reproducible, and
// faster (the atomics were the bound).  Pass 1 (roi_bwd_meta_kernel) computes each RoI's level, image and footprint once.
struct RoiKey { int b_lv; unsigned int ybox, xbox; };          // image << 2 | level; y0 | y1 << 16 (y1 < y0: contributes nothing); x likewise
struct RoiPar { int gh, gw; float sw, sh, bw, bh, inv; int pad; };
struct RoiBwdTileArgs {
    float* dfeat[4];
    int fh[4], fw[4], tiles_x[4], tile_off[5];   // tile_off[l]: first flat tile index of level l (B * tiles_y * tiles_x each)
    float scale[4];
    const float* rois;
    const int* batch_idx;
    const float* dout;
    RoiKey* key;
    RoiPar* par;
    int* range;                                  // [B] first RoI of image b, then [B] one past its last (a tile scans only its image's RoIs)
    int R, P, B;
    int init;                 // 1: the maps hold garbage: every tile is written (zeros where no RoI reaches), nothing is read back
};

__global__ void roi_bwd_meta_kernel(const RoiBwdTileArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    const float x1 = a.rois[4 * r], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
    const int lv = assign_level_b(x1, y1, x2, y2);
    const int b = a.batch_idx ? a.batch_idx[r] : 0;
    const int H = a.fh[lv], W = a.fw[lv];
    const float sc = a.scale[lv];
    RoiPar m;
    m.sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f); m.sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
    const float rw = __fsub_rn(__fsub_rn(__fmul_rn(x2, sc), 0.5f), m.sw), rh = __fsub_rn(__fsub_rn(__fmul_rn(y2, sc), 0.5f), m.sh);
    m.bh = __fdiv_rn(rh, (float)a.P); m.bw = __fdiv_rn(rw, (float)a.P);
    m.gh = (int)ceilf(m.bh); m.gw = (int)ceilf(m.bw);
    m.inv = (m.gh > 0 && m.gw > 0) ? __fdiv_rn(1.0f, (float)(m.gh * m.gw)) : 0.f;
    m.pad = 0;
    // rows / columns any tap can touch (samples lie inside [s, s + r]; a tap is floor(v) or floor(v) + 1, clamped to the map)
    int y0 = max(0, (int)floorf(m.sh)), yl = min(H - 1, (int)floorf(__fadd_rn(m.sh, rh)) + 1);
    int x0 = max(0, (int)floorf(m.sw)), xl = min(W - 1, (int)floorf(__fadd_rn(m.sw, rw)) + 1);
    if (!(m.gh > 0 && m.gw > 0) || !(rh > 0.f) || !(rw > 0.f) || yl < y0 || xl < x0 || b < 0 || b >= a.B) { y0 = 1; yl = 0; x0 = 1; xl = 0; }
    RoiKey k;
    k.b_lv = (b << 2) | lv;
    k.ybox = (unsigned int)y0 | ((unsigned int)yl << 16);
    k.xbox = (unsigned int)x0 | ((unsigned int)xl << 16);
    a.key[r] = k;
    a.par[r] = m;
    if (b >= 0 && b < a.B) { atomicMin(&a.range[b], r); atomicMax(&a.range[a.B + b], r + 1); }
}

// separable tap weight of one axis: sum over the bin's samples of (1 - frac) where the sample's lower tap is `mine` plus frac where
// its upper tap is (the same loop as roi_align_bwd_kernel's `axis`)
__device__ __forceinline__ float roi_axis_weight(int n_samp, float start, float binsz, int pidx, int extent, int mine) {
    float wsum = 0.f;
    for (int i = 0; i < n_samp; ++i) {
        float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)pidx, binsz)), __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), binsz), (float)n_samp));
        if (v < -1.0f || v > (float)extent) continue;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= extent - 1) { lo = hi = extent - 1; v = (float)lo; } else hi = lo + 1;
        const float l = __fsub_rn(v, (float)lo), h = __fsub_rn(1.0f, l);
        if (lo == mine) wsum = __fadd_rn(wsum, h);
        if (hi == mine) wsum = __fadd_rn(wsum, l);
    }
    return wsum;
}

typedef float f32x2p __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void roi_align_bwd_tile_kernel(const RoiBwdTileArgs a) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= a.tile_off[4]) return;
    int lv = 0;
    while (lv < 3 && tile >= a.tile_off[lv + 1]) ++lv;
    const int H = a.fh[lv], W = a.fw[lv];
    const int tx_n = a.tiles_x[lv], ty_n = (H + 3) >> 2;
    int t = tile - a.tile_off[lv];
    const int b = t / (ty_n * tx_n);
    t -= b * ty_n * tx_n;
    const int ty0 = (t / tx_n) * 4, tx0 = (t % tx_n) * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool any = false;
    const int l3 = lane & 3;
    const int want = (b << 2) | lv;
    const int r_end = a.range[a.B + b];
    for (int r0 = a.range[b]; r0 < r_end; r0 += 64) {          // the RoIs of this tile's image (a.range), 64 per trip
        const int r = r0 + lane;
        bool hit = false;
        if (r < r_end) {
            const RoiKey k = a.key[r];
            const int y0 = (int)(k.ybox & 0xffffu), yl = (int)(k.ybox >> 16), x0 = (int)(k.xbox & 0xffffu), xl = (int)(k.xbox >> 16);
            hit = k.b_lv == want && y0 <= ty0 + 3 && yl >= ty0 && x0 <= tx0 + 3 && xl >= tx0;
        }
        unsigned long long pend = __ballot(hit);
        while (pend) {                                             // ascending RoI index: the order of every cell's sum
            const int rr = r0 + __builtin_ctzll(pend);
            pend &= pend - 1;
            const RoiPar pr = a.par[rr];                           // same address in every lane
            const int gh = pr.gh, gw = pr.gw;
            const float sh = pr.sh, sw = pr.sw, bh = pr.bh, bw = pr.bw, inv = pr.inv;
            // bins whose taps can reach the tile (conservative by one bin on each side; a bin that does not gets zero weights)
            const int ph_lo = max(0, (int)floorf(__fdiv_rn((float)(ty0 - 1) - sh, bh)) - 1), ph_hi = min(a.P - 1, (int)floorf(__fdiv_rn((float)(ty0 + 4) - sh, bh)) + 1);
            const int pw_lo = max(0, (int)floorf(__fdiv_rn((float)(tx0 - 1) - sw, bw)) - 1), pw_hi = min(a.P - 1, (int)floorf(__fdiv_rn((float)(tx0 + 4) - sw, bw)) + 1);
            // the separable weights of every (bin, tile row) and (bin, tile column) pair at once: lane 4 q + k holds bin lo + q against
            // row / column k (P <= 16 bins per axis), instead of one loop over the samples per bin pair
            const int q = lane >> 2;
            const float wy_all = (ph_lo + q <= ph_hi) ? roi_axis_weight(gh, sh, bh, ph_lo + q, H, ty0 + l3) : 0.f;
            const float wx_all = (pw_lo + q <= pw_hi) ? roi_axis_weight(gw, sw, bw, pw_lo + q, W, tx0 + l3) : 0.f;
            const unsigned long long nzy = __ballot(wy_all != 0.f), nzx = __ballot(wx_all != 0.f);
            if (nzy == 0ull || nzx == 0ull) continue;
            for (int ph = ph_lo; ph <= ph_hi; ++ph) {
                const int qy = 4 * (ph - ph_lo);
                if (((nzy >> qy) & 0xfull) == 0ull) continue;
                float wyr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) wyr[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wy_all), qy + r));
                for (int pw = pw_lo; pw <= pw_hi; ++pw) {
                    const int qx = 4 * (pw - pw_lo);
                    if (((nzx >> qx) & 0xfull) == 0ull) continue;
                    f32x4 gv = *reinterpret_cast<const f32x4*>(a.dout + ((size_t)(rr * a.P + ph) * a.P + pw) * 256 + 4 * lane);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[e] = __fmul_rn(gv[e], inv);
                    float wxc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) wxc[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wx_all), qx + c));
                    // packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two values per instruction, each product and sum rounded on its own as before)
                    const f32x2p g01 = {gv[0], gv[1]}, g23 = {gv[2], gv[3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float w = __fmul_rn(wyr[r], wxc[c]);
                            const f32x2p w2 = {w, w};
                            const f32x2p a01 = f32x2p{acc[r][c][0], acc[r][c][1]} + w2 * g01;
                            const f32x2p a23 = f32x2p{acc[r][c][2], acc[r][c][3]} + w2 * g23;
                            acc[r][c] = f32x4{a01[0], a01[1], a23[0], a23[1]};
                        }
                    }
                    any = true;
                }
            }
        }
    }
    if (!any && !a.init) return;
    float* f = a.dfeat[lv] + (size_t)b * H * W * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int y = ty0 + r, x = tx0 + c;
            if (y < H && x < W) {
                f32x4* p = reinterpret_cast<f32x4*>(f + ((size_t)y * W + x) * 256) + lane;
                if (a.init) {            // first writer of the map: 0 + acc, bit for bit what the zero-filled map would end with (-0 + 0 = +0), zeros elsewhere
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(0.f, acc[r][c][e]);
                    *p = v;
                } else {
                    f32x4 v = *p;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(v[e], acc[r][c][e]);
                    *p = v;
                }
            }
        }
}

// d_coarse[b,y,x,:] += sum of the 2x2 fine cells (backward of nearest x2 upsampling in the FPN top-down path)
template <bool INIT>      // INIT: dcoarse is written (0 + sum, bit for bit what a zero-filled map would hold), not read
__global__ void upsample2_bwd_kernel(const float* __restrict__ dfine, float* __restrict__ dcoarse, int B, int Hc, int Wc, int C4) {
    const size_t total = (size_t)B * Hc * Wc * C4;
    const f32x4* f = reinterpret_cast<const f32x4*>(dfine);
    f32x4* o = reinterpret_cast<f32x4*>(dcoarse);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int x = (int)(t % Wc); t /= Wc;
        const int y = (int)(t % Hc);
        const int b = (int)(t / Hc);
        const size_t base = ((size_t)(b * 2 * Hc + 2 * y) * (2 * Wc) + 2 * x) * C4 + c;
        const f32x4 v00 = f[base], v01 = f[base + C4], v10 = f[base + (size_t)2 * Wc * C4], v11 = f[base + (size_t)2 * Wc * C4 + C4];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (!INIT) acc = o[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __fadd_rn(acc[q], __fadd_rn(__fadd_rn(v00[q], v01[q]), __fadd_rn(v10[q], v11[q])));
        o[i] = acc;
    }
}

// d_x[b,2y,2x,:] += d_y[b,y,x,:]   (backward of p6 = p5[:, ::2, ::2])
__global__ void subsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    const f32x4* g = reinterpret_cast<const f32x4*>(dy);
    f32x4* o = reinterpret_cast<f32x4*>(dx);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int x = (int)(t % Wo); t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const size_t d = ((size_t)(b * H + 2 * y) * W + 2 * x) * C4 + c;
        f32x4 v = o[d];
        const f32x4 a = g[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(v[q], a[q]);
        o[d] = v;
    }
}

// g = g * (act > 0)
__global__ void relu_mask_kernel(float* __restrict__ g, const float* __restrict__ act, size_t n4) {
    f32x4* gg = reinterpret_cast<f32x4*>(g);
    const f32x4* aa = reinterpret_cast<const f32x4*>(act);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = gg[i];
        const f32x4 m = aa[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = m[q] > 0.f ? v[q] : 0.f;
        gg[i] = v;
    }
}

// the same with act in the split hi|lo' row format (C channels per row, C % 32 == 0): positive iff one of its halves is non-zero
typedef _Float16 rm_f16x4 __attribute__((ext_vector_type(4)));
__global__ void relu_mask_split_kernel(float* __restrict__ g, const float* __restrict__ act, size_t n4, int C4) {
    f32x4* gg = reinterpret_cast<f32x4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C4;
        const int c = (int)(i - row * C4) * 4;
        const char* mb = reinterpret_cast<const char*>(act + row * (size_t)C4 * 4) + (c >> 5) * 128 + (c & 31) * 2;
        const rm_f16x4 mh = *reinterpret_cast<const rm_f16x4*>(mb);
        const rm_f16x4 ml = *reinterpret_cast<const rm_f16x4*>(mb + 64);
        f32x4 v = gg[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ((float)mh[q] + (float)ml[q] * (1.0f / 2048.0f)) > 0.f ? v[q] : 0.f;
        gg[i] = v;
    }
}

// Scaled split gradients (the backbone's backward pass on the ring kernel, model.hip): a gradient tensor d is kept as the split rows of
// d * scale (scale = 2^16: loss gradients of 1e-9..1e-4 would sit in the f16 subnormals), so that data-gradient convolutions stage it by
// LDS-DMA like an activation and the weight-gradient kernel passes its halves through.
// out = split(scale * (act > 0 ? g : 0)): the entry of a stage's chain (g fp32 from the FPN lateral, act = the stage output, split rows)
__global__ void relu_mask_to_split_kernel(const float* __restrict__ g, const float* __restrict__ act, float* __restrict__ out, size_t n4, int C4, float scale) {
    const f32x4* gg = reinterpret_cast<const f32x4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C4;
        const int c = (int)(i - row * C4) * 4;
        const size_t off = (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2;
        const char* mb = reinterpret_cast<const char*>(act + row * (size_t)C4 * 4) + off;
        const rm_f16x4 mh = *reinterpret_cast<const rm_f16x4*>(mb);
        const rm_f16x4 ml = *reinterpret_cast<const rm_f16x4*>(mb + 64);
        const f32x4 v = gg[i];
        rm_f16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x = ((float)mh[q] + (float)ml[q] * (1.0f / 2048.0f)) > 0.f ? v[q] * scale : 0.f;
            const _Float16 h = (_Float16)x;
            hi[q] = h;
            lo[q] = (_Float16)((x - (float)h) * 2048.0f);
        }
        char* ob = reinterpret_cast<char*>(out + row * (size_t)C4 * 4) + off;
        *reinterpret_cast<rm_f16x4*>(ob) = hi;
        *reinterpret_cast<rm_f16x4*>(ob + 64) = lo;
    }
}

// d_x[b,2y,2x,:] += unscale(d_y[b,y,x,:]) with d_y in scaled split rows (the exit of a stage's chain: stride-2 first block)
__global__ void subsample2_bwd_split_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C4, int Ho, int Wo, float inv_scale) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    f32x4* o = reinterpret_cast<f32x4*>(dx);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t t = i / C4;
        const size_t row = t;
        const int x = (int)(t % Wo); t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int c = c4 * 4;
        const char* sb = reinterpret_cast<const char*>(dy + row * (size_t)C4 * 4) + (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2;
        const rm_f16x4 sh = *reinterpret_cast<const rm_f16x4*>(sb);
        const rm_f16x4 sl = *reinterpret_cast<const rm_f16x4*>(sb + 64);
        const size_t d = ((size_t)(b * H + 2 * y) * W + 2 * x) * C4 + c4;
        f32x4 v = o[d];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(v[q], ((float)sh[q] + (float)sl[q] * (1.0f / 2048.0f)) * inv_scale);
        o[d] = v;
    }
}

// dx[row][:] += unscale(dy[row][:]) with dy in scaled split rows, same resolution
__global__ void accumulate_split_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t rows, int C4, float inv_scale) {
    const size_t total = rows * C4;
    f32x4* o = reinterpret_cast<f32x4*>(dx);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const size_t row = i / C4;
        const char* sb = reinterpret_cast<const char*>(dy + row * (size_t)C4 * 4) + (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2;
        const rm_f16x4 sh = *reinterpret_cast<const rm_f16x4*>(sb);
        const rm_f16x4 sl = *reinterpret_cast<const rm_f16x4*>(sb + 64);
        f32x4 v = o[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __fadd_rn(v[q], ((float)sh[q] + (float)sl[q] * (1.0f / 2048.0f)) * inv_scale);
        o[i] = v;
    }
}

// up[b, 2y, 2x, :] = src[b, y, x, :], raw 16-byte chunks (the rest of `up` was zeroed by the caller)
__global__ void scatter2_rows_kernel(const float* __restrict__ src, float* __restrict__ up, int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    const f32x4* s = reinterpret_cast<const f32x4*>(src);
    f32x4* o = reinterpret_cast<f32x4*>(up);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t t = i / C4;
        const int x = (int)(t % Wo); t /= Wo;
        const int y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        o[((size_t)(b * H + 2 * y) * W + 2 * x) * C4 + c4] = s[i];
    }
}

// dx[p][c] = (sum_k dl[p][k] * w[k][c]) * (act[p][c] > 0)     (mask predictor 1x1 conv, K <= 8 classes padded to ld)
__global__ void small_k_dgrad_kernel(const float* __restrict__ dl, int ld, int K, const float* __restrict__ w, int C,
                                     const float* __restrict__ act, float* __restrict__ dx, size_t npix) {
    const int C4 = C >> 2;
    const size_t total = npix * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C4;
        const int c = (int)(i - p * C4) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float d = dl[p * ld + k];
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)k * C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __fadd_rn(acc[q], __fmul_rn(d, wv[q]));
        }
        if (act) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(act + p * C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = m[q] > 0.f ? acc[q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(dx + p * C + c) = acc;
    }
}

// The same product for the RPN predictor on the native trunk (K <= 16, C = 256, act in the split row format), writing what its consumers
// stage: dx * scale as split rows (the dy operand of the RPN conv's weight- and data-gradient on the ring kernels) and, on the side, the
// column sums of dx (that conv's bias gradient) per 512-row slice -- one bandwidth-bound pass instead of an fp32 MFMA launch (0.87 ms at
// p2), a mask pass and a conversion.  A thread owns 8 channels for all its rows: its 16 x 8 weights stay in registers, a row costs four
// 16-B loads of dl (the same address for the 32 threads of the row), the mask halves, 120 FMAs and two 16-B stores.
// 32 column groups x 8 row lanes per workgroup; partial [gridDim.x][C] for colsum_final.
// ACT_SPLIT = false, LD4 = ld / 4 (the mask predictor behind the deconv: dl rows of Kp = 4 .. 16 floats, act = the deconv's fp32 output): the
// same sums in the same order as small_k_dgrad_kernel, so dx is that kernel's value, scaled and split.
typedef _Float16 sk_h8 __attribute__((ext_vector_type(8)));
template <bool ACT_SPLIT, int LD4>
__global__ __launch_bounds__(256) void small_k_dgrad_split_kernel(const float* __restrict__ dl, int K, const float* __restrict__ w, int C,
                                                                   const float* __restrict__ act_split, float* __restrict__ dx_split, int npix,
                                                                   float* __restrict__ partial, float scale) {
    __shared__ float red[8][32][8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int r0 = blockIdx.x * 512, r1 = min(npix, r0 + 512);
    for (int cbase = 0; cbase < C; cbase += 256) {
        const int ch = cbase + 8 * cg;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ch < C) {
            f32x4 wr[4 * LD4][2];
#pragma unroll
            for (int k = 0; k < 4 * LD4; ++k) {
                wr[k][0] = (k < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)k * C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                wr[k][1] = (k < K) ? *reinterpret_cast<const f32x4*>(w + (size_t)k * C + ch + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const size_t col_b = (size_t)(ch >> 5) * 128 + (size_t)(ch & 31) * 2;
            for (int r = r0 + rl; r < r1; r += 8) {
                f32x4 d4[LD4];
#pragma unroll
                for (int q = 0; q < LD4; ++q) d4[q] = *reinterpret_cast<const f32x4*>(dl + (size_t)r * (4 * LD4) + 4 * q);
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4 * LD4; ++k) {
                    const float d = (ACT_SPLIT || k < K) ? d4[k >> 2][k & 3] : 0.f;      // (the mask logits' pad columns are not written)
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = __fadd_rn(v[q], __fmul_rn(d, q < 4 ? wr[k][0][q] : wr[k][1][q - 4]));
                }
                float av[8];
                if (ACT_SPLIT) {
                    const char* mb = reinterpret_cast<const char*>(act_split + (size_t)r * C) + col_b;
                    const sk_h8 mh = *reinterpret_cast<const sk_h8*>(mb), ml = *reinterpret_cast<const sk_h8*>(mb + 64);
#pragma unroll
                    for (int q = 0; q < 8; ++q) av[q] = (float)mh[q] + (float)ml[q] * (1.0f / 2048.0f);
                } else {
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(act_split + (size_t)r * C + ch), m1 = *reinterpret_cast<const f32x4*>(act_split + (size_t)r * C + ch + 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) av[q] = q < 4 ? m0[q] : m1[q - 4];
                }
                sk_h8 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float g = av[q] > 0.f ? v[q] : 0.f;
                    acc[q] = __fadd_rn(acc[q], g);
                    const float x = g * scale;
                    const _Float16 h = (_Float16)x;
                    hi[q] = h;
                    lo[q] = (_Float16)((x - (float)h) * 2048.0f);
                }
                char* ob = reinterpret_cast<char*>(dx_split + (size_t)r * C) + col_b;
                *reinterpret_cast<sk_h8*>(ob) = hi;
                *reinterpret_cast<sk_h8*>(ob + 64) = lo;
            }
        }
        AMP_SYNCTHREADS();      // (builtins: under AMP_NO_PK the header's inline __syncthreads() is a call)
#pragma unroll
        for (int q = 0; q < 8; ++q) red[rl][cg][q] = acc[q];
        AMP_SYNCTHREADS();
        if (rl == 0 && ch < C) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = red[0][cg][q];
                for (int k = 1; k < 8; ++k) t = __fadd_rn(t, red[k][cg][q]);
                partial[(size_t)blockIdx.x * C + ch + q] = t;
            }
        }
    }
}

// out[(t*C2 + co)*Cin + ci] (+)= in[(ci*T + t)*C2 + co]: gradient of the ConvTranspose weight from its [ci][tap][co] wgrad form
__global__ void deconv_grad_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int Cin, int T, int C2, int accumulate) {
    const size_t total = (size_t)Cin * T * C2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const size_t r = i / Cin;            // r = t*C2 + co
        const int co = (int)(r % C2), t = (int)(r / C2);
        const float v = in[((size_t)ci * T + t) * C2 + co];
        out[i] = accumulate ? __fadd_rn(out[i], v) : v;
    }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v, size_t n, float lr, float mu, float wd,
                           float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = __fadd_rn(__fmul_rn(g[i], gscale), __fmul_rn(wd, p[i]));
        const float vi = __fadd_rn(__fmul_rn(mu, v[i]), gi);
        v[i] = vi;
        p[i] = __fsub_rn(p[i], __fmul_rn(lr, vi));
    }
}

// The same update over a table of chunks of one parameter arena (offset and length in floats, offset % 4 == 0): every trainable tensor
// of a model in ONE launch instead of one launch per tensor (80 launches whose host-side issue left the GPU idle at the end of a step).
__global__ __launch_bounds__(256) void sgd_chunks_kernel(const unsigned long long* __restrict__ chunks, float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ v, float lr, float mu, float wd, float gscale) {
    const unsigned long long ch = chunks[blockIdx.x];
    const size_t off = (size_t)(ch & 0xffffffffull);
    const int n = (int)(ch >> 32), n4 = n >> 2;
    f32x4* p4 = reinterpret_cast<f32x4*>(p + off);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + off);
    f32x4* v4 = reinterpret_cast<f32x4*>(v + off);
    for (int i = threadIdx.x; i < n4; i += 256) {
        f32x4 pi = p4[i], vi = v4[i];
        const f32x4 gi = g4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = __fadd_rn(__fmul_rn(gi[e], gscale), __fmul_rn(wd, pi[e]));
            vi[e] = __fadd_rn(__fmul_rn(mu, vi[e]), ge);
            pi[e] = __fsub_rn(pi[e], __fmul_rn(lr, vi[e]));
        }
        v4[i] = vi; p4[i] = pi;
    }
    const int i = 4 * n4 + (int)threadIdx.x;
    if (i < n) {
        const float ge = __fadd_rn(__fmul_rn(g[off + i], gscale), __fmul_rn(wd, p[off + i]));
        const float ve = __fadd_rn(__fmul_rn(mu, v[off + i]), ge);
        v[off + i] = ve;
        p[off + i] = __fsub_rn(p[off + i], __fmul_rn(lr, ve));
    }
}

inline unsigned grid_for(size_t total) { return (unsigned)std::min<size_t>(std::max<size_t>((total + 255) / 256, 1), 4096); }

}  // namespace

int amp::sgd_chunks_run(amp_ctx* ctx, const unsigned long long* chunks_dev, int nchunks, float* p, const float* g, float* v, float lr,
                        float momentum, float weight_decay, float grad_scale) {
    if (nchunks <= 0) return AMP_OK;
    hipLaunchKernelGGL(sgd_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream, chunks_dev, p, g, v, lr, momentum, weight_decay, grad_scale);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

static int g_roi_bwd_atomics = 0;     // tests: 1 = the float-atomic kernel
extern "C" void amp_debug_set_roi_bwd_atomics(int v) { g_roi_bwd_atomics = v; }

extern "C" {

int amp_roi_align_bwd(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                      const int* batch_idx, int R, int P, const float* dout) {
    AMP_REQUIRE(ctx && dfeat && fh && fw && stride && rois && dout, "amp_roi_align_bwd: null argument");
    if (R == 0) return AMP_OK;
    return amp_roi_align_bwd_batched(ctx, dfeat, fh, fw, stride, C, rois, batch_idx, R, P, dout, 0);     // B = 0: derived from batch_idx
}

int amp_roi_align_bwd_batched(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                              const int* batch_idx, int R, int P, const float* dout, int B) {
    return amp::roi_align_bwd_run(ctx, dfeat, fh, fw, stride, C, rois, batch_idx, R, P, dout, B, 0);
}

}  // extern "C"

// init = 1: the gradient maps are uninitialised and this call is their first writer (B > 0 required): the owner-computes kernel visits
// every tile anyway, so it writes zeros where no RoI reaches instead of a separate 1.3-GB zero fill of the maps of a B = 16 step
int amp::roi_align_bwd_run(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                           const int* batch_idx, int R, int P, const float* dout, int B, int init) {
    AMP_REQUIRE(ctx && dfeat && fh && fw && stride && (R == 0 || (rois && dout)), "amp_roi_align_bwd_batched: null argument");
    AMP_REQUIRE(!init || B > 0, "roi_align_bwd_run: init needs the number of images");
    static const bool env_atomics = getenv("AMP_ROI_BWD_ATOMICS") != nullptr;
    const bool tile_path = !(C != 256 || P > 16 || env_atomics || g_roi_bwd_atomics);
    if (init && (R == 0 || !tile_path)) {         // no kernel that visits every tile will run: zero the maps here
        for (int l = 0; l < 4; ++l) AMP_HIP_CHECK(hipMemsetAsync(dfeat[l], 0, (size_t)B * fh[l] * fw[l] * C * sizeof(float), ctx->stream));
        init = 0;
    }
    if (R == 0) return AMP_OK;
    if (!tile_path) {     // other widths / more than 16 bins per axis (the tile kernel's lane
        // layout), or the round-1 kernel for comparison: float atomics
        RoiBwdArgs a;
        for (int l = 0; l < 4; ++l) { a.dfeat[l] = dfeat[l]; a.fh[l] = fh[l]; a.fw[l] = fw[l]; a.scale[l] = 1.0f / (float)stride[l]; }
        a.rois = rois; a.batch_idx = batch_idx; a.dout = dout; a.R = R; a.P = P; a.C = C;
        const long long nbins = (long long)R * P * P;
        hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((unsigned)std::min<long long>((nbins + 3) / 4, 65536)), dim3(256), 0, ctx->stream, a);
        AMP_HIP_CHECK(hipGetLastError());
        return AMP_OK;
    }
    if (B <= 0) {        // images = 1 + the largest batch index (one small read-back; callers that know B pass it)
        B = 1;
        if (batch_idx) {
            std::vector<int> hb(R);
            AMP_HIP_CHECK(hipMemcpyAsync(hb.data(), batch_idx, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (int v : hb) B = std::max(B, v + 1);
        }
    }
    const size_t need = (size_t)R * (sizeof(RoiKey) + sizeof(RoiPar)) + (size_t)2 * B * sizeof(int) + 64;
    if (ctx->topk_bytes < need) {       // the context's scratch buffer (shared with amp_rpn_topk; stream order makes the reuse safe)
        AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->topk_scratch) AMP_HIP_CHECK(hipFree(ctx->topk_scratch));
        ctx->topk_scratch = nullptr; ctx->topk_bytes = 0;
        AMP_HIP_CHECK(hipMalloc(&ctx->topk_scratch, need));
        ctx->topk_bytes = need;
    }
    RoiBwdTileArgs a;
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        a.dfeat[l] = dfeat[l]; a.fh[l] = fh[l]; a.fw[l] = fw[l]; a.scale[l] = 1.0f / (float)stride[l];
        a.tiles_x[l] = (fw[l] + 3) / 4;
        a.tile_off[l] = off;
        off += B * ((fh[l] + 3) / 4) * a.tiles_x[l];
    }
    a.tile_off[4] = off;
    a.rois = rois; a.batch_idx = batch_idx; a.dout = dout; a.R = R; a.P = P; a.B = B; a.init = init;
    a.par = reinterpret_cast<RoiPar*>(ctx->topk_scratch);                      // 32-B entries first (alignment), then keys, then ranges
    a.key = reinterpret_cast<RoiKey*>(a.par + R);
    a.range = reinterpret_cast<int*>(a.key + R);
    AMP_HIP_CHECK(hipMemsetAsync(a.range, 0x7f, (size_t)B * sizeof(int), ctx->stream));          // starts: a huge index (atomicMin target)
    AMP_HIP_CHECK(hipMemsetAsync(a.range + B, 0, (size_t)B * sizeof(int), ctx->stream));         // ends: 0 (atomicMax target)
    hipLaunchKernelGGL(roi_bwd_meta_kernel, dim3(amp::cdiv(R, 256)), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(roi_align_bwd_tile_kernel, dim3(amp::cdiv(off, 4)), dim3(256), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

// init = 1: dcoarse = the 2 x 2 sums (no zero fill in front, no read-back); 0: dcoarse += them
int amp::upsample2_bwd_run(amp_ctx* ctx, const float* dfine, float* dcoarse, int B, int Hc, int Wc, int C, int init) {
    AMP_REQUIRE(ctx && dfine && dcoarse && C % 4 == 0, "amp_upsample2_bwd: bad argument");
    const size_t total = (size_t)B * Hc * Wc * (C / 4);
    if (init) hipLaunchKernelGGL(upsample2_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, dfine, dcoarse, B, Hc, Wc, C / 4);
    else hipLaunchKernelGGL(upsample2_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, dfine, dcoarse, B, Hc, Wc, C / 4);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

extern "C" {

int amp_upsample2_bwd(amp_ctx* ctx, const float* dfine, float* dcoarse, int B, int Hc, int Wc, int C) {
    return amp::upsample2_bwd_run(ctx, dfine, dcoarse, B, Hc, Wc, C, 0);
}

int amp_subsample2_bwd(amp_ctx* ctx, const float* dy, float* dx, int B, int H, int W, int C) {
    AMP_REQUIRE(ctx && dy && dx && C % 4 == 0, "amp_subsample2_bwd: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 4))), dim3(256), 0, ctx->stream, dy, dx, B, H, W, C / 4, Ho, Wo);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_relu_mask(amp_ctx* ctx, float* g, const float* act, size_t n) {
    AMP_REQUIRE(ctx && g && act && n % 4 == 0, "amp_relu_mask: bad argument");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ctx->stream, g, act, n / 4);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_relu_mask_split(amp_ctx* ctx, float* g, const float* act_split, size_t n, int C) {
    AMP_REQUIRE(ctx && g && act_split && C > 0 && C % 32 == 0 && n % (size_t)C == 0, "amp_relu_mask_split: bad argument (C %% 32 != 0?)");
    hipLaunchKernelGGL(relu_mask_split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ctx->stream, g, act_split, n / 4, C / 4);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_relu_mask_to_split(amp_ctx* ctx, const float* g, const float* act_split, float* out_split, size_t n, int C, int shift) {
    AMP_REQUIRE(ctx && g && act_split && out_split && C > 0 && C % 32 == 0 && n % (size_t)C == 0 && shift >= 0 && shift <= 24, "amp_relu_mask_to_split: bad argument");
    hipLaunchKernelGGL(relu_mask_to_split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ctx->stream, g, act_split, out_split, n / 4, C / 4, ldexpf(1.0f, shift));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_subsample2_bwd_split(amp_ctx* ctx, const float* dy_split, float* dx, int B, int H, int W, int C, int shift) {
    AMP_REQUIRE(ctx && dy_split && dx && C % 32 == 0 && shift >= 0 && shift <= 24, "amp_subsample2_bwd_split: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(subsample2_bwd_split_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 4))), dim3(256), 0, ctx->stream, dy_split, dx, B, H, W, C / 4, Ho, Wo,
                       ldexpf(1.0f, -shift));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_accumulate_split(amp_ctx* ctx, const float* dy_split, float* dx, long long rows, int C, int shift) {
    AMP_REQUIRE(ctx && dy_split && dx && rows > 0 && C % 32 == 0 && shift >= 0 && shift <= 24, "amp_accumulate_split: bad argument");
    hipLaunchKernelGGL(accumulate_split_kernel, dim3(grid_for((size_t)rows * (C / 4))), dim3(256), 0, ctx->stream, dy_split, dx, (size_t)rows, C / 4, ldexpf(1.0f, -shift));
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_scatter2_rows(amp_ctx* ctx, const float* src, float* up, int B, int H, int W, int C) {
    AMP_REQUIRE(ctx && src && up && B > 0 && H > 0 && W > 0 && C % 4 == 0, "amp_scatter2_rows: bad argument");
    AMP_HIP_CHECK(hipMemsetAsync(up, 0, (size_t)B * H * W * C * 4, ctx->stream));
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(scatter2_rows_kernel, dim3(grid_for((size_t)B * Ho * Wo * (C / 4))), dim3(256), 0, ctx->stream, src, up, B, H, W, C / 4, Ho, Wo);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_small_k_dgrad(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act, float* dx, size_t npix) {
    AMP_REQUIRE(ctx && dl && w && dx && C % 4 == 0 && K >= 1 && K <= ld, "amp_small_k_dgrad: bad argument");
    hipLaunchKernelGGL(small_k_dgrad_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, ctx->stream, dl, ld, K, w, C, act, dx, npix);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_colsum_finish(amp_ctx* ctx, const float* partial, int parts, int N, float* out, int accumulate);      // wgrad.hip

int amp_small_k_dgrad_split(amp_ctx* ctx, const float* dl, int K, const float* w, int C, const float* act_split, float* dx_split, int npix,
                            int shift, float* scratch, float* colsum_out, int accumulate) {
    AMP_REQUIRE(ctx && dl && w && act_split && dx_split && scratch && colsum_out && C % 32 == 0 && K >= 1 && K <= 16 && npix > 0 && shift >= 0 && shift <= 24,
                "amp_small_k_dgrad_split: bad argument (dl rows are 16 floats, K <= 16)");
    const int parts = (npix + 511) / 512;
    hipLaunchKernelGGL((small_k_dgrad_split_kernel<true, 4>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act_split, dx_split, npix, scratch, ldexpf(1.0f, shift));
    AMP_HIP_CHECK(hipGetLastError());
    return amp_colsum_finish(ctx, scratch, parts, C, colsum_out, accumulate);
}

// act in the split row format and dl rows of ld = 4, 8, 12 or 16 floats (the mask predictor behind a deconv whose output is kept split, round 4)
int amp_small_k_dgrad_split_ld(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act_split, float* dx_split, int npix,
                               int shift, float* scratch, float* colsum_out, int accumulate) {
    AMP_REQUIRE(ctx && dl && w && act_split && dx_split && scratch && colsum_out && C % 32 == 0 && K >= 1 && K <= ld && ld % 4 == 0 && ld <= 16 && npix > 0 && shift >= 0 && shift <= 24,
                "amp_small_k_dgrad_split_ld: bad argument (dl rows are ld = 4, 8, 12 or 16 floats, K <= ld)");
    const int parts = (npix + 511) / 512;
    const float sc = ldexpf(1.0f, shift);
    switch (ld / 4) {
        case 1: hipLaunchKernelGGL((small_k_dgrad_split_kernel<true, 1>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act_split, dx_split, npix, scratch, sc); break;
        case 2: hipLaunchKernelGGL((small_k_dgrad_split_kernel<true, 2>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act_split, dx_split, npix, scratch, sc); break;
        case 3: hipLaunchKernelGGL((small_k_dgrad_split_kernel<true, 3>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act_split, dx_split, npix, scratch, sc); break;
        default: hipLaunchKernelGGL((small_k_dgrad_split_kernel<true, 4>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act_split, dx_split, npix, scratch, sc); break;
    }
    AMP_HIP_CHECK(hipGetLastError());
    return amp_colsum_finish(ctx, scratch, parts, C, colsum_out, accumulate);
}

int amp_small_k_dgrad_split_f32act(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act, float* dx_split, int npix,
                                   int shift, float* scratch, float* colsum_out, int accumulate) {
    AMP_REQUIRE(ctx && dl && w && act && dx_split && scratch && colsum_out && C % 32 == 0 && K >= 1 && K <= ld && ld % 4 == 0 && ld <= 16 && npix > 0 && shift >= 0 && shift <= 24,
                "amp_small_k_dgrad_split_f32act: bad argument (dl rows are ld = 4, 8, 12 or 16 floats, K <= ld)");
    const int parts = (npix + 511) / 512;
    const float sc = ldexpf(1.0f, shift);
    switch (ld / 4) {
        case 1: hipLaunchKernelGGL((small_k_dgrad_split_kernel<false, 1>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act, dx_split, npix, scratch, sc); break;
        case 2: hipLaunchKernelGGL((small_k_dgrad_split_kernel<false, 2>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act, dx_split, npix, scratch, sc); break;
        case 3: hipLaunchKernelGGL((small_k_dgrad_split_kernel<false, 3>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act, dx_split, npix, scratch, sc); break;
        default: hipLaunchKernelGGL((small_k_dgrad_split_kernel<false, 4>), dim3(parts), dim3(256), 0, ctx->stream, dl, K, w, C, act, dx_split, npix, scratch, sc); break;
    }
    AMP_HIP_CHECK(hipGetLastError());
    return amp_colsum_finish(ctx, scratch, parts, C, colsum_out, accumulate);
}

int amp_deconv_grad_transpose(amp_ctx* ctx, const float* in, float* out, int Cin, int T, int C2, int accumulate) {
    AMP_REQUIRE(ctx && in && out, "amp_deconv_grad_transpose: null argument");
    hipLaunchKernelGGL(deconv_grad_transpose_kernel, dim3(grid_for((size_t)Cin * T * C2)), dim3(256), 0, ctx->stream, in, out, Cin, T, C2, accumulate);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_sgd_update(amp_ctx* ctx, float* p, const float* g, float* v, size_t n, float lr, float momentum, float weight_decay, float grad_scale) {
    AMP_REQUIRE(ctx && p && g && v, "amp_sgd_update: null argument");
    if (n == 0) return AMP_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, p, g, v, n, lr, momentum, weight_decay, grad_scale);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"
