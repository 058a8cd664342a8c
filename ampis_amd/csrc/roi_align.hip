// RoIAlign (aligned=True, sampling_ratio=0) over the FPN levels p2..p5 with the level assignment fused.
//
// Replaces detectron2 ROIPooler.forward -> torchvision.ops.roi_align (App. A.4):
//   level = clamp(floor(4 + log2(sqrt(area) / 224 + 1e-8)), 2, 5)
//   roi*scale - 0.5, bin = roi_size / P, sampling grid ceil(roi_h / P) x ceil(roi_w / P), bilinear taps with the
//   torchvision edge rules, mean over the grid.  Arithmetic order follows roi_align_kernel (fp32, no contraction):
//   acc += ((w1*v1 + w2*v2) + w3*v3) + w4*v4, iy outer / ix inner, then acc / count.
//
// Layout: features NHWC; output [R][P][P][C].  One wavefront per output bin: 64 lanes x float4 = 256 channels, so
// every tap is one fully coalesced 1 KiB read; HBM/L2-gather bound, no LDS needed.
#include "common.h"
#include "select.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct RoiArgs {
    const float* feat[4];
    int fh[4], fw[4];
    float scale[4];
    const float* rois;      // [R,4] x1,y1,x2,y2
    const int* batch_idx;   // [R] (may be null -> all 0)
    const int* roi_count;   // device int: number of valid rois (may be null -> R)
    float* out;             // [R,P,P,C]
    int* level_out;         // [R] (may be null)
    int R, P, C;
    int out_split;          // 1: write the AMP_CONV_F16X3 operand format (per 32 channels 64 B of f16 hi halves + 64 B of lo' halves)
    int in_split;           // 1: the feature maps are in that format (the trunk's native activation format in AMP_CONV_F16X3 inference)
    int share_taps;         // roi_align_split_kernel: keep a sample row's taps in registers (EXPERIMENT switch AMP_ROI_SHARE)
    // XCD-major order (roi_order_kernel): workgroup b runs on XCD b % 8 and takes its RoIs from order[(b % 8) * xstride + ...], xlen[b % 8]
    // of them -- per image the x-th eighth of the RoIs sorted by (level, Morton tile of the centre), so that the RoIs an L2 sees one after
    // the other overlap, while all eight XCDs still work on the same image (its maps stay in the Infinity Cache).  null: index order.
    const int* order;
    const int* xlen;
    int xstride;
};

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// 4 channels [4c, 4c + 4) of one feature pixel: an fp32 float4, or -- split rows -- 8 B of hi halves and 8 B of lo' halves,
// decoded exactly (hi + lo' * 2^-11)
template <bool SPLIT>
__device__ __forceinline__ f32x4 load_tap(const float* row, int c) {
    if (!SPLIT) return reinterpret_cast<const f32x4*>(row)[c];
    const int ch = 4 * c;
    const char* base = reinterpret_cast<const char*>(row) + (ch >> 5) * 128 + (ch & 31) * 2;
    const f16x4 h = *reinterpret_cast<const f16x4*>(base);
    const f16x4 l = *reinterpret_cast<const f16x4*>(base + 64);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __fadd_rn((float)h[e], __fmul_rn((float)l[e], 1.0f / 2048.0f));
    return v;
}

__device__ __forceinline__ int assign_level(float x1, float y1, float x2, float y2) {
    const float area = __fmul_rn(__fsub_rn(x2, x1), __fsub_rn(y2, y1));
    const float sz = sqrtf(area);
    float lv = floorf(__fadd_rn(4.0f, log2f(__fadd_rn(__fdiv_rn(sz, 224.0f), 1e-8f))));
    lv = fminf(fmaxf(lv, 2.0f), 5.0f);   // NaN (negative area) -> fmaxf picks 2
    return (int)lv - 2;
}

template <bool IN_SPLIT>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nvalid = a.roi_count ? min(*a.roi_count, a.R) : a.R;
    const long long nbins = (long long)nvalid * a.P * a.P;
    const int C4 = a.C >> 2;
    for (long long bin = (long long)blockIdx.x * 4 + wave; bin < nbins; bin += (long long)gridDim.x * 4) {
        const int pw = (int)(bin % a.P);
        const int ph = (int)((bin / a.P) % a.P);
        const int r = (int)(bin / (a.P * a.P));
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        if (a.level_out && ph == 0 && pw == 0 && lane == 0) a.level_out[r] = lv;
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f);
        const float sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float ew = __fsub_rn(__fmul_rn(x2, sc), 0.5f);
        const float eh = __fsub_rn(__fmul_rn(y2, sc), 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)a.P));
        const int gw = (int)ceilf(__fdiv_rn(rw, (float)a.P));
        const float count = (float)max(gh * gw, 1);
        const float* fb = a.feat[lv] + (size_t)b * H * W * a.C;
        f32x4* o4 = reinterpret_cast<f32x4*>(a.out) + (size_t)bin * C4;
        for (int c = lane; c < C4; c += 64) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int iy = 0; iy < gh; ++iy) {
                float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)),
                                    __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bh), (float)gh));
                const bool ybad = (y < -1.0f) || (y > (float)H);
                if (y <= 0.f) y = 0.f;
                int ylo = (int)y, yhi;
                if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
                const float ly = __fsub_rn(y, (float)ylo), hy = __fsub_rn(1.0f, ly);
                for (int ix = 0; ix < gw; ++ix) {
                    float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)),
                                        __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                    const bool bad = ybad || (x < -1.0f) || (x > (float)W);
                    if (bad) continue;
                    if (x <= 0.f) x = 0.f;
                    int xlo = (int)x, xhi;
                    if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
                    const float lx = __fsub_rn(x, (float)xlo), hx = __fsub_rn(1.0f, lx);
                    const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                    const f32x4 v1 = load_tap<IN_SPLIT>(fb + ((size_t)ylo * W + xlo) * a.C, c);
                    const f32x4 v2 = load_tap<IN_SPLIT>(fb + ((size_t)ylo * W + xhi) * a.C, c);
                    const f32x4 v3 = load_tap<IN_SPLIT>(fb + ((size_t)yhi * W + xlo) * a.C, c);
                    const f32x4 v4 = load_tap<IN_SPLIT>(fb + ((size_t)yhi * W + xhi) * a.C, c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1[e]), __fmul_rn(w2, v2[e])),
                                                            __fmul_rn(w3, v3[e])), __fmul_rn(w4, v4[e]));
                        acc[e] = __fadd_rn(acc[e], s);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __fdiv_rn(acc[e], count);
            if (a.out_split) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 h = (_Float16)acc[e];
                    hi[e] = h;
                    lo[e] = (_Float16)((acc[e] - (float)h) * 2048.0f);
                }
                const int ch = 4 * c;
                char* base = reinterpret_cast<char*>(o4) + (ch >> 5) * 128 + (ch & 31) * 2;
                *reinterpret_cast<f16x4*>(base) = hi;
                *reinterpret_cast<f16x4*>(base + 64) = lo;
            } else {
                o4[c] = acc;
            }
        }
    }
}


// roi_align_lanes_kernel: the same arithmetic, organised for the VALU.  roi_align_kernel is NOT memory-bound: with every RoI on the
// same few cells (all taps from L2) it runs at 80 % of its time on real proposals (tools/bench_roi.py) -- its time is the sampling
// arithmetic, ~60 VALU instructions per sample (two IEEE divisions among them) that all 64 lanes execute redundantly on wave-uniform
// values, next to 32 that do the interpolation.  Here lane j computes the parameters of sample COLUMN j and of sample ROW j of the
// wave's bin once (same operations, same order: bit-identical), and the sample loops fetch them with v_readlane into scalar
// registers: per sample the vector unit is left with 4 weight products, 4 loads and the 32 interpolation operations.
// fp32 feature maps, C <= 256 * k (float4 per lane), sampling grids up to 64 x 64 (larger ones: roi_align_kernel).
template <int DUMMY>
__global__ __launch_bounds__(256) void roi_align_lanes_kernel(const RoiArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nvalid = a.roi_count ? min(*a.roi_count, a.R) : a.R;
    const long long nbins = (long long)nvalid * a.P * a.P;
    const int C4 = a.C >> 2;
    for (long long bin = (long long)blockIdx.x * 4 + wave; bin < nbins; bin += (long long)gridDim.x * 4) {
        const int pw = (int)(bin % a.P);
        const int ph = (int)((bin / a.P) % a.P);
        const int r = (int)(bin / (a.P * a.P));
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        if (a.level_out && ph == 0 && pw == 0 && lane == 0) a.level_out[r] = lv;
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f);
        const float sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float ew = __fsub_rn(__fmul_rn(x2, sc), 0.5f);
        const float eh = __fsub_rn(__fmul_rn(y2, sc), 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)a.P));
        const int gw = (int)ceilf(__fdiv_rn(rw, (float)a.P));
        const float count = (float)max(gh * gw, 1);
        // lane j holds the parameters of sample row iy0 + j / sample column ix0 + j (grids beyond 64 go in chunks of 64; unused lanes
        // compute harmless values)
        auto row_params = [&](int iy0, int& ylo_w, int& yhi_w, float& ly, float& hy) {
            float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)(iy0 + lane), 0.5f), bh), (float)gh));
            const bool ybad = (y < -1.0f) || (y > (float)H);
            if (y <= 0.f) y = 0.f;
            int ylo = (int)y, yhi;
            if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
            ly = __fsub_rn(y, (float)ylo); hy = __fsub_rn(1.0f, ly);
            ylo_w = ybad ? -1 : ylo * W; yhi_w = yhi * W;            // row offsets in pixels; -1 marks a row outside the map
        };
        auto col_params = [&](int ix0, int& xlo, int& xhi, float& lx, float& hx) {
            float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)(ix0 + lane), 0.5f), bw), (float)gw));
            const bool xbad = (x < -1.0f) || (x > (float)W);
            if (x <= 0.f) x = 0.f;
            xlo = (int)x;
            if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
            lx = __fsub_rn(x, (float)xlo); hx = __fsub_rn(1.0f, lx);
            if (xbad) xlo = -1;                                      // -1 marks a column outside the map
        };
        int ylo_w, yhi_w, xlo, xhi;
        float ly, hy, lx, hx;
        row_params(0, ylo_w, yhi_w, ly, hy);
        col_params(0, xlo, xhi, lx, hx);

        const f32x4* f4 = reinterpret_cast<const f32x4*>(a.feat[lv]) + (size_t)b * H * W * C4;
        f32x4* o4 = reinterpret_cast<f32x4*>(a.out) + (size_t)bin * C4;
        for (int c = lane; c < C4; c += 64) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int iy = 0; iy < gh; ++iy) {
                if (gh > 64 && (iy & 63) == 0) row_params(iy, ylo_w, yhi_w, ly, hy);
                const int s_ylo = __builtin_amdgcn_readlane(ylo_w, iy & 63);
                if (s_ylo < 0) continue;
                const int s_yhi = __builtin_amdgcn_readlane(yhi_w, iy & 63);
                const float s_ly = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ly), iy & 63));
                const float s_hy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hy), iy & 63));
                for (int ix = 0; ix < gw; ++ix) {
                    if (gw > 64 && (ix & 63) == 0) col_params(ix, xlo, xhi, lx, hx);
                    const int s_xlo = __builtin_amdgcn_readlane(xlo, ix & 63);
                    if (s_xlo < 0) continue;
                    const int s_xhi = __builtin_amdgcn_readlane(xhi, ix & 63);
                    const float s_lx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lx), ix & 63));
                    const float s_hx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hx), ix & 63));
                    const float w1 = __fmul_rn(s_hy, s_hx), w2 = __fmul_rn(s_hy, s_lx), w3 = __fmul_rn(s_ly, s_hx), w4 = __fmul_rn(s_ly, s_lx);
                    const f32x4 v1 = f4[(size_t)(s_ylo + s_xlo) * C4 + c];
                    const f32x4 v2 = f4[(size_t)(s_ylo + s_xhi) * C4 + c];
                    const f32x4 v3 = f4[(size_t)(s_yhi + s_xlo) * C4 + c];
                    const f32x4 v4 = f4[(size_t)(s_yhi + s_xhi) * C4 + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1[e]), __fmul_rn(w2, v2[e])), __fmul_rn(w3, v3[e])), __fmul_rn(w4, v4[e]));
                        acc[e] = __fadd_rn(acc[e], t);
                    }
                }
                if (gw > 64) col_params(0, xlo, xhi, lx, hx);       // back to the first chunk for the next row
            }
            if (gh > 64) row_params(0, ylo_w, yhi_w, ly, hy);       // ... and for the next channel group
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __fdiv_rn(acc[e], count);
            if (a.out_split) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 h = (_Float16)acc[e];
                    hi[e] = h;
                    lo[e] = (_Float16)((acc[e] - (float)h) * 2048.0f);
                }
                const int ch = 4 * c;
                char* base = reinterpret_cast<char*>(o4) + (ch >> 5) * 128 + (ch & 31) * 2;
                *reinterpret_cast<f16x4*>(base) = hi;
                *reinterpret_cast<f16x4*>(base + 64) = lo;
            } else {
                o4[c] = acc;
            }
        }
    }
}

// roi_align_rows_kernel: the "tile-major" formulation -- one workgroup (8 waves) per BIN ROW of a RoI copies the distinct cells the
// bin row touches once into LDS (each cell read once, split rows decoded once), builds the sample-row / sample-column tables once
// (one thread per sample row / column, the operations of roi_align_kernel in the same order) and then wave w reduces bin w entirely
// from LDS.  Bit-identical to roi_align_kernel (tests/test_stages_gpu.py).  MEASURED SLOWER than the per-bin kernels on the bench's
// shapes (8000 proposals, P = 7: 1245 us against 899 us; tools/bench_roi.py) and therefore NOT the default (AMP_ROI_LANES=3 selects
// it).  Why: RoIAlign here is bound neither by HBM nor by arithmetic but by dependent memory round trips.  With every box on the
// same cells (all taps L2 hits) the per-bin kernel still needs 70-80 % of its time; a bin is a chain of ~11 round trips (box ->
// 9 samples -> store) and 28 resident waves per CU hide it up to ~19 TB/s of L2 -> CU traffic.  This kernel cuts that traffic 4x
// but its chain per bin row (box -> tables -> cells -> barrier -> reduce -> store, ~9 us) runs with only two 72-KiB workgroups per
// CU; a per-wave LDS patch of 16 cells (tried, removed) had the same problem at 8 waves per CU (1363 us).
constexpr int ROI_FOOT = 72;          // cells of 1 KiB: 72 KiB per workgroup, two workgroups per CU
constexpr int ROI_TAB_ROWS = 64;      // sample rows of a bin
constexpr int ROI_TAB_COLS = 256;     // sample columns of a whole bin row (P * gw)
struct RoiTab { int lo, hi; float l, h; };   // lo < 0: the sample is outside the map

template <bool IN_SPLIT>
__global__ __launch_bounds__(512, 2) void roi_align_rows_kernel(const RoiArgs a) {
    __shared__ __attribute__((aligned(16))) f32x4 foot[ROI_FOOT][64];
    __shared__ RoiTab rowtab[ROI_TAB_ROWS];
    __shared__ RoiTab coltab[ROI_TAB_COLS];
    __shared__ int bounds[4];          // y0, y1, x0, x1 of the cells the valid samples touch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nvalid = a.roi_count ? min(*a.roi_count, a.R) : a.R;
    const long long nrows = (long long)nvalid * a.P;
    constexpr int C4 = 64;
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int ph = (int)(row % a.P);
        const int r = (int)(row / a.P);
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        if (a.level_out && ph == 0 && tid == 0) a.level_out[r] = lv;
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f);
        const float sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float ew = __fsub_rn(__fmul_rn(x2, sc), 0.5f);
        const float eh = __fsub_rn(__fmul_rn(y2, sc), 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)a.P));
        const int gw = (int)ceilf(__fdiv_rn(rw, (float)a.P));
        const float count = (float)max(gh * gw, 1);
        const float* fb = a.feat[lv] + (size_t)b * H * W * a.C;
        const bool tables = gh <= ROI_TAB_ROWS && a.P * gw <= ROI_TAB_COLS && gh > 0 && gw > 0;

        if (tid < 4) bounds[tid] = (tid & 1) ? -1 : (1 << 30);
        __syncthreads();
        if (tables) {
            if (tid < gh) {                                      // sample row tid of this bin row
                float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)tid, 0.5f), bh), (float)gh));
                const bool bad = (y < -1.0f) || (y > (float)H);
                if (y <= 0.f) y = 0.f;
                int lo = (int)y, hi;
                if (lo >= H - 1) { lo = hi = H - 1; y = (float)lo; } else { hi = lo + 1; }
                RoiTab t;
                t.l = __fsub_rn(y, (float)lo); t.h = __fsub_rn(1.0f, t.l); t.lo = bad ? -1 : lo; t.hi = hi;
                rowtab[tid] = t;
                if (!bad) { atomicMin(&bounds[0], lo); atomicMax(&bounds[1], hi); }
            }
            if (tid < a.P * gw) {                                // sample column ix of bin pw
                const int pw = tid / gw, ix = tid - pw * gw;
                float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                const bool bad = (x < -1.0f) || (x > (float)W);
                if (x <= 0.f) x = 0.f;
                int lo = (int)x, hi;
                if (lo >= W - 1) { lo = hi = W - 1; x = (float)lo; } else { hi = lo + 1; }
                RoiTab t;
                t.l = __fsub_rn(x, (float)lo); t.h = __fsub_rn(1.0f, t.l); t.lo = bad ? -1 : lo; t.hi = hi;
                coltab[tid] = t;
                if (!bad) { atomicMin(&bounds[2], lo); atomicMax(&bounds[3], hi); }
            }
        }
        __syncthreads();
        const int y0 = bounds[0], yl = bounds[1], x0 = bounds[2], xl = bounds[3];
        const bool any = tables && yl >= 0 && xl >= 0;           // at least one sample inside the map
        const int ny = yl - y0 + 1, nx = xl - x0 + 1;
        const bool staged = any && ny * nx <= ROI_FOOT;
        if (staged) {
            // every distinct cell once: wave w takes cells w, w + 8, ...; split rows are decoded here, once per cell
            // (all of a wave's loads are issued before the first LDS write: one cell per loop trip cost a full memory round trip each)
            constexpr int PER_WAVE = ROI_FOOT / 8;
            f32x4 t[PER_WAVE];
#pragma unroll
            for (int u = 0; u < PER_WAVE; ++u) {
                const int q = wave + 8 * u;
                if (q < ny * nx) {
                    const int cy = q / nx, cx = q - cy * nx;
                    t[u] = load_tap<IN_SPLIT>(fb + ((size_t)(y0 + cy) * W + x0 + cx) * a.C, lane);
                }
            }
#pragma unroll
            for (int u = 0; u < PER_WAVE; ++u)
                if (wave + 8 * u < ny * nx) foot[wave + 8 * u][lane] = t[u];
        }
        __syncthreads();
        for (int pw = wave; pw < a.P; pw += 8) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (staged) {
                for (int iy = 0; iy < gh; ++iy) {
                    const RoiTab ty = rowtab[iy];
                    if (ty.lo < 0) continue;
                    const int ry0 = (ty.lo - y0) * nx - x0, ry1 = (ty.hi - y0) * nx - x0;
                    for (int ix = 0; ix < gw; ++ix) {
                        const RoiTab tx = coltab[pw * gw + ix];
                        if (tx.lo < 0) continue;
                        const float w1 = __fmul_rn(ty.h, tx.h), w2 = __fmul_rn(ty.h, tx.l), w3 = __fmul_rn(ty.l, tx.h), w4 = __fmul_rn(ty.l, tx.l);
                        const f32x4 v1 = foot[ry0 + tx.lo][lane];
                        const f32x4 v2 = foot[ry0 + tx.hi][lane];
                        const f32x4 v3 = foot[ry1 + tx.lo][lane];
                        const f32x4 v4 = foot[ry1 + tx.hi][lane];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1[e]), __fmul_rn(w2, v2[e])), __fmul_rn(w3, v3[e])), __fmul_rn(w4, v4[e]));
                            acc[e] = __fadd_rn(acc[e], t);
                        }
                    }
                }
            } else if (!tables || any) {
                // too many cells or too large a grid for the tables: the direct path (roi_align_kernel's loop)
                for (int iy = 0; iy < gh; ++iy) {
                    float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bh), (float)gh));
                    const bool ybad = (y < -1.0f) || (y > (float)H);
                    if (y <= 0.f) y = 0.f;
                    int ylo = (int)y, yhi;
                    if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
                    const float ly = __fsub_rn(y, (float)ylo), hy = __fsub_rn(1.0f, ly);
                    for (int ix = 0; ix < gw; ++ix) {
                        float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                        if (ybad || (x < -1.0f) || (x > (float)W)) continue;
                        if (x <= 0.f) x = 0.f;
                        int xlo = (int)x, xhi;
                        if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
                        const float lx = __fsub_rn(x, (float)xlo), hx = __fsub_rn(1.0f, lx);
                        const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                        const f32x4 v1 = load_tap<IN_SPLIT>(fb + ((size_t)ylo * W + xlo) * a.C, lane);
                        const f32x4 v2 = load_tap<IN_SPLIT>(fb + ((size_t)ylo * W + xhi) * a.C, lane);
                        const f32x4 v3 = load_tap<IN_SPLIT>(fb + ((size_t)yhi * W + xlo) * a.C, lane);
                        const f32x4 v4 = load_tap<IN_SPLIT>(fb + ((size_t)yhi * W + xhi) * a.C, lane);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w1, v1[e]), __fmul_rn(w2, v2[e])), __fmul_rn(w3, v3[e])), __fmul_rn(w4, v4[e]));
                            acc[e] = __fadd_rn(acc[e], t);
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __fdiv_rn(acc[e], count);
            const long long bin = row * a.P + pw;
            f32x4* o4 = reinterpret_cast<f32x4*>(a.out) + (size_t)bin * C4;
            if (a.out_split) {
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 h = (_Float16)acc[e];
                    hi[e] = h;
                    lo[e] = (_Float16)((acc[e] - (float)h) * 2048.0f);
                }
                const int ch = 4 * lane;
                char* base = reinterpret_cast<char*>(o4) + (ch >> 5) * 128 + (ch & 31) * 2;
                *reinterpret_cast<f16x4*>(base) = hi;
                *reinterpret_cast<f16x4*>(base + 64) = lo;
            } else {
                o4[lane] = acc;
            }
        }
        __syncthreads();                                         // the tables and the cells are re-used by the next bin row
    }
}

// Split-row feature maps (the trunk's native activation format): TWO bins per wave, a lane owns 8 channels of its half-wave's bin, so
// a tap is one 16-B load of hi halves and one of lo' halves per lane (with 4 channels per lane they were 8-B loads and the kernel ran
// 50 % longer).  Bins 2w and 2w+1 mostly belong to one RoI (same sampling grid); where they do not, the two halves of the wave run
// loops of different length under the exec mask.  Arithmetic and its order per output value are exactly those of roi_align_kernel.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));
typedef float f32x2r __attribute__((ext_vector_type(2)));
// (packed fp32: v_pk_mul_f32 / v_pk_add_f32 work on two values per instruction, each rounded on its own -- the same value per element as
//  the scalar form; the kernel's interpolation arithmetic is half its instruction stream)
__device__ __forceinline__ void load_tap8(const float* row, int c8, f32x2r (&v)[4]) {
    const int ch = 8 * c8;
    const char* base = reinterpret_cast<const char*>(row) + (ch >> 5) * 128 + (ch & 31) * 2;
    const f16x8 h = *reinterpret_cast<const f16x8*>(base);
    const f16x8 l = *reinterpret_cast<const f16x8*>(base + 64);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2r hf = __builtin_convertvector(f16x2r{h[2 * p], h[2 * p + 1]}, f32x2r);
        const f32x2r lf = __builtin_convertvector(f16x2r{l[2 * p], l[2 * p + 1]}, f32x2r);
        v[p] = hf + lf * (1.0f / 2048.0f);
    }
}

// XCD-major processing order (RoiArgs::order).  Keys (Morton tile of the box centre, 32 x 32 px | level | index) of an image's RoIs sorted
// bitonically in LDS (R <= 8192), then the sorted RoIs dealt out to the 8 XCDs in contiguous eighths.
constexpr int ORDER_MAX = 8192;
__device__ __forceinline__ unsigned int morton6(unsigned int y, unsigned int x) {
    unsigned int m = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) m |= ((x >> b) & 1u) << (2 * b) | ((y >> b) & 1u) << (2 * b + 1);
    return m;
}
// One workgroup per image (grid = 32, the images a key can name): it counts the RoIs of every image (the offsets of its chunks in the
// eight lists depend on the images in front of it), collects and sorts its own, and deals them out.
__global__ __launch_bounds__(1024) void roi_order_kernel(const RoiArgs a, int* __restrict__ order, int* __restrict__ xlen, int xstride) {
    __shared__ unsigned long long keys[ORDER_MAX];      // ~key in the low word of a 64-bit word: amp::bitonic_desc sorts descending
    __shared__ int cnt[32], off[8], n_mine;
    const int tid = threadIdx.x, lane = tid & 63;
    const int img = blockIdx.x;
    const int R = a.roi_count ? min(*a.roi_count, a.R) : a.R;        // (a device-side count: the mask pooler is queued before the host knows it)
    if (tid < 32) cnt[tid] = 0;
    if (tid == 0) n_mine = 0;
    __syncthreads();
    for (int r0 = (tid & ~63); r0 < R; r0 += 1024) {            // counts: one ballot per image present in the wave's 64 RoIs
        const int r = r0 + lane;
        const int b = (r < R) ? min(a.batch_idx ? a.batch_idx[r] : 0, 31) : -1;
        unsigned long long todo = __ballot(b >= 0);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int bl = __shfl(b, leader, 64);
            const unsigned long long same = __ballot(b == bl);
            if (lane == leader) atomicAdd(&cnt[bl], (int)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    const int n = cnt[img];
    if (img == 0 && tid < 8) {       // chunk x of image i = sorted ranks [ceil(x n / 8), ceil((x + 1) n / 8)): the lists' lengths
        int o = 0;
        for (int i = 0; i < 32; ++i) o += (((tid + 1) * cnt[i] + 7) >> 3) - ((tid * cnt[i] + 7) >> 3);
        xlen[tid] = o;
    }
    if (n == 0) return;
    if (tid < 8) {
        int o = 0;
        for (int i = 0; i < img; ++i) o += (((tid + 1) * cnt[i] + 7) >> 3) - ((tid * cnt[i] + 7) >> 3);
        off[tid] = o;
    }
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    for (int r = tid; r < R; r += 1024) {
        const int b = min(a.batch_idx ? a.batch_idx[r] : 0, 31);
        if (b != img) continue;
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        // tile of the box centre in IMAGE pixels (32 x 32 px = 8 x 8 cells of p2), levels mixed: a chunk of the sorted list then holds the
        // same mix of cheap (p2, 1-4 samples per bin axis) and expensive (p3..p5, 4-8) RoIs as every other chunk -- sorting by level
        // first left whole XCDs with the expensive ones
        const int cx = min(max((int)((x1 + x2) * 0.5f), 0) >> 5, 63), cy = min(max((int)((y1 + y2) * 0.5f), 0) >> 5, 63);
        const unsigned int k = (morton6((unsigned)cy, (unsigned)cx) << 15) | ((unsigned)lv << 13) | (unsigned)r;
        keys[atomicAdd(&n_mine, 1)] = (unsigned long long)(~k);          // slot order does not matter: the keys (with r) are unique
    }
    for (int q = n + tid; q < n2; q += 1024) keys[q] = 0ull;
    amp::bitonic_desc<1024>(keys, n2);          // (steps below stride 64 in registers: select.h)
    for (int s_ = tid; s_ < n; s_ += 1024) {
        const unsigned int k = ~(unsigned int)keys[s_];
        const int r = (int)(k & 8191u);
        int x = min((s_ * 8) / n, 7);
        while (x < 7 && s_ >= (((x + 1) * n + 7) >> 3)) ++x;
        while (x > 0 && s_ < ((x * n + 7) >> 3)) --x;
        order[x * xstride + off[x] + (s_ - ((x * n + 7) >> 3))] = r;
    }
}

__global__ __launch_bounds__(256) void roi_align_split_kernel(const RoiArgs a) {
    const bool g_share = a.share_taps != 0;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, c8 = lane & 31;                 // C == 256: 32 lanes x 8 channels
    const int nvalid = a.roi_count ? min(*a.roi_count, a.R) : a.R;
    const long long nbins = (long long)nvalid * a.P * a.P;
    const int PP = a.P * a.P;
    const long long npairs_x = a.order ? ((long long)a.xlen[blockIdx.x & 7] * PP + 1) / 2 : 0;     // XCD-major: this XCD's own bin pairs
    const long long pair0 = a.order ? (long long)(blockIdx.x >> 3) * 4 + wave : (long long)blockIdx.x * 4 + wave;
    const long long pstep = a.order ? (long long)(gridDim.x >> 3) * 4 : (long long)gridDim.x * 4;
    for (long long pair = pair0; a.order ? pair < npairs_x : pair * 2 < nbins; pair += pstep) {
        long long bin = pair * 2 + half;
        if (a.order) {
            const long long u = bin;                                   // bin unit inside the XCD's sequence
            const int slot = (int)(u / PP);
            if (slot >= a.xlen[blockIdx.x & 7]) continue;
            bin = (long long)a.order[(blockIdx.x & 7) * a.xstride + slot] * PP + (u - (long long)slot * PP);
        }
        if (bin >= nbins) continue;
        const int pw = (int)(bin % a.P);
        const int ph = (int)((bin / a.P) % a.P);
        const int r = (int)(bin / (a.P * a.P));
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        if (a.level_out && ph == 0 && pw == 0 && c8 == 0) a.level_out[r] = lv;
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f);
        const float sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float ew = __fsub_rn(__fmul_rn(x2, sc), 0.5f);
        const float eh = __fsub_rn(__fmul_rn(y2, sc), 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)a.P));
        const int gw = (int)ceilf(__fdiv_rn(rw, (float)a.P));
        const float count = (float)max(gh * gw, 1);
        const float* fb = a.feat[lv] + (size_t)b * H * W * a.C;
        f32x2r acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = f32x2r{0.f, 0.f};
        for (int iy = 0; iy < gh; ++iy) {
            float y = __fadd_rn(__fadd_rn(sh, __fmul_rn((float)ph, bh)), __fdiv_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), bh), (float)gh));
            const bool ybad = (y < -1.0f) || (y > (float)H);
            if (y <= 0.f) y = 0.f;
            int ylo = (int)y, yhi;
            if (ylo >= H - 1) { ylo = yhi = H - 1; y = (float)ylo; } else { yhi = ylo + 1; }
            const float ly = __fsub_rn(y, (float)ylo), hy = __fsub_rn(1.0f, ly);
            // The samples of a row are less than one cell apart (gw = ceil(bin width)), so consecutive samples use the same cell pair or
            // the next one: the four taps stay in registers and only the columns that changed are fetched (the kernel is bound by the
            // 64 B/clk of the CU's vector L1, not by HBM or arithmetic: tools/bench_roi.py).  Same values, same operations, same order.
            int cxlo = -2, cxhi = -2;
            f32x2r v1[4], v2[4], v3[4], v4[4];
            for (int ix = 0; ix < gw; ++ix) {
                float x = __fadd_rn(__fadd_rn(sw, __fmul_rn((float)pw, bw)), __fdiv_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), bw), (float)gw));
                const bool bad = ybad || (x < -1.0f) || (x > (float)W);
                if (bad) continue;
                if (x <= 0.f) x = 0.f;
                int xlo = (int)x, xhi;
                if (xlo >= W - 1) { xlo = xhi = W - 1; x = (float)xlo; } else { xhi = xlo + 1; }
                const float lx = __fsub_rn(x, (float)xlo), hx = __fsub_rn(1.0f, lx);
                const float w1 = __fmul_rn(hy, hx), w2 = __fmul_rn(hy, lx), w3 = __fmul_rn(ly, hx), w4 = __fmul_rn(ly, lx);
                if (g_share && xlo == cxhi && xlo != cxlo) {               // advanced by one cell: the right column becomes the left one
#pragma unroll
                    for (int p = 0; p < 4; ++p) { v1[p] = v2[p]; v3[p] = v4[p]; }
                    cxlo = xlo;
                }
                if (!g_share || xlo != cxlo) {
                    load_tap8(fb + ((size_t)ylo * W + xlo) * a.C, c8, v1);
                    load_tap8(fb + ((size_t)yhi * W + xlo) * a.C, c8, v3);
                    cxlo = xlo;
                }
                if (!g_share || xhi != cxhi) {
                    load_tap8(fb + ((size_t)ylo * W + xhi) * a.C, c8, v2);
                    load_tap8(fb + ((size_t)yhi * W + xhi) * a.C, c8, v4);
                    cxhi = xhi;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2r s2 = ((v1[p] * w1 + v2[p] * w2) + v3[p] * w3) + v4[p] * w4;     // (-ffp-contract=off: products and sums stay separate)
                    acc[p] = acc[p] + s2;
                }
            }
        }
        float accs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) accs[e] = __fdiv_rn(acc[e >> 1][e & 1], count);
        float* orow = a.out + (size_t)bin * a.C;
        if (a.out_split) {
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)accs[e];
                hi[e] = h;
                lo[e] = (_Float16)((accs[e] - (float)h) * 2048.0f);
            }
            const int ch = 8 * c8;
            char* base = reinterpret_cast<char*>(orow) + (ch >> 5) * 128 + (ch & 31) * 2;
            *reinterpret_cast<f16x8*>(base) = hi;
            *reinterpret_cast<f16x8*>(base + 64) = lo;
        } else {
            reinterpret_cast<f32x4*>(orow)[2 * c8] = f32x4{accs[0], accs[1], accs[2], accs[3]};
            reinterpret_cast<f32x4*>(orow)[2 * c8 + 1] = f32x4{accs[4], accs[5], accs[6], accs[7]};
        }
    }
}


// roi_align_split_kernel with the sample parameters out of the inner loops.  There every lane evaluates every sample's coordinate -- two
// IEEE divisions, the edge rules, ~40 VALU instructions on values that are the same for the 32 lanes of a bin -- next to 32 instructions of
// interpolation; here lane j of a half-wave evaluates sample ROW j and sample COLUMN j of its bin once (the same operations in the same
// order: bit-identical), leaves them in a wave-private LDS table and the loops fetch a sample's {lo, hi, l, h} with one broadcast 16-B
// read.  The split rows are decoded with one fused multiply-add per value (lo' * 2^-11 is exact, so fma(lo', 2^-11, hi) rounds once exactly
// as the separate product and sum do; on gfx950 it is v_fma_mix_f32 straight from the f16 halves, one instruction instead of three).
// Sampling grids beyond 32 x 32 per bin keep the in-loop evaluation.
typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fma_mix_lo(unsigned int l, unsigned int h) {      // fma((float)l.lo16, 2^-11, (float)h.lo16)
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(l), "v"(1.0f / 2048.0f), "v"(h));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned int l, unsigned int h) {      // ... of the high halves
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(l), "v"(1.0f / 2048.0f), "v"(h));
    return d;
}
__device__ __forceinline__ void load_tap8_fma(const float* row, int c8, f32x2r (&v)[4]) {
    const int ch = 8 * c8;
    const char* base = reinterpret_cast<const char*>(row) + (ch >> 5) * 128 + (ch & 31) * 2;
    const u32x4r h = *reinterpret_cast<const u32x4r*>(base);
    const u32x4r l = *reinterpret_cast<const u32x4r*>(base + 64);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        v[p][0] = fma_mix_lo(l[p], h[p]);
        v[p][1] = fma_mix_hi(l[p], h[p]);
    }
}

__device__ __forceinline__ RoiTab sample_param(float start, int pidx, float binsz, int i, int g, int extent) {
    float v = __fadd_rn(__fadd_rn(start, __fmul_rn((float)pidx, binsz)), __fdiv_rn(__fmul_rn(__fadd_rn((float)i, 0.5f), binsz), (float)g));
    const bool bad = (v < -1.0f) || (v > (float)extent);
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= extent - 1) { lo = hi = extent - 1; v = (float)lo; } else { hi = lo + 1; }
    RoiTab t;
    t.l = __fsub_rn(v, (float)lo); t.h = __fsub_rn(1.0f, t.l); t.lo = bad ? -1 : lo; t.hi = hi;
    return t;
}

__global__ __launch_bounds__(256) void roi_align_split_tab_kernel(const RoiArgs a) {
    __shared__ __attribute__((aligned(16))) RoiTab tabs[4][2][64];          // [wave][rows | columns][half * 32 + sample]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5, c8 = lane & 31;                 // C == 256: 32 lanes x 8 channels
    RoiTab* rowtab = &tabs[wave][0][half * 32];
    RoiTab* coltab = &tabs[wave][1][half * 32];
    const int nvalid = a.roi_count ? min(*a.roi_count, a.R) : a.R;
    const long long nbins = (long long)nvalid * a.P * a.P;
    const int PP = a.P * a.P;
    const long long npairs_x = a.order ? ((long long)a.xlen[blockIdx.x & 7] * PP + 1) / 2 : 0;
    const long long pair0 = a.order ? (long long)(blockIdx.x >> 3) * 4 + wave : (long long)blockIdx.x * 4 + wave;
    const long long pstep = a.order ? (long long)(gridDim.x >> 3) * 4 : (long long)gridDim.x * 4;
    for (long long pair = pair0; a.order ? pair < npairs_x : pair * 2 < nbins; pair += pstep) {
        long long bin = pair * 2 + half;
        if (a.order) {
            const long long u = bin;
            const int slot = (int)(u / PP);
            if (slot >= a.xlen[blockIdx.x & 7]) continue;
            bin = (long long)a.order[(blockIdx.x & 7) * a.xstride + slot] * PP + (u - (long long)slot * PP);
        }
        if (bin >= nbins) continue;
        const int pw = (int)(bin % a.P);
        const int ph = (int)((bin / a.P) % a.P);
        const int r = (int)(bin / (a.P * a.P));
        const float x1 = a.rois[4 * r + 0], y1 = a.rois[4 * r + 1], x2 = a.rois[4 * r + 2], y2 = a.rois[4 * r + 3];
        const int lv = assign_level(x1, y1, x2, y2);
        if (a.level_out && ph == 0 && pw == 0 && c8 == 0) a.level_out[r] = lv;
        const int b = a.batch_idx ? a.batch_idx[r] : 0;
        const int H = a.fh[lv], W = a.fw[lv];
        const float sc = a.scale[lv];
        const float sw = __fsub_rn(__fmul_rn(x1, sc), 0.5f);
        const float sh = __fsub_rn(__fmul_rn(y1, sc), 0.5f);
        const float ew = __fsub_rn(__fmul_rn(x2, sc), 0.5f);
        const float eh = __fsub_rn(__fmul_rn(y2, sc), 0.5f);
        const float rw = __fsub_rn(ew, sw), rh = __fsub_rn(eh, sh);
        const float bh = __fdiv_rn(rh, (float)a.P), bw = __fdiv_rn(rw, (float)a.P);
        const int gh = (int)ceilf(__fdiv_rn(rh, (float)a.P));
        const int gw = (int)ceilf(__fdiv_rn(rw, (float)a.P));
        const float count = (float)max(gh * gw, 1);
        const float* fb = a.feat[lv] + (size_t)b * H * W * a.C;
        const bool tab = gh <= 32 && gw <= 32;
        if (tab) {          // lane c8 of the half: sample row c8 and sample column c8 (lanes beyond the grid write entries nobody reads)
            rowtab[c8] = sample_param(sh, ph, bh, c8, gh, H);
            coltab[c8] = sample_param(sw, pw, bw, c8, gw, W);
        }
        f32x2r acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = f32x2r{0.f, 0.f};
        for (int iy = 0; iy < gh; ++iy) {
            const RoiTab ty = tab ? rowtab[iy] : sample_param(sh, ph, bh, iy, gh, H);
            if (ty.lo < 0) continue;
            const float* frow_lo = fb + (size_t)ty.lo * W * a.C;
            const float* frow_hi = fb + (size_t)ty.hi * W * a.C;
            int cxlo = -2, cxhi = -2;
            f32x2r v1[4], v2[4], v3[4], v4[4];
            for (int ix = 0; ix < gw; ++ix) {
                const RoiTab tx = tab ? coltab[ix] : sample_param(sw, pw, bw, ix, gw, W);
                if (tx.lo < 0) continue;
                const float w1 = __fmul_rn(ty.h, tx.h), w2 = __fmul_rn(ty.h, tx.l), w3 = __fmul_rn(ty.l, tx.h), w4 = __fmul_rn(ty.l, tx.l);
                if (tx.lo == cxhi && tx.lo != cxlo) {               // advanced by one cell: the right column becomes the left one
#pragma unroll
                    for (int p = 0; p < 4; ++p) { v1[p] = v2[p]; v3[p] = v4[p]; }
                    cxlo = tx.lo;
                }
                if (tx.lo != cxlo) {
                    load_tap8_fma(frow_lo + (size_t)tx.lo * a.C, c8, v1);
                    load_tap8_fma(frow_hi + (size_t)tx.lo * a.C, c8, v3);
                    cxlo = tx.lo;
                }
                if (tx.hi != cxhi) {
                    load_tap8_fma(frow_lo + (size_t)tx.hi * a.C, c8, v2);
                    load_tap8_fma(frow_hi + (size_t)tx.hi * a.C, c8, v4);
                    cxhi = tx.hi;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2r s2 = ((v1[p] * w1 + v2[p] * w2) + v3[p] * w3) + v4[p] * w4;
                    acc[p] = acc[p] + s2;
                }
            }
        }
        float accs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) accs[e] = __fdiv_rn(acc[e >> 1][e & 1], count);
        float* orow = a.out + (size_t)bin * a.C;
        if (a.out_split) {
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)accs[e];
                hi[e] = h;
                lo[e] = (_Float16)((accs[e] - (float)h) * 2048.0f);
            }
            const int ch = 8 * c8;
            char* base = reinterpret_cast<char*>(orow) + (ch >> 5) * 128 + (ch & 31) * 2;
            *reinterpret_cast<f16x8*>(base) = hi;
            *reinterpret_cast<f16x8*>(base + 64) = lo;
        } else {
            reinterpret_cast<f32x4*>(orow)[2 * c8] = f32x4{accs[0], accs[1], accs[2], accs[3]};
            reinterpret_cast<f32x4*>(orow)[2 * c8 + 1] = f32x4{accs[4], accs[5], accs[6], accs[7]};
        }
    }
}


// (Round 4, measured and removed: a ROW CACHE -- each lane keeps the decoded values of the last lower cell row it fetched, its own 8 channels of up to 4 or 5
//  cells, in LDS (per-lane storage, no barrier) and takes the upper row of the next sample row from there instead of through the L1.  Bit-identical, and
//  SLOWER: 780 against 682-777 us at P = 7 and 365 against 321 us at P = 14 with 4 cells (40 KB of LDS, four workgroups per CU still), 920-975 / 430 us with
//  5 cells (three per CU): the two LDS round trips and the select per cell cost more than the L1 line they save.)
// (A variant with a whole BIN ROW per half-wave -- box arithmetic and column tables shared by the P bins of a row, 28 % fewer VALU
//  instructions -- was measured in round 3 and removed: faster when every tap is a cache hit (429 against 482 us at P = 7, 183 against 249 us
//  at P = 14) but SLOWER on proposal-like boxes over the p2 / p3 maps (1020 against 767 us), and touching the next bin's cache lines ahead made
//  it slower still (1765 us): what bounds the kernel there is the line traffic through the CU's vector L1, not instruction count or latency.)

}  // namespace

static int g_roi_share = getenv("AMP_ROI_SHARE") ? atoi(getenv("AMP_ROI_SHARE")) : 1;
static int g_roi_xcd = getenv("AMP_ROI_XCD") ? atoi(getenv("AMP_ROI_XCD")) : 1;      // XCD-major RoI order (roi_order_kernel): 1 = for >= 1024 RoIs (the box pooler: -10 %; the mask pooler's 1600: +5 % for a third of the fabric traffic), 2 = always, 0 = never
extern "C" void amp_debug_set_roi_xcd(int v) { g_roi_xcd = v; }
static int g_roi_tab = getenv("AMP_ROI_TAB") ? atoi(getenv("AMP_ROI_TAB")) : 1;      // 1: sample tables in LDS (roi_align_split_tab_kernel); 0: roi_align_split_kernel
extern "C" void amp_debug_set_roi_tab(int v) { g_roi_tab = v; }
static int g_roi_lanes = getenv("AMP_ROI_LANES") ? atoi(getenv("AMP_ROI_LANES")) : 1;   // 1: lane-parallel sample parameters (default); 0: the reference kernel (every lane computes every sample's parameters); 3: one workgroup per bin row, cells staged in LDS (slower, see roi_align_rows_kernel)
extern "C" void amp_debug_set_roi_lanes(int v) { g_roi_lanes = v; }

namespace amp {
int roi_align_run(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count, int R, int P,
                  float* out, int* level_out, int out_split, int in_split);
}
extern "C" int amp_roi_align(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx,
                             const int* roi_count, int R, int P, float* out, int* level_out) {
    return amp::roi_align_run(ctx, f, rois, batch_idx, roi_count, R, P, out, level_out, 0, 0);
}

extern "C" int amp_roi_align_fmt(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count, int R, int P,
                                 float* out, int* level_out, int fmt) {
    AMP_REQUIRE(fmt >= 0 && fmt < 4, "amp_roi_align_fmt: fmt is AMP_FMT_X_SPLIT | AMP_FMT_Y_SPLIT");
    return amp::roi_align_run(ctx, f, rois, batch_idx, roi_count, R, P, out, level_out, (fmt & 2) ? 1 : 0, (fmt & 1) ? 1 : 0);
}

int amp::roi_align_run(amp_ctx* ctx, const amp_fpn_feats* f, const float* rois, const int* batch_idx, const int* roi_count, int R, int P,
                       float* out, int* level_out, int out_split, int in_split) {
    AMP_REQUIRE(ctx && f && rois && out, "amp_roi_align: null argument");
    AMP_REQUIRE((!out_split && !in_split) || f->C % 32 == 0, "amp_roi_align: the split format needs C %% 32 == 0");
    AMP_REQUIRE(R >= 0 && P > 0 && f->C % 4 == 0, "amp_roi_align: bad shape");
    if (R == 0) return AMP_OK;
    RoiArgs a;
    for (int l = 0; l < 4; ++l) {
        AMP_REQUIRE(f->feat[l] && f->h[l] > 0 && f->w[l] > 0, "amp_roi_align: missing level %d", l + 2);
        a.feat[l] = f->feat[l];
        a.fh[l] = f->h[l];
        a.fw[l] = f->w[l];
        a.scale[l] = 1.0f / (float)f->stride[l];
    }
    a.rois = rois; a.batch_idx = batch_idx; a.roi_count = roi_count; a.out = out; a.level_out = level_out;
    a.R = R; a.P = P; a.C = f->C; a.out_split = out_split; a.in_split = in_split; a.share_taps = g_roi_share;
    a.order = nullptr; a.xlen = nullptr; a.xstride = 0;
    const long long nbins = (long long)R * P * P;
    long long g = (nbins + 3) / 4;
    if (g > 65536) g = 65536;
    if (g_roi_lanes == 3 && f->C == 256) {
        long long gr = (long long)R * P;
        if (gr > 16384) gr = 16384;
        if (in_split) hipLaunchKernelGGL(roi_align_rows_kernel<true>, dim3((unsigned)gr), dim3(512), 0, ctx->stream, a);
        else hipLaunchKernelGGL(roi_align_rows_kernel<false>, dim3((unsigned)gr), dim3(512), 0, ctx->stream, a);
    } else if (in_split && f->C == 256) {
        long long g2 = (nbins + 7) / 8;              // two bins per wave
        if (g2 > 65536) g2 = 65536;
        if (g_roi_xcd && R <= ORDER_MAX && R >= (g_roi_xcd >= 2 ? 64 : 1024)) {
            const int xstride = R / 8 + 64;
            const size_t need = (size_t)8 * xstride + 8;
            if (ctx->roi_order_ints < need) {
                AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (ctx->roi_order) AMP_HIP_CHECK(hipFree(ctx->roi_order));
                ctx->roi_order = nullptr; ctx->roi_order_ints = 0;
                AMP_HIP_CHECK(hipMalloc(&ctx->roi_order, need * sizeof(int)));
                ctx->roi_order_ints = need;
            }
            int* xlen = ctx->roi_order + (size_t)8 * xstride;
            hipLaunchKernelGGL(roi_order_kernel, dim3(32), dim3(1024), 0, ctx->stream, a, ctx->roi_order, xlen, xstride);
            a.order = ctx->roi_order; a.xlen = xlen; a.xstride = xstride;
            // every XCD gets the same number of workgroups: enough for the longest chunk (R / 8 + one RoI of rounding per image, <= 32 images)
            long long per_x = ((long long)(R / 8 + 33) * P * P + 7) / 8;
            if (per_x > 8192) per_x = 8192;
            g2 = per_x * 8;
        }
        if (g_roi_tab) hipLaunchKernelGGL(roi_align_split_tab_kernel, dim3((unsigned)g2), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(roi_align_split_kernel, dim3((unsigned)g2), dim3(256), 0, ctx->stream, a);
    } else if (in_split) hipLaunchKernelGGL(roi_align_kernel<true>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a);
    else if (g_roi_lanes) hipLaunchKernelGGL(roi_align_lanes_kernel<0>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(roi_align_kernel<false>, dim3((unsigned)g), dim3(256), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}
