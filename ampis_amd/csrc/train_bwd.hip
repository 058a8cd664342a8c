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

// d_coarse[b,y,x,:] += sum of the 2x2 fine cells (backward of nearest x2 upsampling in the FPN top-down path)
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
        f32x4 acc = o[i];
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

inline unsigned grid_for(size_t total) { return (unsigned)std::min<size_t>(std::max<size_t>((total + 255) / 256, 1), 4096); }

}  // namespace

extern "C" {

int amp_roi_align_bwd(amp_ctx* ctx, float* const dfeat[4], const int fh[4], const int fw[4], const int stride[4], int C, const float* rois,
                      const int* batch_idx, int R, int P, const float* dout) {
    AMP_REQUIRE(ctx && dfeat && fh && fw && stride && rois && dout, "amp_roi_align_bwd: null argument");
    if (R == 0) return AMP_OK;
    RoiBwdArgs a;
    for (int l = 0; l < 4; ++l) { a.dfeat[l] = dfeat[l]; a.fh[l] = fh[l]; a.fw[l] = fw[l]; a.scale[l] = 1.0f / (float)stride[l]; }
    a.rois = rois; a.batch_idx = batch_idx; a.dout = dout; a.R = R; a.P = P; a.C = C;
    const long long nbins = (long long)R * P * P;
    hipLaunchKernelGGL(roi_align_bwd_kernel, dim3((unsigned)std::min<long long>((nbins + 3) / 4, 65536)), dim3(256), 0, ctx->stream, a);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_upsample2_bwd(amp_ctx* ctx, const float* dfine, float* dcoarse, int B, int Hc, int Wc, int C) {
    AMP_REQUIRE(ctx && dfine && dcoarse && C % 4 == 0, "amp_upsample2_bwd: bad argument");
    const size_t total = (size_t)B * Hc * Wc * (C / 4);
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, dfine, dcoarse, B, Hc, Wc, C / 4);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
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

int amp_small_k_dgrad(amp_ctx* ctx, const float* dl, int ld, int K, const float* w, int C, const float* act, float* dx, size_t npix) {
    AMP_REQUIRE(ctx && dl && w && dx && C % 4 == 0 && K >= 1 && K <= ld, "amp_small_k_dgrad: bad argument");
    hipLaunchKernelGGL(small_k_dgrad_kernel, dim3(grid_for(npix * (C / 4))), dim3(256), 0, ctx->stream, dl, ld, K, w, C, act, dx, npix);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
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
