// uint8 bilinear resize with the arithmetic of Pillow's ImagingResample (the `Image.resize(..., BILINEAR)` detectron2's
// ResizeShortestEdge / ResizeTransform applies to every image before DefaultPredictor's forward): antialiased
// triangle filter whose support grows with the down-scale factor, coefficients normalised in double and rounded to 22-bit fixed
// point, a horizontal pass then a vertical pass with the intermediate rounded to uint8.  Bit-exact with PIL (tests/test_resize_gpu.py),
// so the network sees the same bytes as with the host resize it replaces -- which cost more than the batch-1 forward pass.
#include <cmath>
#include <vector>

#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0)
void coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
    const double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            const double w = t < 1.0 ? 1.0 - t : 0.0;
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

__device__ __forceinline__ unsigned char clip8(int v) {
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out[y][xx][c] = clip8(round_half + sum_x in[y][xmin+x][c] * k[xx][x])
// opitch: pixels per output row (>= w: the output may be the top-left part of a wider frame); flip: output column xx lands at w - 1 - xx
__global__ void resample_h_kernel(const unsigned char* __restrict__ in, int H, int W, unsigned char* __restrict__ out, int w,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int opitch, int flip) {
    const size_t total = (size_t)H * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % w), y = (int)(i / w);
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        const int* k = kk + (size_t)xx * ksize;
        const unsigned char* p = in + ((size_t)y * W + xmin) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int c = k[x];
            s0 += p[3 * x] * c; s1 += p[3 * x + 1] * c; s2 += p[3 * x + 2] * c;
        }
        unsigned char* o = out + ((size_t)y * opitch + (flip ? w - 1 - xx : xx)) * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}

__global__ void resample_v_kernel(const unsigned char* __restrict__ in, int H, int W, unsigned char* __restrict__ out, int h,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int opitch, int flip) {
    const size_t total = (size_t)h * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), yy = (int)(i / W);
        const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
        const int* k = kk + (size_t)yy * ksize;
        const unsigned char* p = in + ((size_t)ymin * W + x) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const int c = k[y];
            const unsigned char* q = p + (size_t)y * W * 3;
            s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c;
        }
        unsigned char* o = out + ((size_t)yy * opitch + (flip ? W - 1 - x : x)) * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}

// no resize: rows copied into a pitched frame, optionally mirrored
__global__ void copy_flip_kernel(const unsigned char* __restrict__ in, int H, int W, unsigned char* __restrict__ out, int opitch, int flip) {
    const size_t total = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const unsigned char* p = in + i * 3;
        unsigned char* o = out + ((size_t)y * opitch + (flip ? W - 1 - x : x)) * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    }
}

}  // namespace

extern "C" {

/* src [H,W,3] u8 (device) -> dst [h,w,3] u8 (device); tmp: device scratch of at least H*w*3 + 4*(2*(h+w) + h*ky + w*kx) bytes,
 * amp_resize_scratch_bytes(H, W, h, w) says how much. */
size_t amp_resize_scratch_bytes(int H, int W, int h, int w) {
    if (H <= 0 || W <= 0 || h <= 0 || w <= 0) return 0;
    const int kx = (int)ceil(std::max(1.0, (double)W / w)) * 2 + 1, ky = (int)ceil(std::max(1.0, (double)H / h)) * 2 + 1;
    return (((size_t)H * w * 3 + 255) & ~(size_t)255) + 4 * ((size_t)2 * (h + w) + (size_t)h * ky + (size_t)w * kx) + 1024;
}

/* The same resize into the top-left h x w pixels of a frame whose rows are dst_pitch pixels apart (dst_pitch >= w), optionally mirrored
 * left-right afterwards -- detectron2's ResizeShortestEdge + RandomFlip of a training image, written where ImageList.from_tensors would
 * stack it (ampis/data_utils.py:171-175 DatasetMapper; round 4: the train loader does this on the device).  Pixels of the frame outside
 * the h x w part are not touched. */
int amp_resize_flip_u8(amp_ctx* ctx, const unsigned char* src, int H, int W, unsigned char* dst, int dst_pitch, int h, int w, int flip, void* tmp) {
    AMP_REQUIRE(ctx && src && dst && H > 0 && W > 0 && h > 0 && w > 0 && dst_pitch >= w, "amp_resize_flip_u8: bad argument");
    AMP_REQUIRE(tmp || (h == H && w == W), "amp_resize_flip_u8: a resize needs scratch (amp_resize_scratch_bytes)");
    if (h == H && w == W) {
        if (!flip && dst_pitch == w) AMP_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, ctx->stream));
        else hipLaunchKernelGGL(copy_flip_kernel, dim3((unsigned)std::min<size_t>(((size_t)H * W + 255) / 256, 16384)), dim3(256), 0, ctx->stream, src, H, W, dst, dst_pitch, flip);
        AMP_HIP_CHECK(hipGetLastError());
        return AMP_OK;
    }
    std::vector<int> xb, xk, yb, yk;
    int kx = 0, ky = 0;
    coeffs(W, w, xb, xk, kx);
    coeffs(H, h, yb, yk, ky);
    unsigned char* mid = reinterpret_cast<unsigned char*>(tmp);
    int* tab = reinterpret_cast<int*>(mid + (((size_t)H * w * 3 + 255) & ~(size_t)255));
    int* d_xb = tab; int* d_xk = d_xb + xb.size(); int* d_yb = d_xk + xk.size(); int* d_yk = d_yb + yb.size();
    AMP_HIP_CHECK(hipMemcpyAsync(d_xb, xb.data(), xb.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipMemcpyAsync(d_xk, xk.data(), xk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipMemcpyAsync(d_yb, yb.data(), yb.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipMemcpyAsync(d_yk, yk.data(), yk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // the host vectors go out of scope
    const unsigned char* vin = src;
    int vW = W;
    const bool two = (w != W) && (h != H);
    if (w != W) {       // Pillow: horizontal pass first, into a temporary when a vertical pass follows (then the flip belongs to the vertical pass)
        hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)std::min<size_t>(((size_t)H * w + 255) / 256, 16384)), dim3(256), 0, ctx->stream,
                           src, H, W, two ? mid : dst, w, d_xb, d_xk, kx, two ? w : dst_pitch, two ? 0 : flip);
        vin = mid; vW = w;
    }
    if (h != H) {
        hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)std::min<size_t>(((size_t)h * vW + 255) / 256, 16384)), dim3(256), 0, ctx->stream,
                           vin, H, vW, dst, h, d_yb, d_yk, ky, dst_pitch, flip);
    }
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp_resize_bilinear_u8(amp_ctx* ctx, const unsigned char* src, int H, int W, unsigned char* dst, int h, int w, void* tmp) {
    AMP_REQUIRE(ctx && src && dst && tmp && H > 0 && W > 0 && h > 0 && w > 0, "amp_resize_bilinear_u8: bad argument");
    return amp_resize_flip_u8(ctx, src, H, W, dst, w, h, w, 0, tmp);
}

}  // extern "C"
