// Memory-bound NHWC helpers of the backbone:
//   preprocess  : uint8 BGR [B,H,W,3] -> fp32 [B,Hp,Wp,4], (x - mean) / std, zero pad (GeneralizedRCNN.preprocess_image
//                 + ImageList.from_tensors); the 4th channel is a zero so the stem conv reads aligned float4 pixels
//   maxpool3x3s2: stem max-pool (kernel 3, stride 2, pad 1)
//   subsample2  : p6 = max_pool2d(p5, kernel 1, stride 2) = p5[:, ::2, ::2]
// All are HBM-bound: one float4 (16 B) per lane, channels fastest, grid-stride loops.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 pp_h4 __attribute__((ext_vector_type(4)));
// SPLIT: a pixel's 16 bytes are the f16 hi halves of its 4 values, then the lo' halves ((v - hi) * 2^11) -- what the f16x3 stem would make
// of the fp32 pixel tap by tap (conv.hip split8); stem_pool_f16x3_kernel then stages the halves as they are
template <bool SPLIT>
__global__ void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int H, int W, int Hp,
                                  int Wp, float m0, float m1, float m2, float s0, float s1, float s2, const int* __restrict__ img_hw) {
    const size_t total = (size_t)B * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp);
        const size_t t = i / Wp;
        const int y = (int)(t % Hp);
        const int b = (int)(t / Hp);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int vh = img_hw ? img_hw[2 * b] : H, vw = img_hw ? img_hw[2 * b + 1] : W;   // ImageList.from_tensors pads with 0 AFTER normalising
        if (y < vh && x < vw) {
            const uint8_t* p = img + ((size_t)(b * H + y) * W + x) * 3;
            v[0] = __fdiv_rn(__fsub_rn((float)p[0], m0), s0);
            v[1] = __fdiv_rn(__fsub_rn((float)p[1], m1), s1);
            v[2] = __fdiv_rn(__fsub_rn((float)p[2], m2), s2);
        }
        if (SPLIT) {
            pp_h4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 h = (_Float16)v[e];
                hi[e] = h;
                lo[e] = (_Float16)((v[e] - (float)h) * 2048.0f);
            }
            pp_h4* o = reinterpret_cast<pp_h4*>(out) + 2 * i;
            o[0] = hi; o[1] = lo;
        } else {
            reinterpret_cast<f32x4*>(out)[i] = v;
        }
    }
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <bool Y_SPLIT>
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4,
                                    int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = x4[((size_t)(b * H + iy) * W + ix) * C4 + c];
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        if (Y_SPLIT) {      // the trunk's native activation format (AMP_CONV_F16X3 inference): per 32 channels 64 B of hi halves, 64 B of lo' halves
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hh = (_Float16)m[e];
                hi[e] = hh;
                lo[e] = (_Float16)((m[e] - (float)hh) * 2048.0f);
            }
            const int ch = 4 * c;
            char* base = reinterpret_cast<char*>(y + (i - c) * 4) + (ch >> 5) * 128 + (ch & 31) * 2;
            *reinterpret_cast<f16x4*>(base) = hi;
            *reinterpret_cast<f16x4*>(base + 64) = lo;
        } else {
            reinterpret_cast<f32x4*>(y)[i] = m;
        }
    }
}

__global__ void subsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4, int Ho,
                                  int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        reinterpret_cast<f32x4*>(y)[i] = x4[((size_t)(b * H + 2 * oy) * W + 2 * ox) * C4 + c];
    }
}

inline int grid_for(size_t total, int block) {
    size_t g = (total + block - 1) / block;
    return (int)(g < 2048 ? (g ? g : 1) : 2048);
}

}  // namespace

extern "C" {

int amp_preprocess(amp_ctx* ctx, const uint8_t* img_bgr, int B, int H, int W, int Hp, int Wp, const float mean[3],
                   const float std[3], const int* img_hw, float* out) {
    return amp::preprocess_run(ctx, img_bgr, B, H, W, Hp, Wp, mean, std, img_hw, out, 0);
}

int amp_maxpool3x3s2(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y) {
    return amp::maxpool_run(ctx, x, B, H, W, C, y, 0);
}

int amp_subsample2(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y) {
    AMP_REQUIRE(ctx && x && y, "amp_subsample2: null argument");
    AMP_REQUIRE(C % 4 == 0 && B > 0 && H > 0 && W > 0, "amp_subsample2: bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, y, B, H, W, C / 4, Ho,
                       Wo);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

}  // extern "C"

int amp::preprocess_run(amp_ctx* ctx, const uint8_t* img_bgr, int B, int H, int W, int Hp, int Wp, const float mean[3], const float std[3],
                        const int* img_hw, float* out, int out_split) {
    AMP_REQUIRE(ctx && img_bgr && out && mean && std, "amp_preprocess: null argument");
    AMP_REQUIRE(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, "amp_preprocess: bad shape");
    const size_t total = (size_t)B * Hp * Wp;
    if (out_split) hipLaunchKernelGGL(preprocess_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, img_bgr, out, B, H, W,
                                      Hp, Wp, mean[0], mean[1], mean[2], std[0], std[1], std[2], img_hw);
    else hipLaunchKernelGGL(preprocess_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, img_bgr, out, B, H, W,
                            Hp, Wp, mean[0], mean[1], mean[2], std[0], std[1], std[2], img_hw);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}

int amp::maxpool_run(amp_ctx* ctx, const float* x, int B, int H, int W, int C, float* y, int y_split) {
    AMP_REQUIRE(ctx && x && y, "amp_maxpool3x3s2: null argument");
    AMP_REQUIRE(C % 4 == 0 && B > 0 && H > 0 && W > 0, "amp_maxpool3x3s2: bad shape (C %% 4 != 0?)");
    AMP_REQUIRE(!y_split || C % 32 == 0, "amp_maxpool3x3s2: the split format needs C %% 32 == 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    if (y_split) hipLaunchKernelGGL(maxpool3x3s2_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, y, B, H, W, C / 4, Ho, Wo);
    else hipLaunchKernelGGL(maxpool3x3s2_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, x, y, B, H, W, C / 4, Ho, Wo);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}
