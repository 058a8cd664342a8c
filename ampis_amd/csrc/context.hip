// Context, error reporting and raw device-memory helpers of the C ABI.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace amp {
thread_local hipEvent_t prof_e0 = nullptr, prof_e1 = nullptr;      // events of the timed launch being issued (common.h AMP_TIMED_LAUNCH)
}

namespace amp {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace amp

extern "C" {

const char* amp_last_error(void) { return amp::g_err; }

int amp_version(void) { return 100; }

int amp_init(int device, void* hip_stream, int flags, amp_ctx** out) {
    AMP_REQUIRE(out != nullptr, "amp_init: out is null");
    int ndev = 0;
    AMP_HIP_CHECK(hipGetDeviceCount(&ndev));
    AMP_REQUIRE(device >= 0 && device < ndev, "amp_init: device %d out of range (found %d HIP devices)", device, ndev);
    AMP_HIP_CHECK(hipSetDevice(device));
    amp_ctx* c = new amp_ctx();
    c->device = device;
    if (flags & AMP_STREAM_BORROW) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            amp::set_error("amp_init: hipStreamCreate -> %s", hipGetErrorString(e));
            delete c;
            return AMP_ERR_HIP;
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        amp::set_error("amp_init: hipEventCreate failed");
        delete c;
        return AMP_ERR_HIP;
    }
    if (hipMalloc(&c->zero_page, 256) != hipSuccess || hipMemset(c->zero_page, 0, 256) != hipSuccess) {
        amp::set_error("amp_init: zero page allocation failed");
        delete c;
        return AMP_ERR_HIP;
    }
    if (hipMalloc(&c->d_conv_flag, 16) != hipSuccess || hipMemset(c->d_conv_flag, 0, 16) != hipSuccess) {
        amp::set_error("amp_init: flag allocation failed");
        delete c;
        return AMP_ERR_HIP;
    }
    c->conv_mode = AMP_CONV_F16X3;
    if (const char* e = getenv("AMP_CONV_MODE")) {
        if (!strcmp(e, "f32")) c->conv_mode = AMP_CONV_F32;
        else if (!strcmp(e, "f16x3")) c->conv_mode = AMP_CONV_F16X3;
        else { amp::set_error("amp_init: AMP_CONV_MODE must be f32 or f16x3, got '%s'", e); delete c; return AMP_ERR_ARG; }
    }
    *out = c;
    return AMP_OK;
}

int amp_set_conv_mode(amp_ctx* ctx, int mode) {
    AMP_REQUIRE(ctx && (mode == AMP_CONV_F32 || mode == AMP_CONV_F16X3), "amp_set_conv_mode: bad argument");
    ctx->conv_mode = mode;
    return AMP_OK;
}
int amp_get_conv_mode(amp_ctx* ctx) { return ctx ? ctx->conv_mode : -1; }

int amp_conv_range_flag(amp_ctx* ctx, int clear, int* flag_h) {
    AMP_REQUIRE(ctx && flag_h, "amp_conv_range_flag: null argument");
    AMP_HIP_CHECK(hipMemcpyAsync(flag_h, ctx->d_conv_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (clear && *flag_h) AMP_HIP_CHECK(hipMemsetAsync(ctx->d_conv_flag, 0, sizeof(int), ctx->stream));
    return AMP_OK;
}

int amp_timer_start(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "amp_timer_start: null ctx");
    AMP_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    return AMP_OK;
}

int amp_timer_stop(amp_ctx* ctx, float* ms_h) {
    AMP_REQUIRE(ctx && ms_h, "amp_timer_stop: null argument");
    AMP_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    AMP_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    AMP_HIP_CHECK(hipEventElapsedTime(ms_h, ctx->ev0, ctx->ev1));
    return AMP_OK;
}

int amp_prof_begin(amp_ctx* ctx, int max_launches) {
    AMP_REQUIRE(ctx && max_launches > 0, "amp_prof_begin: bad argument");
    while ((int)ctx->prof_pool.size() < max_launches) {
        amp_prof_rec r;
        AMP_HIP_CHECK(hipEventCreate(&r.e0));
        AMP_HIP_CHECK(hipEventCreate(&r.e1));
        r.flops = 0; r.variant = 0;
        r.bytes = 0; r.M = r.N = r.K = 0;
        ctx->prof_pool.push_back(r);
    }
    ctx->prof_used = 0;
    ctx->prof_truncated = false;
    ctx->prof_on = true;
    return AMP_OK;
}

int amp_prof_pause(amp_ctx* ctx, int paused) {
    AMP_REQUIRE(ctx, "amp_prof_pause: null context");
    ctx->prof_on = !paused;
    return AMP_OK;
}

int amp_prof_end(amp_ctx* ctx, amp_prof_summary* out) {
    AMP_REQUIRE(ctx && out, "amp_prof_end: null argument");
    ctx->prof_on = false;
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof(*out));
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        const amp_prof_rec& r = ctx->prof_pool[i];
        float ms = 0.f;
        AMP_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
        const int v = (r.variant >= 0 && r.variant < 3) ? r.variant : 1;
        out->launches[v] += 1;
        out->ms[v] += ms;
        out->flops[v] += r.flops;
    }
    out->truncated = ctx->prof_truncated ? 1 : 0;
    return AMP_OK;
}

int amp_prof_launches(amp_ctx* ctx, amp_prof_launch* out, int cap, int* n_out) {
    AMP_REQUIRE(ctx && n_out && (out || cap == 0) && cap >= 0, "amp_prof_launches: bad argument");
    AMP_REQUIRE(!ctx->prof_on, "amp_prof_launches: call amp_prof_end first (it waits for the events)");
    *n_out = (int)ctx->prof_used;
    for (size_t i = 0; i < ctx->prof_used && (int)i < cap; ++i) {
        const amp_prof_rec& r = ctx->prof_pool[i];
        float ms = 0.f;
        AMP_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
        out[i].ms = ms; out[i].flops = r.flops; out[i].bytes = r.bytes;
        out[i].M = r.M; out[i].N = r.N; out[i].K = r.K; out[i].slot = r.variant;
    }
    return AMP_OK;
}

void amp_destroy(amp_ctx* ctx) {
    if (!ctx) return;
    (void)amp_comm_destroy(ctx);
    for (auto& r : ctx->prof_pool) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    for (int p = 0; p < 2; ++p) { if (ctx->wg_ev[p]) (void)hipEventDestroy(ctx->wg_ev[p]); if (ctx->side_ev[p]) (void)hipEventDestroy(ctx->side_ev[p]); }
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    (void)hipFree(ctx->zero_page);
    (void)hipFree(ctx->d_conv_flag);
    (void)hipFree(ctx->split_scratch);
    (void)hipFree(ctx->topk_scratch);
    if (ctx->rowtab_arena) (void)hipFree(ctx->rowtab_arena);
    (void)hipFree(ctx->roi_order);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int amp_sync(amp_ctx* ctx) {
    AMP_REQUIRE(ctx, "amp_sync: null ctx");
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return AMP_OK;
}

void* amp_stream(amp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int amp_malloc(amp_ctx* ctx, size_t bytes, void** out) {
    AMP_REQUIRE(ctx && out, "amp_malloc: null argument");
    AMP_HIP_CHECK(hipSetDevice(ctx->device));
    AMP_HIP_CHECK(hipMalloc(out, bytes ? bytes : 16));
    return AMP_OK;
}

int amp_free(amp_ctx* ctx, void* p) {
    AMP_REQUIRE(ctx, "amp_free: null ctx");
    if (p) AMP_HIP_CHECK(hipFree(p));
    return AMP_OK;
}

int amp_memcpy_h2d(amp_ctx* ctx, void* dst, const void* src_h, size_t bytes) {
    AMP_REQUIRE(ctx && dst && src_h, "amp_memcpy_h2d: null argument");
    AMP_HIP_CHECK(hipSetDevice(ctx->device));      // callable from any host thread (the train loader uploads from its collating thread)
    AMP_HIP_CHECK(hipMemcpyAsync(dst, src_h, bytes, hipMemcpyHostToDevice, ctx->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return AMP_OK;
}

int amp_memcpy_d2h(amp_ctx* ctx, void* dst_h, const void* src, size_t bytes) {
    AMP_REQUIRE(ctx && dst_h && src, "amp_memcpy_d2h: null argument");
    AMP_HIP_CHECK(hipSetDevice(ctx->device));
    AMP_HIP_CHECK(hipMemcpyAsync(dst_h, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AMP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return AMP_OK;
}

int amp_memset(amp_ctx* ctx, void* dst, int value, size_t bytes) {
    AMP_REQUIRE(ctx && dst, "amp_memset: null argument");
    AMP_HIP_CHECK(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return AMP_OK;
}

}  // extern "C"

// ---- matrix-pipe ceiling as this chip sustains it (tools/mfma_peak.py): back-to-back MFMAs on registers, no memory ----
// RANDOM = false: every lane multiplies the same two constants for ever (the multiplier inputs never toggle: the chip holds its
// full clock).  RANDOM = true: per-lane pseudo-random operands, a different pair for each of the 4 accumulators, so consecutive MFMAs
// see different inputs like a real GEMM's do -- the clock the chip holds under THAT load sets the practical ceiling.
namespace {
typedef float pk_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float pk_rand(unsigned int h) {
    h *= 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float)(h & 0xffff) / 32768.0f - 1.0f;      // [-1, 1)
}
template <int KIND, bool RANDOM>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    pk_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float a32[4], b32[4];
    pk_f16x8 a16[4], b16[4];
    const unsigned int id = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a32[t] = RANDOM ? pk_rand(id * 8 + t) : 1.0f + threadIdx.x * 1e-3f;
        b32[t] = RANDOM ? pk_rand(id * 8 + 4 + t) : 1.0f - threadIdx.x * 1e-3f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a16[t][e] = (_Float16)(RANDOM ? pk_rand(id * 64 + t * 8 + e) : 1.0f + e * 0.01f);
            b16[t][e] = (_Float16)(RANDOM ? pk_rand(id * 64 + 32 + t * 8 + e) : 1.0f - e * 0.01f);
        }
    }
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)        // static register indices: the operand pairing rotates every round
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[t], b32[(t + u) & 3], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16[t], b16[(t + u) & 3], acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 1.2345e-30f) out[0] = s;
}
}  // namespace

/* kind: 0 fp32 MFMA, 1 f16 MFMA, constant operands; 2 / 3: the same on pseudo-random operands */
extern "C" int amp_debug_mfma_peak(amp_ctx* ctx, int kind, int iters, int waves_per_simd, float* tflops_h) {
    AMP_REQUIRE(ctx && tflops_h && iters > 0 && kind >= 0 && kind <= 3 && waves_per_simd >= 1 && waves_per_simd <= 2, "amp_debug_mfma_peak: bad argument");
    hipDeviceProp_t p;
    AMP_HIP_CHECK(hipGetDeviceProperties(&p, ctx->device));
    const int blocks = p.multiProcessorCount * waves_per_simd;
    float* d = nullptr;
    AMP_HIP_CHECK(hipMalloc(&d, 64));
    for (int rep = 0; rep < 2; ++rep) {
        AMP_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
        if (kind == 0) hipLaunchKernelGGL((mfma_peak_kernel<0, false>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters);
        else if (kind == 1) hipLaunchKernelGGL((mfma_peak_kernel<1, false>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters);
        else if (kind == 2) hipLaunchKernelGGL((mfma_peak_kernel<0, true>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters);
        else hipLaunchKernelGGL((mfma_peak_kernel<1, true>), dim3(blocks), dim3(256), 0, ctx->stream, d, iters);
        AMP_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
        AMP_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    }
    float ms = 0.f;
    AMP_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    const double flop_per_mfma = (kind % 2 == 0) ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    *tflops_h = (float)((double)blocks * 4 * iters * 4 * flop_per_mfma / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
    return AMP_OK;
}
