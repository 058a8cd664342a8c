// Hardware probe (round 4): which forms of the packed-FP32 instructions return wrong lanes while OTHER kernels run on the card?
// Round 3 found box coordinates replaced by the box centre under multi-context concurrency and blamed VALU-mask wait states; hand-edited
// ISA variants of that kernel (tools/_probe) showed the wait states are irrelevant and that replacing the three
//     v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]
// of the kernel by scalar adds cures it.  This probe issues each operand-select form of v_pk_add / v_pk_mul / v_pk_fma_f32 in a loop on known
// data (inline asm: the exact encodings), checks every result against scalar v_add / v_mul / v_fma of the same operands, and counts
// mismatches per form.  Run it alone and beside a partner load (tools/pk_probe/run_probe.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define NPAT 18
struct Log { unsigned int pat, lane, wave, iter, got0, got1, want0, want1; };

__device__ __forceinline__ float sadd(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float smul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

#define CHECK(P, d, w0, w1)                                                                                             \
    do {                                                                                                                \
        if (__float_as_uint(d.x) != __float_as_uint(w0) || __float_as_uint(d.y) != __float_as_uint(w1)) {               \
            const unsigned long long n = atomicAdd(&counts[P], 1ull);                                                   \
            const unsigned long long s = atomicAdd(&counts[NPAT], 1ull);                                                \
            if (s < 256) log[s] = Log{P, threadIdx.x & 63u, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), (unsigned)it, \
                                      __float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(w0), __float_as_uint(w1)};   \
            (void)n;                                                                                                    \
        }                                                                                                               \
    } while (0)

extern "C" __global__ void pk_probe_kernel(const float* __restrict__ in, unsigned long long* counts, Log* log, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = {in[4 * t + 0], in[4 * t + 1]}, b = {in[4 * t + 2], in[4 * t + 3]};
    f2 c = {b.y, a.x};
    for (int it = 0; it < iters; ++it) {
        f2 d;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        CHECK(0, d, sadd(a.x, b.x), sadd(a.y, b.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));      // the form of box_candidates_kernel
        CHECK(1, d, sadd(a.x, b.y), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));                    // src1.lo broadcast
        CHECK(2, d, sadd(a.x, b.x), sadd(a.y, b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));                       // src1.hi broadcast
        CHECK(3, d, sadd(a.x, b.y), sadd(a.y, b.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));      // src0 swapped
        CHECK(4, d, sadd(a.y, b.x), sadd(a.x, b.y));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        CHECK(5, d, smul(a.x, b.y), smul(a.y, b.x));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        CHECK(6, d, smul(a.x, b.x), smul(a.y, b.y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        CHECK(7, d, sfma(a.x, b.y, c.x), sfma(a.y, b.x, c.y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        CHECK(8, d, sfma(a.x, b.x, c.x), sfma(a.y, b.y, c.y));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        CHECK(9, d, sadd(a.x, -b.y), sadd(a.y, -b.x));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));                       // src0.hi broadcast
        CHECK(10, d, sadd(a.y, b.x), sadd(a.y, b.y));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // src2 swapped
        CHECK(11, d, sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // src0 swapped
        CHECK(12, d, sfma(a.y, b.x, c.x), sfma(a.x, b.y, c.y));
        {   // v_fma_mix_f32 as roi_align.hip uses it: f16 halves of src0 / src2 picked by op_sel, src1 fp32
            const unsigned int ha = __float_as_uint(a.x), hc = __float_as_uint(b.y);      // any bits; the f16 halves are what they are
            float r, w;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(ha), "v"(b.x), "v"(hc));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(w) : "v"(ha));
            float w2; asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(w2) : "v"(hc));
            f2 dd = {r, 0.f}; const float want = sfma(w, b.x, w2);
            if (want == want) CHECK(13, dd, want, 0.f);
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(ha), "v"(b.x), "v"(hc));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(w) : "v"(ha));
            asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(w2) : "v"(hc));
            dd.x = r; const float want0 = sfma(w, b.x, w2);
            if (want0 == want0) CHECK(14, dd, want0, 0.f);
        }
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        CHECK(15, d, a.y, b.x);
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        CHECK(16, d, a.x, b.y);
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        CHECK(17, d, smul(a.x, b.y), smul(a.y, b.y));
        a.x = sadd(a.x, 0.37f); a.y = smul(a.y, 1.0001f); b.x = sadd(b.x, -0.11f); b.y = smul(b.y, 0.9999f);
        if ((it & 255) == 255) { a.x = in[4 * t + 0]; b.x = in[4 * t + 2]; }
    }
    if (a.x == 12345.678f) counts[NPAT + 1] = 1;      // keep the loop live
}

static hipStream_t g_st; static float* g_in; static unsigned long long* g_counts; static Log* g_log; static int g_n;
extern "C" int pk_probe_init(int blocks) {
    g_n = blocks * 256;
    if (hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking) != hipSuccess) return -1;
    float* h = (float*)malloc(sizeof(float) * 4 * g_n);
    unsigned int s = 12345u;
    for (int i = 0; i < 4 * g_n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((int)(s >> 8) % 200000 - 100000) * 1e-3f + 0.123f; }
    if (hipMalloc(&g_in, sizeof(float) * 4 * g_n) != hipSuccess) return -2;
    hipMemcpy(g_in, h, sizeof(float) * 4 * g_n, hipMemcpyHostToDevice); free(h);
    hipMalloc(&g_counts, sizeof(unsigned long long) * (NPAT + 2)); hipMemset(g_counts, 0, sizeof(unsigned long long) * (NPAT + 2));
    hipMalloc(&g_log, sizeof(Log) * 256); hipMemset(g_log, 0, sizeof(Log) * 256);
    return 0;
}
// launches x (blocks x 256 threads x iters iterations); counts_out [NPAT + 2], log_out [256 * 8 uint]
extern "C" int pk_probe_run(int blocks, int iters, int launches, unsigned long long* counts_out, unsigned int* log_out) {
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(pk_probe_kernel, dim3(blocks), dim3(256), 0, g_st, g_in, g_counts, g_log, iters);
    if (hipStreamSynchronize(g_st) != hipSuccess) return -1;
    hipMemcpy(counts_out, g_counts, sizeof(unsigned long long) * (NPAT + 2), hipMemcpyDeviceToHost);
    hipMemcpy(log_out, g_log, sizeof(Log) * 256, hipMemcpyDeviceToHost);
    return 0;
}
