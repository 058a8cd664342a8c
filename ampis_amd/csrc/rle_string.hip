// COCO compressed-RLE "counts" strings on the device (what compress_pred stores: ampis/data_utils.py:275 ->
// pycocotools mask.encode -> maskApi.c rleToString).  The run lengths of a mask are difference-coded (run i > 2 minus run i - 2),
// each value is cut into 5-bit groups, low group first, bit 5 of a character says "more follows", characters are offset by 48.
// Host restatement: rle_host.hip amp_rle_to_string; this file produces the same bytes without the run lengths ever crossing PCIe
// (11.5 MB of uint32 runs against 3.6 MB of characters per batch of 1600 masks, and 6 ms of host encoding per batch gone).
//
// One workgroup per mask, a thread per run: the character count of a run depends on its own value only, so the layout is a prefix sum --
// inside a wave with shuffles, across the four waves and the 256-run chunks with a running base, across masks with a one-workgroup scan.
#include "common.h"

namespace {

// characters of the signed value x (maskApi.c rleToString): groups of 5 bits until the rest is all sign
__device__ __forceinline__ int rle_nchar(long long x) {
    int n = 0;
    bool more = true;
    while (more) {
        const int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? (x != -1) : (x != 0);
        ++n;
    }
    return n;
}

__device__ __forceinline__ long long rle_delta(const unsigned int* cnt, int i) {
    long long x = (long long)cnt[i];
    if (i > 2) x -= (long long)cnt[i - 2];
    return x;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// str_len[i] = characters of mask i (0 for an empty run list).  One 256-thread workgroup per mask: the masks of a batch differ by two
// orders of magnitude in their number of runs, and with a wave per mask the longest one was the kernel's time.
__global__ __launch_bounds__(256) void rle_str_len_kernel(const unsigned int* __restrict__ pool, const unsigned long long* __restrict__ off,
                                                          const int* __restrict__ len, int n, int* __restrict__ str_len) {
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x;
    const int m = len[i];
    const unsigned int* cnt = pool + off[i];
    int total = 0;
    for (int r0 = threadIdx.x; r0 < m; r0 += 4 * 256) {          // four runs per thread in flight
        long long x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = (r0 + 256 * u < m) ? rle_delta(cnt, r0 + 256 * u) : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) total += (r0 + 256 * u < m) ? rle_nchar(x[u]) : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
    if (lane == 0) wsum[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) str_len[i] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of str_len -> str_off (one workgroup; n is a batch's detection count: thousands), total -> *total_out
__global__ __launch_bounds__(1024) void rle_str_scan_kernel(const int* __restrict__ str_len, int n, unsigned long long* __restrict__ str_off,
                                                            unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const unsigned long long v = (i < n) ? (unsigned long long)str_len[i] : 0ull;
        unsigned long long s = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(s, d, 64);
            if (lane >= d) s += o;
        }
        if (lane == 63) wsum[wave] = s;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (i < n) str_off[i] = before + s - v;
        __syncthreads();
        if (tid == 1023) carry = before + s;
        __syncthreads();
    }
    if (tid == 0) *total_out = carry;
}

// the characters of mask i at str + str_off[i]; nothing is written beyond cap (the caller checks *total_out against cap).
// One workgroup per mask, 256 runs per trip: positions from a wave scan + the four wave totals.
__global__ __launch_bounds__(256) void rle_str_write_kernel(const unsigned int* __restrict__ pool, const unsigned long long* __restrict__ off,
                                                            const int* __restrict__ len, int n, const unsigned long long* __restrict__ str_off,
                                                            char* __restrict__ str, unsigned long long cap) {
    __shared__ int wtot[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int m = len[i];
    const unsigned int* cnt = pool + off[i];
    unsigned long long base = str_off[i];
    long long x_next = (tid < m) ? rle_delta(cnt, tid) : 0;          // the next chunk's values are fetched while this one is written
    int buf = 0;
    for (int r0 = 0; r0 < m; r0 += 256, buf ^= 1) {
        const int r = r0 + tid;
        long long x = x_next;
        x_next = (r + 256 < m) ? rle_delta(cnt, r + 256) : 0;
        const int nc = (r < m) ? rle_nchar(x) : 0;
        const int incl = wave_incl_scan(nc, lane);
        if (lane == 63) wtot[buf][wave] = incl;
        __syncthreads();                       // (two buffers: the next trip's totals do not overwrite what a slower wave still reads)
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int t = wtot[buf][w];
            if (w < wave) before += t;
            all += t;
        }
        unsigned long long p = base + (unsigned long long)(before + incl - nc);
        if (r < m && p + (unsigned long long)nc <= cap) {
            bool more = true;
            while (more) {
                int c = (int)(x & 0x1f);
                x >>= 5;
                more = (c & 0x10) ? (x != -1) : (x != 0);
                if (more) c |= 0x20;
                str[p++] = (char)(c + 48);
            }
        }
        base += (unsigned long long)all;
    }
}

}  // namespace

namespace amp {
int rle_strings_run(amp_ctx* ctx, const unsigned int* pool, const unsigned long long* off, const int* len, int n, char* str,
                    unsigned long long cap, unsigned long long* str_off, int* str_len, unsigned long long* total) {
    if (n <= 0) return hipMemsetAsync(total, 0, sizeof(unsigned long long), ctx->stream) == hipSuccess ? AMP_OK : AMP_ERR_HIP;
    hipLaunchKernelGGL(rle_str_len_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, pool, off, len, n, str_len);
    hipLaunchKernelGGL(rle_str_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, str_len, n, str_off, total);
    hipLaunchKernelGGL(rle_str_write_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, pool, off, len, n, str_off, str, cap);
    AMP_HIP_CHECK(hipGetLastError());
    return AMP_OK;
}
}  // namespace amp

/* Device op behind the model's string output, exposed for the parity tests: all pointers are device pointers.
 * pool: uint32 run lengths; mask i = pool[off[i] .. off[i] + len[i]); str (cap bytes) receives the n counts strings back to back,
 * string i = str[str_off[i] .. str_off[i] + str_len[i]); *total = bytes needed (> cap: the tail was not written). */
extern "C" int amp_rle_strings_device(amp_ctx* ctx, const uint32_t* pool, const unsigned long long* off, const int* len, int n, char* str,
                                      unsigned long long cap, unsigned long long* str_off, int* str_len, unsigned long long* total) {
    AMP_REQUIRE(ctx && total && n >= 0 && (n == 0 || (pool && off && len && str && str_off && str_len)), "amp_rle_strings_device: bad argument");
    return amp::rle_strings_run(ctx, pool, off, len, n, str, cap, str_off, str_len, total);
}
