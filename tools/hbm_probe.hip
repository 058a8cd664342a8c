// What the memory system of this chip sustains for the access shapes of the byte-bound convolution layers (tools/layer_roofline.py):
// whole-buffer float4 copy / fill / read, and the split-row epilogue's shape (a wave instruction = 16 rows x 64 B at a 1-KB row pitch).
// Buffers of 512 MiB (twice the Infinity Cache), three pairs used in turn.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void copy16(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void fill16(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = x;
}
__global__ void read16(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s[0] == 1.2345f && s[1] == 5.f) out[0] = s;
}
// rows of 1 KB; a workgroup of 4 waves owns 64 rows at a time (tile-kernel-like: wave w = 256-B column slab), a wave instruction covers
// 16 rows x 64 B: lane (l15 = row, lq = 16-B piece); per row group: 2 x (hi 64 B, lo 64 B) per 128-B line, 2 lines per wave
template <bool READ, bool WRITE>
__global__ void rows64(const char* __restrict__ in, char* __restrict__ out, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t r0 = (size_t)blockIdx.x * 64; r0 < nrows; r0 += (size_t)gridDim.x * 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t off = (r0 + i * 16 + l15) * 1024 + wave * 256 + lq * 16;
            f32x4 v[4] = {acc, acc, acc, acc};
            if (READ) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(in + off + q * 64);
            }
            if (WRITE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(out + off + q * 64) = v[q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += v[q];
            }
        }
    }
    if (!WRITE && acc[0] == 1.2345f && acc[1] == 5.f) *reinterpret_cast<f32x4*>(out) = acc;
}


// L2-resident traffic (every workgroup sweeps the same 2 MiB = 2048 rows of 1 KB, `passes` times): what a CU's vector memory path moves
// per access SHAPE, with HBM out of the picture.  PAT 0: a wave instruction = 16 rows x 64 B (the split-row epilogue: hi halves of 16
// rows, then their lo halves); PAT 1: 8 rows x 128 B (whole lines); PAT 2: 4 rows x 256 B; PAT 3: 1 KB contiguous.
template <int PAT, bool WRITE>
__global__ void l2_shape(char* __restrict__ buf, int passes, f32x4* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int rows_per_instr = PAT == 0 ? 16 : PAT == 1 ? 8 : PAT == 2 ? 4 : 1;
    const int lanes_per_row = 64 / rows_per_instr;               // 4, 8, 16, 64
    const int r = lane / lanes_per_row, p = lane % lanes_per_row;
    const int ninstr = 1024 / (lanes_per_row * 16);              // instructions that cover the rows' 1 KB: 16, 8, 4, 1
    for (int it = 0; it < passes; ++it) {
        for (int r0 = ((blockIdx.x * 4 + wave) * 16) & 2047, n = 0; n < 2048 / 64; ++n, r0 = (r0 + 64) & 2047) {
            // this wave covers rows r0 .. r0+15 completely (16 KB), in instructions of its shape
#pragma unroll 4
            for (int g = 0; g < 16 / rows_per_instr; ++g)
#pragma unroll 4
                for (int q = 0; q < ninstr; ++q) {
                    char* a = buf + (size_t)(r0 + g * rows_per_instr + r) * 1024 + q * (lanes_per_row * 16) + p * 16;
                    if (WRITE) *reinterpret_cast<f32x4*>(a) = acc;
                    else acc += *reinterpret_cast<const f32x4*>(a);
                }
        }
    }
    if (acc[0] == 1.2345f && acc[1] == 5.f) sink[0] = acc;
}


// data-dependent rates: the same copies and fills with RANDOM words (hash of the index) instead of constant bytes
__global__ void fill_hash(unsigned int* __restrict__ out, size_t n, unsigned int seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned int x = (unsigned int)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        out[i] = x;
    }
}
__global__ void fill16_hash(f32x4* __restrict__ out, size_t n, unsigned int seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned int x = (unsigned int)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        f32x4 v; v[0] = __uint_as_float(x & 0x3fffffffu); v[1] = __uint_as_float((x * 3u) & 0x3fffffffu); v[2] = __uint_as_float((x * 5u) & 0x3fffffffu); v[3] = __uint_as_float((x * 7u) & 0x3fffffffu);
        out[i] = v;
    }
}

// `hbm_probe chain`: a producer -> consumer chain (copy x -> y, then y -> x, ...) on buffers of S MiB: what a layer gets that reads the
// tensor the previous launch wrote, as a function of the tensor size against the 256-MB Infinity Cache
static int chain_main() {
    const size_t cap = (size_t)512 << 20;
    char *x, *y; CK(hipMalloc(&x, cap)); CK(hipMalloc(&y, cap)); CK(hipMemset(x, 1, cap)); CK(hipMemset(y, 2, cap));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("producer -> consumer chain, split-row shape copy x -> y -> x ... (grid 8192 x 256):\n");
    for (int mib : {16, 32, 48, 64, 96, 128, 192, 256, 512}) {
        const size_t bytes = (size_t)mib << 20, nrows = bytes / 1024;
        const int reps = 24;
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((rows64<true, true>), dim3(8192), dim3(256), 0, 0, (i & 1) ? y : x, (i & 1) ? x : y, nrows);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rows64<true, true>), dim3(8192), dim3(256), 0, 0, (i & 1) ? y : x, (i & 1) ? x : y, nrows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  tensor %4d MiB (working set %4d MiB)  %8.1f us  %6.2f TB/s\n", mib, 2 * mib, ms / reps * 1e3, 2.0 * bytes / (ms / reps * 1e-3) / 1e12);
    }
    printf("read-only sweeps of one buffer, repeated (float4 read, grid 8192 x 256):\n");
    for (int mib : {16, 32, 64, 128, 192, 256, 512}) {
        const size_t bytes = (size_t)mib << 20;
        const int reps = 24;
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, bytes / 16);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(read16, dim3(8192), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  buffer %4d MiB  %8.1f us  %6.2f TB/s\n", mib, ms / reps * 1e3, 1.0 * bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "chain") return chain_main();
    const size_t bytes = (size_t)512 << 20, n16 = bytes / 16, nrows = bytes / 1024;
    std::vector<char*> a(3), b(3);
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&a[i], bytes)); CK(hipMalloc(&b[i], bytes)); CK(hipMemset(a[i], 1, bytes)); CK(hipMemset(b[i], 2, bytes)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto launch) {
        for (int i = 0; i < 3; ++i) launch(i % 3);
        CK(hipDeviceSynchronize());
        const int reps = 12;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(i % 3);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %8.1f us  %6.2f TB/s\n", name, ms / reps * 1e3, moved / (ms / reps * 1e-3) / 1e12);
    };
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d x 256 threads\n", grid);
        timeit("float4 copy (512 MiB -> 512 MiB)", 2.0 * bytes, [&](int i) { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const f32x4*)a[i], (f32x4*)b[i], n16); });
        timeit("float4 fill (512 MiB)", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL(fill16, dim3(grid), dim3(256), 0, 0, (f32x4*)b[i], n16, 1.f); });
        timeit("float4 read (512 MiB)", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const f32x4*)a[i], (f32x4*)b[i], n16); });
        timeit("split-row shape, copy", 2.0 * bytes, [&](int i) { hipLaunchKernelGGL((rows64<true, true>), dim3(grid), dim3(256), 0, 0, a[i], b[i], nrows); });
        timeit("split-row shape, write only", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL((rows64<false, true>), dim3(grid), dim3(256), 0, 0, a[i], b[i], nrows); });
        timeit("split-row shape, read only", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL((rows64<true, false>), dim3(grid), dim3(256), 0, 0, a[i], b[i], nrows); });
    }

    {
        const int passes = 8, grid = 1024;
        const double moved = (double)grid * 4 * passes * (2048 / 64) * 16384.0;
        printf("L2-resident 2 MiB, %d workgroups x 4 waves, every wave sweeps 16-row groups (bytes requested by the instructions):\n", grid);
        timeit("  read, 16 rows x 64 B per instruction", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<0, false>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  read, 8 rows x 128 B", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<1, false>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  read, 4 rows x 256 B", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<2, false>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  read, 1 KB contiguous", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<3, false>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  write, 16 rows x 64 B per instruction", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<0, true>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  write, 8 rows x 128 B", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<1, true>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  write, 4 rows x 256 B", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<2, true>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
        timeit("  write, 1 KB contiguous", moved, [&](int i) { hipLaunchKernelGGL((l2_shape<3, true>), dim3(grid), dim3(256), 0, 0, a[0], passes, (f32x4*)b[0]); });
    }

    printf("RANDOM data in the source buffers:\n");
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fill_hash, dim3(4096), dim3(256), 0, 0, (unsigned int*)a[i], bytes / 4, 17u + i);
    CK(hipDeviceSynchronize());
    for (int grid : {2048, 32768}) {
        printf("grid %d x 256 threads\n", grid);
        timeit("float4 copy of random data", 2.0 * bytes, [&](int i) { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const f32x4*)a[i], (f32x4*)b[i], n16); });
        timeit("float4 read of random data", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const f32x4*)a[i], (f32x4*)b[i], n16); });
        timeit("float4 fill with random words", 1.0 * bytes, [&](int i) { hipLaunchKernelGGL(fill16_hash, dim3(grid), dim3(256), 0, 0, (f32x4*)b[i], n16, 99u + i); });
        timeit("split-row shape, copy of random data", 2.0 * bytes, [&](int i) { hipLaunchKernelGGL((rows64<true, true>), dim3(grid), dim3(256), 0, 0, a[i], b[i], nrows); });
    }
    // the mix of a res2 conv3 + residual launch: read 1.25 x, write 1 x  (copy + an extra quarter read)
    return 0;
}
