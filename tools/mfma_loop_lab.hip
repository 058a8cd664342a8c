// Synthetic compute-loop lab for the f16x3 convolution / wgrad kernels: LDS fragment reads + the 3-MFMA split product + barrier, no
// global-memory traffic.  Answers "which loop structure could run faster" without writing the kernel: tile per wave, waves per
// workgroup, K-groups per barrier, one or two accumulator sets, workgroups per CU, and the data in LDS (zeros / small pattern /
// random).  Stand-alone:   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/mfma_loop_lab.hip -o /tmp/lab && /tmp/lab
// Result (profiles/r01/mfma_loop_lab.txt): every structure lands at 1520-1690 TFLOP/s of raw f16 MFMA on random data, the MFMA-only
// loop itself at 1640 (2247 on zeros): the loop structure is not the limiter, the clock under toggling operands is.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// WAVES per WG; wave tile MT x NT (32x32 tiles); KK k-groups of 16 per barrier; ACC accumulator sets (1 or 2); READS/BAR switches; PF: fragments double-buffered in registers
template <int WAVES, int MT, int NT, int KK, int ACC, bool READS, bool BAR, int MINB>
__global__ __launch_bounds__(WAVES * 64, MINB) void lab(float* out, int nsteps, int lds_bytes_used, int fill) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // fill LDS with small numbers
    for (int i = tid; i < lds_bytes_used / 2; i += WAVES * 64) { unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        reinterpret_cast<_Float16*>(lds)[i] = fill == 0 ? (_Float16)0.f : fill == 1 ? (_Float16)(0.001f * (float)((i * 37 + 11) % 97)) : (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f) * 2.0f); }
    __syncthreads();
    f32x16 acc[MT][NT], acx[ACC == 2 ? MT : 1][ACC == 2 ? NT : 1];
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (ACC == 2) for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) acx[i][j][e] = 0.f;
    // row pitch 144 B (conflict-free b128 reads), A rows: wave-dependent, B rows after
    const int arow = (wave * MT * 32) % 128, brow = 160 + (wave * NT * 32) % 64;
    const unsigned char* A = lds + (size_t)(arow + l31) * 144 + lh * 16;
    const unsigned char* B = lds + (size_t)(brow + l31) * 144 + lh * 16;
    f16x8 ah[MT], al[MT], bh[NT], bl[NT];
    for (int i = 0; i < MT; ++i) { ah[i] = *reinterpret_cast<const f16x8*>(A + i * 32 * 144); al[i] = *reinterpret_cast<const f16x8*>(A + i * 32 * 144 + 64); }
    for (int j = 0; j < NT; ++j) { bh[j] = *reinterpret_cast<const f16x8*>(B + j * 32 * 144); bl[j] = *reinterpret_cast<const f16x8*>(B + j * 32 * 144 + 64); }
    for (int step = 0; step < nsteps; ++step) {
        const int so = (step & 3) * 8 * 144;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (READS) {
#pragma unroll
                for (int i = 0; i < MT; ++i) { ah[i] = *reinterpret_cast<const f16x8*>(A + i * 32 * 144 + (kk & 1) * 32 + so); al[i] = *reinterpret_cast<const f16x8*>(A + i * 32 * 144 + 64 + (kk & 1) * 32 + so); }
#pragma unroll
                for (int j = 0; j < NT; ++j) { bh[j] = *reinterpret_cast<const f16x8*>(B + j * 32 * 144 + (kk & 1) * 32 + so); bl[j] = *reinterpret_cast<const f16x8*>(B + j * 32 * 144 + 64 + (kk & 1) * 32 + so); }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (ACC == 2) {
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
                        acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (BAR) __syncthreads();
        else if (READS) asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int e = 0; e < 16; ++e) { s += acc[i][j][e]; if (ACC == 2) s += acx[i][j][e]; }
    if (s == 123.456f) out[blockIdx.x * blockDim.x + tid] = s;
}

template <int WAVES, int MT, int NT, int KK, int ACC, bool READS, bool BAR, int MINB>
void run(const char* name, int wg_per_cu, int lds_bytes, int fill = 2) {
    float* out; hipMalloc(&out, 1 << 24);
    auto k = lab<WAVES, MT, NT, KK, ACC, READS, BAR, MINB>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, WAVES * 64, lds_bytes);
    const int nsteps = 4000 / KK;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds_bytes, 0, out, nsteps, 60000, fill);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds_bytes, 0, out, nsteps, 60000, fill);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * WAVES * nsteps * KK * MT * NT * 3;
    const double tf = mfmas * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-46s fill %d waves/WG %d tile %dx%d kk %d acc %d reads %d bar %d | WG/CU %d (occ %d) lds %6d | %7.1f TF (%4.1f %% of 2500) | MFMA/barrier %d\n", name, fill, WAVES, MT * 32, NT * 32, KK, ACC,
           (int)READS, (int)BAR, wg_per_cu, occ, lds_bytes, tf, tf / 25.0, KK * MT * NT * 3);
    fflush(stdout);
    hipFree(out);
}

int main() {
    // baseline family: 4 waves 64x64, kk 2, two accumulator sets, 2 WG/CU, 64 KB LDS
    run<4, 2, 2, 2, 2, false, false, 2>("A0 mfma only, zeros", 2, 65536, 0);
    run<4, 2, 2, 2, 2, false, false, 2>("A0 mfma only, small pattern", 2, 65536, 1);
    run<4, 2, 2, 2, 2, false, false, 2>("A0 mfma only, random", 2, 65536, 2);
    run<4, 2, 2, 2, 2, true, true, 2>("A3 zeros", 2, 65536, 0);
    run<4, 2, 2, 2, 2, false, true, 2>("A1 mfma+barrier", 2, 65536);
    run<4, 2, 2, 2, 2, true, false, 2>("A2 mfma+reads", 2, 65536);
    run<4, 2, 2, 2, 2, true, true, 2>("A3 mfma+reads+barrier (today)", 2, 65536);
    run<4, 2, 2, 2, 1, true, true, 2>("A4 same, one accumulator set", 2, 65536);
    // 8 waves 64x64 one WG/CU (today's BN=256)
    run<8, 2, 2, 2, 2, true, true, 1>("B0 8 waves 1 WG/CU (today BN256)", 1, 98304);
    // one accumulator set, higher occupancy
    run<4, 2, 2, 1, 1, true, true, 2>("C0 4w 64x64 kk1 acc1 2 WG/CU", 2, 65536);
    
    run<8, 2, 2, 1, 1, true, true, 2>("D0 8w 64x64 kk1 acc1 2 WG/CU", 2, 65536);
    run<8, 2, 2, 2, 1, true, true, 2>("D1 8w 64x64 kk2 acc1 2 WG/CU", 2, 65536);
    // fat waves
    run<4, 4, 2, 2, 1, true, true, 2>("E0 4w 128x64 kk2 acc1 2 WG/CU", 2, 65536);
    run<4, 4, 2, 1, 1, true, true, 2>("E1 4w 128x64 kk1 acc1 2 WG/CU", 2, 65536);
    run<4, 4, 2, 2, 2, true, true, 1>("F0 4w 128x64 kk2 acc2 1 WG/CU", 1, 98304);
    run<4, 4, 2, 2, 2, true, false, 1>("F1 same, no barrier", 1, 98304);
    run<4, 4, 2, 2, 2, false, false, 1>("F2 same, mfma only", 1, 98304);
    run<4, 2, 4, 2, 2, true, true, 1>("F3 4w 64x128 kk2 acc2 1 WG/CU", 1, 98304);
    run<4, 4, 4, 1, 1, true, true, 1>("G0 4w 128x128 kk1 acc1 1 WG/CU", 1, 98304);
    return 0;
}
