// probe of ds_read_b64_tr_b16 (gfx950): which element does each lane receive?  build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;          // element (row, col) = row * 64 + col
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    __attribute__((address_space(3))) s16x4* a = (__attribute__((address_space(3))) s16x4*)(lds + (4 * g + q) * 64 + 4 * p);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 1) {
        if (lane % 16 < 3 || lane % 16 == 15) {
            printf("lane %2d:", lane);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[lane * 4 + e] / 64, h[lane * 4 + e] % 64);
            printf("\n");
        }
    }
    return 0;
}
