// Empirical semantics of ds_read_b64_tr_b16 on gfx950: fill LDS with element indices, give every lane its own address,
// print what each lane receives.  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, int* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    int elem;   // element index of the lane's 8-byte address
    if (mode == 0) elem = l * 4;                                   // lane-linear
    else if (mode == 1) elem = (l & 15) * 4 + (l >> 4) * 256;      // 16-lane groups 256 elements apart
    else elem = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16;  // rows of 64 elements: lane i -> row i/4, col 4(i%4) (+16 per group)
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
    int h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
