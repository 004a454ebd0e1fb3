// Stand-alone probe: can a wave that loads MFMA-shaped fragments of a wide fp32 row-major tensor STRAIGHT from global memory (no LDS ring, no
// barrier) stream it at the copy rate?  Pattern of a 32x32x16 bf16-MFMA A operand: lane (row i = lane & 31, half h = lane >> 5) reads, per
// 16-column stage, two 16-B chunks of row i (columns 16s + 4h.. and 16s + 8 + 4h..).  A wave owns 32 rows and walks all N/16 stages, the
// loads of the next tile are issued before the current one is consumed.  Output: one float4 per lane per tile (a stand-in for the thin dX).
// build: hipcc --offload-arch=gfx950 -O3 rowfrag_probe.hip -o rowfrag_probe ; usage: rowfrag_probe M N
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// MODE 2: reduction-major operand (weight-gradient orientation): lane (channel c = lane & 31, half h) reads, per 16-row group and 32-channel
// block, eight DWORDS: rows 8h .. 8h + 7 of its channel (a wave instruction = two 128-B row segments)
template <int NB>
__global__ __launch_bounds__(256) void kcol(const float* __restrict__ g, float* __restrict__ out, long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int N = NB * 32;
    const long ngroups = M / 16;
    float acc = 0.f;
    float cur[8 * NB], nxt[8 * NB];
    auto load = [&](long grp, float (&r)[8 * NB]) {
        const float* base = g + (grp * 16 + 8 * h) * N + c;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) r[b * 8 + e] = base[(long)e * N + 32 * b];
    };
    long grp = (long)blockIdx.x * 4 + wave;
    const long stride = (long)gridDim.x * 4;
    if (grp < ngroups) load(grp, cur);
    for (; grp < ngroups; grp += stride) {
        if (grp + stride < ngroups) load(grp + stride, nxt);
#pragma unroll
        for (int s = 0; s < 8 * NB; ++s) acc += cur[s];
#pragma unroll
        for (int s = 0; s < 8 * NB; ++s) cur[s] = nxt[s];
    }
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NS, int MODE>      // NS = N / 16 stages; MODE 0: MFMA-fragment pattern, 1: plain row-linear float4 stream (reference)
__global__ __launch_bounds__(256) void k(const float* __restrict__ g, float* __restrict__ out, long M, int blocks_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int N = NS * 16;
    const long ntiles = M / 32;
    float4 acc = make_float4(0, 0, 0, 0);
    float4 cur[2 * NS], nxt[2 * NS];
    auto load = [&](long tile, float4 (&r)[2 * NS]) {
        if (MODE == 0) {
            const float* row = g + (tile * 32 + i) * N + 4 * h;
#pragma unroll
            for (int s = 0; s < NS; ++s) { r[2 * s] = ld4(row + 16 * s); r[2 * s + 1] = ld4(row + 16 * s + 8); }
        } else {
            const float* base = g + tile * 32 * N;
#pragma unroll
            for (int s = 0; s < 2 * NS; ++s) r[s] = ld4(base + (s * 64 + lane) * 4);
        }
    };
    long tile = (long)blockIdx.x * 4 + wave;
    const long stride = (long)gridDim.x * 4;
    if (tile < ntiles) load(tile, cur);
    for (; tile < ntiles; tile += stride) {
        if (tile + stride < ntiles) load(tile + stride, nxt);
#pragma unroll
        for (int s = 0; s < 2 * NS; ++s) { acc.x += cur[s].x; acc.y += cur[s].y; acc.z += cur[s].z; acc.w += cur[s].w; }
        if (lane < 32) *reinterpret_cast<float4*>(out + (tile * 32 + lane) * 4) = acc;       // thin output
#pragma unroll
        for (int s = 0; s < 2 * NS; ++s) cur[s] = nxt[s];
    }
}

template <int NS, int MODE>
static void run(const float* g, float* out, long M, const char* name) {
    const int grid = 256 * 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<NS, MODE>), dim3(grid), dim3(256), 0, 0, g, out, M, 0);
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<NS, MODE>), dim3(grid), dim3(256), 0, 0, g, out, M, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-28s N=%3d  %.3f ms  %.0f GB/s\n", name, NS * 16, ms, (double)M * NS * 16 * 4 / ms / 1e6);
}

template <int NB>
static void runcol(const float* g, float* out, long M) {
    const int grid = 256 * 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kcol<NB>), dim3(grid), dim3(256), 0, 0, g, out, M);
    CK(hipEventRecord(a));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((kcol<NB>), dim3(grid), dim3(256), 0, 0, g, out, M);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
    printf("%-28s N=%3d  %.3f ms  %.0f GB/s\n", "reduction-major dwords", NB * 32, ms, (double)M * NB * 32 * 4 / ms / 1e6);
}

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 7929856;
    float *g, *out;
    CK(hipMalloc(&g, (size_t)M * 192 * 4)); CK(hipMalloc(&out, (size_t)M * 4 * 4));
    CK(hipMemset(g, 0, (size_t)M * 192 * 4));
    run<6, 0>(g, out, M, "fragment pattern"); run<6, 1>(g, out, M, "row-linear reference");
    run<9, 0>(g, out, M / 4, "fragment pattern"); run<9, 1>(g, out, M / 4, "row-linear reference");
    run<12, 0>(g, out, M / 16, "fragment pattern"); run<12, 1>(g, out, M / 16, "row-linear reference");
    runcol<3>(g, out, M); runcol<6>(g, out, M / 16); runcol<1>(g, out, M);
    return 0;
}
