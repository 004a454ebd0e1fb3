// Drivable-area segmentation loss of the BDD100K config (include/mnyolo.h: mny_seg_loss, mny_seg_sigmoid).
// Replaces models/seg_loss.py:51-80 and its autograd: sigmoid + MSE * 0.05 over the channels-last seg head, the two
// monitoring means, and dL/dhead — the reference's custom sigmoid hands the gradient through unchanged (:24-32), so
// dhead = 0.05 * 2 * (sigmoid(x) - t) / numel with NO sigma' factor.  One streaming pass + a fixed-order fp64 finalize.
#include "common.h"

namespace mny {
namespace {

constexpr int kSegThreads = 256;
constexpr int kSegMaxBlocks = 1024;

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }   // seg_loss.py:19

__global__ __launch_bounds__(kSegThreads) void seg_loss_kernel(const float* __restrict__ head, const float* __restrict__ truth, int64_t n, float gscale,
                                                               float* __restrict__ dhead, double* __restrict__ parts) {
    double sq = 0, so = 0, sn = 0, co = 0, cn = 0;
    for (int64_t i = (int64_t)blockIdx.x * kSegThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kSegThreads) {
        const float s = sigmoid_ref(head[i]), t = truth[i];
        const float d = s - t;
        dhead[i] = gscale * d;
        sq += (double)(d * d);                                                  // :42 (input - target)**2 in fp32
        if (t >= 0.5f) { so += s; co += 1; } else if (t < 0.5f) { sn += s; cn += 1; }   // :67-68 (a NaN truth joins neither)
    }
    __shared__ double red[kSegThreads / kWave][5];
    double v[5] = {sq, so, sn, co, cn};
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (int o = 32; o; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 5) {
        double s = 0;
        for (int w = 0; w < kSegThreads / kWave; ++w) s += red[w][threadIdx.x];
        parts[(size_t)blockIdx.x * 5 + threadIdx.x] = s;
    }
}

__global__ void seg_finalize_kernel(const double* __restrict__ parts, int blocks, int64_t n, float* __restrict__ out3) {
    __shared__ double tot[5];
    if (threadIdx.x < 5) {
        double s = 0;
        for (int b = 0; b < blocks; ++b) s += parts[(size_t)b * 5 + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out3[0] = (float)(tot[0] / (double)n * 0.05);                           // :43-46 weights = 1, total = numel; :76 * 0.05
        out3[1] = (float)(tot[1] / tot[3]);                                     // torch.mean(obj): 0/0 = NaN like torch
        out3[2] = (float)(tot[2] / tot[4]);
    }
}

__global__ void seg_sigmoid_kernel(const float* __restrict__ head, int h, int w, int C, float* __restrict__ out) {   // :78-79
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // over [h][w][C] of image 0
    if (i >= h * w * C) return;
    const int c = i % C, p = i / C;
    out[(size_t)c * h * w + p] = sigmoid_ref(head[i]);
}

int seg_blocks(int64_t n) { return (int)(cdiv(n, kSegThreads) < kSegMaxBlocks ? cdiv(n, kSegThreads) : kSegMaxBlocks); }

}  // namespace
}  // namespace mny

using namespace mny;

extern "C" size_t mny_seg_loss_ws_bytes(int64_t n) { return n > 0 ? (size_t)seg_blocks(n) * 5 * sizeof(double) : 0; }

extern "C" int mny_seg_loss(const float* head, const float* seg_maps, int64_t n, float* out3, float* dhead, void* ws, void* stream) {
    MNY_REQUIRE(head && seg_maps && out3 && dhead && ws, "mny_seg_loss: null pointer");
    MNY_REQUIRE(n > 0, "mny_seg_loss: empty head (n=%lld)", (long long)n);
    hipStream_t st = (hipStream_t)stream;
    const int blocks = seg_blocks(n);
    const float gscale = (float)(0.05 * 2.0 / (double)n);
    seg_loss_kernel<<<blocks, kSegThreads, 0, st>>>(head, seg_maps, n, gscale, dhead, (double*)ws);
    seg_finalize_kernel<<<1, 64, 0, st>>>((const double*)ws, blocks, n, out3);
    return check_launch("mny_seg_loss");
}

extern "C" int mny_seg_sigmoid(const float* head, int h, int w, int C, float* out, void* stream) {
    MNY_REQUIRE(head && out, "mny_seg_sigmoid: null pointer");
    MNY_REQUIRE(h > 0 && w > 0 && C > 0, "mny_seg_sigmoid: bad shape %dx%dx%d", h, w, C);
    seg_sigmoid_kernel<<<(int)cdiv((int64_t)h * w * C, 256), 256, 0, (hipStream_t)stream>>>(head, h, w, C, out);
    return check_launch("mny_seg_sigmoid");
}
