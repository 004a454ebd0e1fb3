// Stem: 3x3 stride-2 pad-1 convolution 3 -> Cout, NCHW fp32 input, NHWC fp32 output (+BN statistics),
// and its weight gradient.  HBM-bound (AI 9.8 FLOP/B): 1.0 GB of the 1.4 GB traffic at bs=256 is the
// output store, so the kernel is organised around fully coalesced float4 NHWC stores; the 27 input taps
// of a pixel are shared by the Cout/4 lanes that own it (L1 broadcast).  The input image needs no
// gradient (it is data), so there is no backward-data kernel.
//
// replaces conv_3x3_bn's nn.Conv2d(3,32,3,2,1) at models/mobilenetv2.py:40 (used :113) and
// nn.Conv2d(3,16,3,2,1) at models/mobilenetv3.py:80.
#include "common.h"

namespace mny {

struct StemGeom { int N, H, W, Ho, Wo, Cout, cgb, ppb; int64_t npix; };

template <int MODE>   // 0: forward (+stats), 1: weight gradient
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   float* __restrict__ y, const float* __restrict__ dy,
                                                   float* __restrict__ parts, StemGeom g) {
    __shared__ float4 wl[27 * 64];      // [tap][cg] (Cout <= 256)
    __shared__ float4 red[256 * 2];
    const int tid = threadIdx.x;
    const int cgl = tid % g.cgb, pix = tid / g.cgb;
    const int c = cgl * 4;
    if (MODE == 0) {
        for (int i = tid; i < 27 * g.cgb; i += blockDim.x) {
            const int tap = i / g.cgb, q = i % g.cgb;   // w[co][ci][kh][kw] -> tap = ci*9+kh*3+kw
            wl[tap * g.cgb + q] = make_float4(w[(q * 4 + 0) * 27 + tap], w[(q * 4 + 1) * 27 + tap],
                                              w[(q * 4 + 2) * 27 + tap], w[(q * 4 + 3) * 27 + tap]);
        }
        __syncthreads();
    }
    float4 s1 = f4zero(), s2 = f4zero();
    float4 wacc[27];
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 27; ++t) wacc[t] = f4zero();
    }
    const int64_t plane = (int64_t)g.H * g.W;
    for (int64_t p = (int64_t)blockIdx.x * g.ppb + pix; p < g.npix; p += (int64_t)gridDim.x * g.ppb) {
        const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho);
        const int64_t n = p / ((int64_t)g.Wo * g.Ho);
        const float* xn = x + n * 3 * plane;
        float in[27];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
                    in[ci * 9 + kh * 3 + kw] = (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? xn[ci * plane + (int64_t)hi * g.W + wi] : 0.f;
                }
        if (MODE == 0) {
            float4 o = f4zero();
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const float4 wv = wl[t * g.cgb + cgl];
                o.x = fmaf(in[t], wv.x, o.x); o.y = fmaf(in[t], wv.y, o.y);
                o.z = fmaf(in[t], wv.z, o.z); o.w = fmaf(in[t], wv.w, o.w);
            }
            st4(y + p * g.Cout + c, o);
            add4(s1, o);
            fma4(s2, o, o);
        } else {
            const float4 d = ld4(dy + p * g.Cout + c);
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                wacc[t].x = fmaf(in[t], d.x, wacc[t].x); wacc[t].y = fmaf(in[t], d.y, wacc[t].y);
                wacc[t].z = fmaf(in[t], d.z, wacc[t].z); wacc[t].w = fmaf(in[t], d.w, wacc[t].w);
            }
        }
    }
    if (!parts) return;
    if (MODE == 0) {
        red[tid * 2] = s1; red[tid * 2 + 1] = s2;
        __syncthreads();
        if (pix == 0) {
            float4 a = f4zero(), b = f4zero();
            for (int q = 0; q < g.ppb; ++q) { add4(a, red[(q * g.cgb + cgl) * 2]); add4(b, red[(q * g.cgb + cgl) * 2 + 1]); }
            st4(parts + (int64_t)blockIdx.x * 2 * g.Cout + c, a);
            st4(parts + (int64_t)blockIdx.x * 2 * g.Cout + g.Cout + c, b);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            __syncthreads();
            red[tid] = wacc[t];
            __syncthreads();
            if (pix == 0) {
                float4 a = f4zero();
                for (int q = 0; q < g.ppb; ++q) add4(a, red[q * g.cgb + cgl]);
                float* dst = parts + (int64_t)blockIdx.x * g.Cout * 27;
                dst[(c + 0) * 27 + t] = a.x; dst[(c + 1) * 27 + t] = a.y;
                dst[(c + 2) * 27 + t] = a.z; dst[(c + 3) * 27 + t] = a.w;
            }
        }
    }
}

static int stem_geom(StemGeom& g, int& gx, int N, int H, int W, int Cout) {
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "stem: empty tensor");
    MNY_REQUIRE(Cout % 4 == 0 && Cout >= 4 && Cout <= 256, "stem: Cout=%d must be a multiple of 4 in [4,256]", Cout);
    g.N = N; g.H = H; g.W = W; g.Cout = Cout;
    g.Ho = (H + 2 - 3) / 2 + 1; g.Wo = (W + 2 - 3) / 2 + 1;
    g.cgb = Cout / 4; g.ppb = 256 / g.cgb;
    g.npix = (int64_t)N * g.Ho * g.Wo;
    int64_t want = cdiv(g.npix, g.ppb);
    gx = (int)(want < kMaxParts ? want : kMaxParts);
    return MNY_OK;
}

}  // namespace mny

using namespace mny;

extern "C" int mny_stem_stat_parts(int N, int H, int W, int Cout) {
    StemGeom g; int gx;
    if (stem_geom(g, gx, N, H, W, Cout)) return MNY_EINVAL;
    return gx;
}
extern "C" int mny_stem_wgrad_parts(int N, int H, int W, int Cout) { return mny_stem_stat_parts(N, H, W, Cout); }

extern "C" int mny_stem_fwd(const float* x_nchw, const float* w, float* y, float* stats, int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(x_nchw && w && y, "stem_fwd: null pointer");
    StemGeom g; int gx;
    int rc = stem_geom(g, gx, N, H, W, Cout);
    if (rc) return rc;
    hipLaunchKernelGGL((stem_kernel<0>), dim3(gx), dim3(g.cgb * g.ppb), 0, (hipStream_t)stream, x_nchw, w, y, nullptr, stats, g);
    return check_launch("stem_kernel<fwd>");
}

extern "C" int mny_stem_wgrad(const float* x_nchw, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(x_nchw && dy && dw && ws, "stem_wgrad: null pointer");
    StemGeom g; int gx;
    int rc = stem_geom(g, gx, N, H, W, Cout);
    if (rc) return rc;
    hipLaunchKernelGGL((stem_kernel<1>), dim3(gx), dim3(g.cgb * g.ppb), 0, (hipStream_t)stream, x_nchw, nullptr, nullptr, dy, ws, g);
    rc = check_launch("stem_kernel<wgrad>");
    if (rc) return rc;
    return launch_reduce_parts(ws, gx, Cout * 27, dw, (hipStream_t)stream);
}
