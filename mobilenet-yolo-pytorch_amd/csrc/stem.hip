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

template <typename T, int MODE>   // 0: forward (+stats), 1: weight gradient
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   T* __restrict__ y, const T* __restrict__ dy,
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
            st4_stream(y + p * g.Cout + c, o);
            o = stored4<T>(o);
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

// ---- tiled variant: a block owns an 8x32 tile of output pixels of one image; the 17x65x3 input patch is staged in
// LDS with loads that are contiguous along W (the NCHW fast axis), each thread keeps the 27 filter taps of its 4
// output channels in registers and reads inputs as LDS broadcasts (the Cout/4 lanes of a pixel share an address).
// Requires (Cout/4) to divide 256 (Cout = 16, 32, 64 ...).
// LDS row = 72 floats: the patch column `col` (0..64, image column 2*wo0 - 1 + col) sits at index col + 3, so that index 0 is image
// column 64*tw - 4 — 16-byte aligned in the NCHW row — and a row is staged by 18 float4 loads when W % 4 == 0
constexpr int ST_TH = 8, ST_TW = 32, ST_IH = 2 * ST_TH + 1, ST_IW = 2 * ST_TW + 1, ST_IWP = 72, ST_C0 = 3;

template <typename T, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void stem_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        T* __restrict__ y, const T* __restrict__ dy,
                                                        float* __restrict__ parts, StemGeom g, int tiles_h, int tiles_w,
                                                        const T* __restrict__ yraw = nullptr, const float* __restrict__ scale = nullptr,
                                                        const float* __restrict__ shift = nullptr, const float* __restrict__ coef = nullptr, int act = 0) {
    // MODE 2 = MODE 1 with dY rebuilt on load from the unit's output gradient (`dy` argument) and its raw output `yraw`:
    // dY = ca * G * act'(scale*y+shift) + cb * y + cc  (what mny_bn_bwd_apply would have written and this kernel re-read)
    __shared__ float tile[3 * ST_IH * ST_IWP];
    __shared__ float4 red[256 * 2];
    __shared__ float4 cst[MODE == 2 ? 5 * 64 : 1];
    const int tid = threadIdx.x;
    const int cgn = g.cgb;                      // channel groups (divides 256)
    const int cgl = tid % cgn, c = cgl * 4;
    const int ppi = 256 / cgn;                  // pixels handled per pass
    float4 wv[27];
    if (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 27; ++t) wv[t] = make_float4(w[(c + 0) * 27 + t], w[(c + 1) * 27 + t], w[(c + 2) * 27 + t], w[(c + 3) * 27 + t]);
    } else {
#pragma unroll
        for (int t = 0; t < 27; ++t) wv[t] = f4zero();     // accumulators in MODE 1
    }
    if (MODE == 2) {
        if (tid < cgn) {
            cst[tid] = ld4(scale + tid * 4); cst[64 + tid] = ld4(shift + tid * 4); cst[128 + tid] = ld4(coef + tid * 4);
            cst[192 + tid] = ld4(coef + g.Cout + tid * 4); cst[256 + tid] = ld4(coef + 2 * g.Cout + tid * 4);
        }
    }
    float4 s1 = f4zero(), s2 = f4zero();
    const int64_t plane = (int64_t)g.H * g.W;
    const int64_t ntiles = (int64_t)g.N * tiles_h * tiles_w;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tw = (int)(tl % tiles_w), th = (int)((tl / tiles_w) % tiles_h);
        const int64_t n = tl / ((int64_t)tiles_w * tiles_h);
        const int ho0 = th * ST_TH, wo0 = tw * ST_TW;
        const int hi0 = 2 * ho0 - 1, wi0 = 2 * wo0 - 1;
        __syncthreads();                        // previous tile fully consumed
        if ((g.W & 3) == 0) {                   // 4x fewer, 4x wider loads: the staging loop, not the arithmetic, bounds this kernel
            for (int i = tid; i < 3 * ST_IH * (ST_IWP / 4); i += 256) {
                const int q4 = i % (ST_IWP / 4), row = (i / (ST_IWP / 4)) % ST_IH, ci = i / ((ST_IWP / 4) * ST_IH);
                const int hi = hi0 + row, wi = wi0 - ST_C0 + 4 * q4;         // multiple of 4: a float4 is wholly inside or wholly outside the row
                float4 v = f4zero();
                if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) v = ld4(x + (n * 3 + ci) * plane + (int64_t)hi * g.W + wi);
                *reinterpret_cast<float4*>(tile + (ci * ST_IH + row) * ST_IWP + 4 * q4) = v;
            }
        } else {
            for (int i = tid; i < 3 * ST_IH * ST_IW; i += 256) {
                const int col = i % ST_IW, row = (i / ST_IW) % ST_IH, ci = i / (ST_IW * ST_IH);
                const int hi = hi0 + row, wi = wi0 + col;
                tile[(ci * ST_IH + row) * ST_IWP + ST_C0 + col] =
                    (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? x[(n * 3 + ci) * plane + (int64_t)hi * g.W + wi] : 0.f;
            }
        }
        __syncthreads();
        // weight-gradient modes: the (dY, Y) rows of the NEXT pixel pass are requested before this pass's 108 FMAs (one pass of loads in
        // flight per lane while it computes; fetched right where they are used, a lane idled a full round trip per pass)
        float4 d_nx = f4zero(), y_nx = f4zero();
        auto pass_off = [&](int pp, bool& ok) {
            const int pr = pp / ST_TW, pc = pp % ST_TW;
            ok = pp < ST_TH * ST_TW && ho0 + pr < g.Ho && wo0 + pc < g.Wo;
            return ((n * g.Ho + ho0 + pr) * g.Wo + wo0 + pc) * g.Cout + c;
        };
        if (MODE != 0) {
            bool ok;
            const int64_t o0 = pass_off(tid / cgn, ok);
            if (ok) { d_nx = ld4(dy + o0); if (MODE == 2) y_nx = ld4(yraw + o0); }
        }
        for (int pp = tid / cgn; pp < ST_TH * ST_TW; pp += ppi) {
            const int pr = pp / ST_TW, pc = pp % ST_TW;
            const int ho = ho0 + pr, wo = wo0 + pc;
            float4 d_cur = d_nx, y_cur = y_nx;
            if (MODE != 0) {
                bool ok;
                const int64_t o1 = pass_off(pp + ppi, ok);
                if (ok) { d_nx = ld4(dy + o1); if (MODE == 2) y_nx = ld4(yraw + o1); }
            }
            if (ho >= g.Ho || wo >= g.Wo) continue;
            const int64_t o = ((n * g.Ho + ho) * g.Wo + wo) * g.Cout + c;
            const float* tp = tile + (2 * pr) * ST_IWP + ST_C0 + 2 * pc;
            if (MODE == 0) {
                float4 acc = f4zero();
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float v = tp[(ci * ST_IH + kh) * ST_IWP + kw];
                            const float4 ww = wv[ci * 9 + kh * 3 + kw];
                            acc.x = fmaf(v, ww.x, acc.x); acc.y = fmaf(v, ww.y, acc.y); acc.z = fmaf(v, ww.z, acc.z); acc.w = fmaf(v, ww.w, acc.w);
                        }
                st4_stream(y + o, acc);
                acc = stored4<T>(acc);
                add4(s1, acc);
                fma4(s2, acc, acc);
            } else {
                float4 d = d_cur;
                if (MODE == 2) {
                    const float4 yv = y_cur;
                    const float4 sc = cst[cgl], sh = cst[64 + cgl], ca = cst[128 + cgl], cb = cst[192 + cgl], cc = cst[256 + cgl];
                    d.x = fmaf(ca.x, d.x * act_bwd(fmaf(yv.x, sc.x, sh.x), act), fmaf(cb.x, yv.x, cc.x));
                    d.y = fmaf(ca.y, d.y * act_bwd(fmaf(yv.y, sc.y, sh.y), act), fmaf(cb.y, yv.y, cc.y));
                    d.z = fmaf(ca.z, d.z * act_bwd(fmaf(yv.z, sc.z, sh.z), act), fmaf(cb.z, yv.z, cc.z));
                    d.w = fmaf(ca.w, d.w * act_bwd(fmaf(yv.w, sc.w, sh.w), act), fmaf(cb.w, yv.w, cc.w));
                }
#pragma unroll
                for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float v = tp[(ci * ST_IH + kh) * ST_IWP + kw];
                            float4& a = wv[ci * 9 + kh * 3 + kw];
                            a.x = fmaf(v, d.x, a.x); a.y = fmaf(v, d.y, a.y); a.z = fmaf(v, d.z, a.z); a.w = fmaf(v, d.w, a.w);
                        }
            }
        }
    }
    if (!parts) return;
    const int pix = tid / cgn;
    if (MODE == 0) {
        __syncthreads();
        red[tid * 2] = s1; red[tid * 2 + 1] = s2;
        __syncthreads();
        if (pix == 0) {
            float4 a = f4zero(), b = f4zero();
            for (int q = 0; q < ppi; ++q) { add4(a, red[(q * cgn + cgl) * 2]); add4(b, red[(q * cgn + cgl) * 2 + 1]); }
            st4(parts + (int64_t)blockIdx.x * 2 * g.Cout + c, a);
            st4(parts + (int64_t)blockIdx.x * 2 * g.Cout + g.Cout + c, b);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            __syncthreads();
            red[tid] = wv[t];
            __syncthreads();
            if (pix == 0) {
                float4 a = f4zero();
                for (int q = 0; q < ppi; ++q) add4(a, red[q * cgn + cgl]);
                float* dst = parts + (int64_t)blockIdx.x * g.Cout * 27;
                dst[(c + 0) * 27 + t] = a.x; dst[(c + 1) * 27 + t] = a.y; dst[(c + 2) * 27 + t] = a.z; dst[(c + 3) * 27 + t] = a.w;
            }
        }
    }
}

// ---- weight gradient on the matrix cores (Cout = 16 / 32) ---------------------------------------------------------------
// dW[Cout][27] = sum over output pixels of dY[pixel][Cout] (x) patch[pixel][27] is a GEMM whose reduction runs over the PIXELS:
// v_mfma_f32_16x16x4_f32 (exact fp32) takes four pixels per instruction, A = dY^T (16 channels x 4 pixels), B = the pixels' taps
// (4 pixels x 16 of the 27 taps, two instructions per four pixels).  A workgroup owns the same 8 x 32 tile of output pixels as the
// vector-ALU form above and stages the same input patch; dY of the tile (rebuilt from (G, Y, coef) when REB — the arithmetic of
// stem_tile_kernel's MODE 2, value for value) goes through LDS once as [pixel][channel] fp32, written with 16-B global loads of 8
// channels per lane and read back as the A operand (64 consecutive dwords per wave: conflict-free).  Lane (l16, lg) reads tap
// l16 (+16) of pixel lg of the group for B — a lane-constant patch offset.  The vector-ALU form spent 108 FMAs + 27 LDS broadcasts per
// lane and pixel pass on a kernel whose bytes would take a third of its time (stem_tile_kernel<bf16, 2>, 512^2 bs 64: 0.256 ms at
// 2.5 TB/s counted); here a lane's work per tile is two 16-B loads per tensor, the rebuild of 16 values and 48 LDS reads.
typedef float stem_f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct StemRaw8;
template <> struct StemRaw8<float> { float4 a, b; };
template <> struct StemRaw8<bf16_t> { uint4 u; };
__device__ __forceinline__ void stem_ld8(StemRaw8<float>& r, const float* p) { r.a = ld4(p); r.b = ld4(p + 4); }
__device__ __forceinline__ void stem_ld8(StemRaw8<bf16_t>& r, const bf16_t* p) { r.u = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stem_widen8(const StemRaw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void stem_widen8(const StemRaw8<bf16_t>& r, float (&v)[8]) {
    const unsigned u[4] = {r.u.x, r.u.y, r.u.z, r.u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}

template <typename T, int NCT, int REB>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const float* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ yraw,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ coef, int act, float* __restrict__ parts, StemGeom g,
                                                              int tiles_h, int tiles_w) {
    constexpr int CO = 16 * NCT, NH = CO / 8, NPIX = ST_TH * ST_TW;
    static_assert(NPIX == 256, "one (pixel, 8-channel part) item per thread and pass");
    __shared__ __attribute__((aligned(16))) float tile[3 * ST_IH * ST_IWP];
    __shared__ __attribute__((aligned(16))) float dyT[NPIX * CO];          // [pixel][channel]; the end-of-kernel fold reuses it
    __shared__ float cst[REB ? 5 * CO : 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    if (REB) {
        for (int i = tid; i < CO; i += 256) {
            cst[i] = scale[i]; cst[CO + i] = shift[i]; cst[2 * CO + i] = coef[i]; cst[3 * CO + i] = coef[g.Cout + i]; cst[4 * CO + i] = coef[2 * g.Cout + i];
        }
    }
    int toff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = min(16 * h + l16, 26);                   // taps 27..31 of the second half: a valid address, the column is discarded
        const int ci = t / 9, kh = (t % 9) / 3, kw = t % 3;
        toff[h] = (ci * ST_IH + kh) * ST_IWP + kw + ST_C0;
    }
    stem_f32x4 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { acc[ct][0] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; acc[ct][1] = stem_f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int64_t plane = (int64_t)g.H * g.W;
    const int64_t ntiles = (int64_t)g.N * tiles_h * tiles_w;
    constexpr int NX4 = (3 * ST_IH * (ST_IWP / 4) + 255) / 256;           // 16-B patch loads per thread and tile
    const bool w4 = (g.W & 3) == 0;                                        // a float4 of the patch is wholly inside or wholly outside the row
    // Everything a tile reads from HBM is REQUESTED one tile ahead, into registers, right before the previous tile's matrix phase
    // (NX4 + 2 NH 16-B loads per lane in flight under 32 MFMAs per wave), and lands in LDS behind the barrier that retires that phase.
    StemRaw8<T> rg[NH], ry[NH];
    float4 xr[NX4];
    auto origin = [&](int64_t tl, int64_t& n, int& ho0, int& wo0) {       // 32-bit tile arithmetic (the launcher refuses >= 2^31 tiles)
        const unsigned t = (unsigned)tl, r = t / (unsigned)tiles_w;
        n = r / (unsigned)tiles_h;
        ho0 = (int)(r % (unsigned)tiles_h) * ST_TH; wo0 = (int)(t % (unsigned)tiles_w) * ST_TW;
    };
    auto request = [&](int64_t tl) {
        int64_t n; int ho0, wo0;
        origin(tl, n, ho0, wo0);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int item = tid + 256 * i, px = item / NH, part = item % NH;
            const int ho = min(ho0 + px / ST_TW, g.Ho - 1), wo = min(wo0 + px % ST_TW, g.Wo - 1);
            const int64_t o = ((n * g.Ho + ho) * g.Wo + wo) * g.Cout + 8 * part;
            stem_ld8(rg[i], dy + o);
            if (REB) stem_ld8(ry[i], yraw + o);
        }
        if (w4) {
            const int hi0 = 2 * ho0 - 1, wi0 = 2 * wo0 - 1;
#pragma unroll
            for (int k = 0; k < NX4; ++k) {
                const int i = tid + 256 * k;
                const int q4 = i % (ST_IWP / 4), row = (i / (ST_IWP / 4)) % ST_IH, ci = i / ((ST_IWP / 4) * ST_IH);
                const int hi = hi0 + row, wi = wi0 - ST_C0 + 4 * q4;
                xr[k] = f4zero();
                if (i < 3 * ST_IH * (ST_IWP / 4) && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) xr[k] = ld4(x + (n * 3 + ci) * plane + (int64_t)hi * g.W + wi);
            }
        }
    };
    request(blockIdx.x);
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        int64_t n; int ho0, wo0;
        origin(tl, n, ho0, wo0);
        __syncthreads();                        // previous tile fully consumed
        if (w4) {
#pragma unroll
            for (int k = 0; k < NX4; ++k) {
                const int i = tid + 256 * k;
                const int q4 = i % (ST_IWP / 4), row = (i / (ST_IWP / 4)) % ST_IH, ci = i / ((ST_IWP / 4) * ST_IH);
                if (i < 3 * ST_IH * (ST_IWP / 4)) *reinterpret_cast<float4*>(tile + (ci * ST_IH + row) * ST_IWP + 4 * q4) = xr[k];
            }
        } else {
            const int hi0 = 2 * ho0 - 1, wi0 = 2 * wo0 - 1;
            for (int i = tid; i < 3 * ST_IH * ST_IW; i += 256) {
                const int col = i % ST_IW, row = (i / ST_IW) % ST_IH, ci = i / (ST_IW * ST_IH);
                const int hi = hi0 + row, wi = wi0 + col;
                tile[(ci * ST_IH + row) * ST_IWP + ST_C0 + col] =
                    (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? x[(n * 3 + ci) * plane + (int64_t)hi * g.W + wi] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int item = tid + 256 * i, px = item / NH, part = item % NH;
            const bool ok = ho0 + px / ST_TW < g.Ho && wo0 + px % ST_TW < g.Wo;
            float d[8];
            stem_widen8(rg[i], d);
            if (REB) {
                float yv[8];
                stem_widen8(ry[i], yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 8 * part + e;
                    d[e] = fmaf(cst[2 * CO + c], d[e] * act_bwd(fmaf(yv[e], cst[c], cst[CO + c]), act), fmaf(cst[3 * CO + c], yv[e], cst[4 * CO + c]));
                }
            }
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = 0.f;
            }
            float* q = dyT + px * CO + 8 * part;
            *reinterpret_cast<float4*>(q) = make_float4(d[0], d[1], d[2], d[3]);
            *reinterpret_cast<float4*>(q + 4) = make_float4(d[4], d[5], d[6], d[7]);
        }
        __syncthreads();
        if (tl + gridDim.x < ntiles) request(tl + gridDim.x);
#pragma unroll 4
        for (int gi = 0; gi < 16; ++gi) {
            const int px = 64 * wave + 4 * gi + lg;
            const float* tp = tile + (2 * (px >> 5)) * ST_IWP + 2 * (px & 31);
            const float b0 = tp[toff[0]], b1 = tp[toff[1]];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float a = dyT[px * CO + 16 * ct + l16];
                acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[ct][0], 0, 0, 0);
                acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[ct][1], 0, 0, 0);
            }
        }
    }
    // fold the four waves' accumulators in wave order; lane (l16, lg) holds dW[channel 16 ct + 4 lg + v][tap 16 h + l16]
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int v = 0; v < 4; ++v) dyT[((wave * NCT + ct) * 2 + h) * 256 + lane * 4 + v] = acc[ct][h][v];
    __syncthreads();
    float* dst = parts + (int64_t)blockIdx.x * g.Cout * 27;
    for (int e = tid; e < NCT * 2 * 256; e += 256) {
        const int v = e & 3, ln = (e >> 2) & 63, h = (e >> 8) & 1, ct = e >> 9;
        const int ch = 16 * ct + 4 * (ln >> 4) + v, tap = 16 * h + (ln & 15);
        if (tap < 27) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += dyT[((w * NCT + ct) * 2 + h) * 256 + ln * 4 + v];
            dst[ch * 27 + tap] = a;
        }
    }
}

static bool stem_mfma_ok(int Cout) {
    static const bool off = getenv("MNY_STEM_WGRAD_VALU") && getenv("MNY_STEM_WGRAD_VALU")[0] == '1';
    return !off && (Cout == 16 || Cout == 32);
}
static int stem_mfma_grid(const StemGeom& g) {
    static const int env = getenv("MNY_STEM_WG_GRID") ? atoi(getenv("MNY_STEM_WG_GRID")) : 0;
    const int64_t want = (int64_t)g.N * cdiv(g.Ho, ST_TH) * cdiv(g.Wo, ST_TW);
    const int cap = env > 0 ? env : (g.Cout == 16 ? 768 : 512);      // 148-160 VGPRs at 16 channels (three resident workgroups per CU), 176-200 at 32 (two)
    return (int)(want < cap ? want : cap);
}
template <typename T, int REB>
static int stem_mfma_launch(const float* x, const T* dy, const T* yraw, const float* scale, const float* shift, const float* coef, int act, float* ws,
                            const StemGeom& g, hipStream_t st) {
    const int gx = stem_mfma_grid(g), th = (int)cdiv(g.Ho, ST_TH), tw = (int)cdiv(g.Wo, ST_TW);
    MNY_REQUIRE((int64_t)g.N * th * tw < ((int64_t)1 << 31), "stem_wgrad: too many tiles");
    if (g.Cout == 16) hipLaunchKernelGGL((stem_wgrad_mfma_kernel<T, 1, REB>), dim3(gx), dim3(256), 0, st, x, dy, yraw, scale, shift, coef, act, ws, g, th, tw);
    else hipLaunchKernelGGL((stem_wgrad_mfma_kernel<T, 2, REB>), dim3(gx), dim3(256), 0, st, x, dy, yraw, scale, shift, coef, act, ws, g, th, tw);
    return check_launch("stem_wgrad_mfma_kernel");
}

static bool stem_tiled_ok(int Cout) { const int cgn = Cout / 4; return cgn > 0 && 256 % cgn == 0; }

static int stem_geom(StemGeom& g, int& gx, int N, int H, int W, int Cout) {
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "stem: empty tensor");
    MNY_REQUIRE(Cout % 4 == 0 && Cout >= 4 && Cout <= 256, "stem: Cout=%d must be a multiple of 4 in [4,256]", Cout);
    g.N = N; g.H = H; g.W = W; g.Cout = Cout;
    g.Ho = (H + 2 - 3) / 2 + 1; g.Wo = (W + 2 - 3) / 2 + 1;
    g.cgb = Cout / 4; g.ppb = 256 / g.cgb;
    g.npix = (int64_t)N * g.Ho * g.Wo;
    int64_t want = stem_tiled_ok(Cout) ? (int64_t)N * cdiv(g.Ho, ST_TH) * cdiv(g.Wo, ST_TW) : cdiv(g.npix, g.ppb);
    const int64_t cap = 768;            // 160 VGPRs -> 3 resident workgroups per CU: one full wave of blocks, no tail
    gx = (int)(want < cap ? want : cap);
    return MNY_OK;
}

}  // namespace mny

using namespace mny;

extern "C" int mny_stem_stat_parts(int N, int H, int W, int Cout) {
    StemGeom g; int gx;
    if (stem_geom(g, gx, N, H, W, Cout)) return MNY_EINVAL;
    return gx;
}
extern "C" int mny_stem_wgrad_parts(int N, int H, int W, int Cout) {
    StemGeom g; int gx;
    if (stem_geom(g, gx, N, H, W, Cout)) return MNY_EINVAL;
    return stem_mfma_ok(Cout) ? stem_mfma_grid(g) : gx;
}

template <typename T>
static int stem_fwd_impl(const float* x_nchw, const float* w, T* y, float* stats, int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(x_nchw && w && y, "stem_fwd: null pointer");
    StemGeom g; int gx;
    int rc = stem_geom(g, gx, N, H, W, Cout);
    if (rc) return rc;
    if (stem_tiled_ok(Cout)) {
        hipLaunchKernelGGL((stem_tile_kernel<T, 0>), dim3(gx), dim3(256), 0, (hipStream_t)stream, x_nchw, w, y, (const T*)nullptr, stats, g,
                           (int)cdiv(g.Ho, ST_TH), (int)cdiv(g.Wo, ST_TW));
        return check_launch("stem_tile_kernel<fwd>");
    }
    hipLaunchKernelGGL((stem_kernel<T, 0>), dim3(gx), dim3(g.cgb * g.ppb), 0, (hipStream_t)stream, x_nchw, w, y, (const T*)nullptr, stats, g);
    return check_launch("stem_kernel<fwd>");
}
extern "C" int mny_stem_fwd(const float* x_nchw, const float* w, float* y, float* stats, int N, int H, int W, int Cout, void* stream) {
    return stem_fwd_impl<float>(x_nchw, w, y, stats, N, H, W, Cout, stream);
}
extern "C" int mny_stem_fwd_bf16(const float* x_nchw, const float* w, void* y, float* stats, int N, int H, int W, int Cout, void* stream) {
    return stem_fwd_impl<bf16_t>(x_nchw, w, (bf16_t*)y, stats, N, H, W, Cout, stream);
}

template <typename T>
static int stem_wgrad_impl(const float* x_nchw, const T* dy, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(x_nchw && dy && ws, "stem_wgrad: null pointer");
    StemGeom g; int gx;
    int rc = stem_geom(g, gx, N, H, W, Cout);
    if (rc) return rc;
    if (stem_mfma_ok(Cout)) {
        gx = stem_mfma_grid(g);
        rc = stem_mfma_launch<T, 0>(x_nchw, dy, (const T*)nullptr, nullptr, nullptr, nullptr, 0, ws, g, (hipStream_t)stream);
        if (rc || !dw) return rc;
        return launch_reduce_parts(ws, gx, Cout * 27, dw, (hipStream_t)stream);
    }
    if (stem_tiled_ok(Cout))
        hipLaunchKernelGGL((stem_tile_kernel<T, 1>), dim3(gx), dim3(256), 0, (hipStream_t)stream, x_nchw, (const float*)nullptr, (T*)nullptr, dy, ws, g,
                           (int)cdiv(g.Ho, ST_TH), (int)cdiv(g.Wo, ST_TW));
    else
        hipLaunchKernelGGL((stem_kernel<T, 1>), dim3(gx), dim3(g.cgb * g.ppb), 0, (hipStream_t)stream, x_nchw, (const float*)nullptr, (T*)nullptr, dy, ws, g);
    rc = check_launch("stem_kernel<wgrad>");
    if (rc || !dw) return rc;                       // dw == NULL: partials only (combined later by mny_reduce_batch)
    return launch_reduce_parts(ws, gx, Cout * 27, dw, (hipStream_t)stream);
}
// weight gradient of the stem conv straight from the unit's OUTPUT gradient: BN-backward-apply + activation backward are redone on
// load (coef from mny_bn_bwd_finalize), so the dY tensor is neither written nor re-read
extern "C" int mny_stem_bnwgrad_supported(int Cout) { return stem_tiled_ok(Cout) && Cout <= 256 ? 1 : 0; }
template <typename T>
static int stem_bnwgrad_impl(const float* x_nchw, const T* gout, const T* y, const float* scale, const float* shift, int act, const float* coef,
                             float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(x_nchw && gout && y && scale && shift && coef && ws, "stem_bnwgrad: null pointer");
    MNY_REQUIRE(mny_stem_bnwgrad_supported(Cout) == 1, "stem_bnwgrad: Cout=%d unsupported", Cout);
    StemGeom g; int gx;
    int rc = stem_geom(g, gx, N, H, W, Cout);
    if (rc) return rc;
    if (stem_mfma_ok(Cout)) {
        gx = stem_mfma_grid(g);
        rc = stem_mfma_launch<T, 1>(x_nchw, gout, y, scale, shift, coef, act, ws, g, (hipStream_t)stream);
        if (rc || !dw) return rc;
        return launch_reduce_parts(ws, gx, Cout * 27, dw, (hipStream_t)stream);
    }
    hipLaunchKernelGGL((stem_tile_kernel<T, 2>), dim3(gx), dim3(256), 0, (hipStream_t)stream, x_nchw, (const float*)nullptr, (T*)nullptr, gout, ws, g,
                       (int)cdiv(g.Ho, ST_TH), (int)cdiv(g.Wo, ST_TW), y, scale, shift, coef, act);
    rc = check_launch("stem_tile_kernel<bn-wgrad>");
    if (rc || !dw) return rc;
    return launch_reduce_parts(ws, gx, Cout * 27, dw, (hipStream_t)stream);
}
extern "C" int mny_stem_bnwgrad(const float* x_nchw, const float* g, const float* y, const float* scale, const float* shift, int act,
                                const float* coef, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    return stem_bnwgrad_impl<float>(x_nchw, g, y, scale, shift, act, coef, dw, ws, N, H, W, Cout, stream);
}
extern "C" int mny_stem_bnwgrad_bf16(const float* x_nchw, const void* g, const void* y, const float* scale, const float* shift, int act,
                                     const float* coef, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    return stem_bnwgrad_impl<bf16_t>(x_nchw, (const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, dw, ws, N, H, W, Cout, stream);
}

extern "C" int mny_stem_wgrad(const float* x_nchw, const float* dy, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    return stem_wgrad_impl<float>(x_nchw, dy, dw, ws, N, H, W, Cout, stream);
}
extern "C" int mny_stem_wgrad_bf16(const float* x_nchw, const void* dy, float* dw, float* ws, int N, int H, int W, int Cout, void* stream) {
    return stem_wgrad_impl<bf16_t>(x_nchw, (const bf16_t*)dy, dw, ws, N, H, W, Cout, stream);
}
