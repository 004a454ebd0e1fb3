// Fused backward of a depthwise 3x3 stride-1 conv+BN+activation unit.
//
// Unfused, the unit's backward touches its (hidden-width) tensors 7 times after the BN reduction:
//     bn_bwd_apply : read G, Y -> write dY          dw_bwd_weight : read dY, X          dw_bwd_data : read dY -> write dX
// Here one kernel reads G, Y, X once and writes dX once (4 passes): dY = ca*G*act'(sc*Y+sh) + cb*Y + cc is rebuilt in
// registers from (G, Y) and the per-channel coefficients of mny_bn_bwd_finalize, and both gradients are taken from it.
//
// Thread = 4 channels x one column, walking DOWN a strip of rows with row-sized state only:
//   * data gradient as a scatter: dY row r (columns w-1..w+1) updates the three partial output rows r-1, r, r+1 of column
//     w; row r-1 is complete after that and is stored (3 float4 of state instead of a 3x3 window of dY);
//   * weight gradient with the input row delayed by one step: input row r-1 meets the centre-column dY of rows r, r-1, r-2
//     (2 float4 of history instead of a 3x3 window of X).
// Halo rows/columns are re-read by neighbouring threads/strips from L2; strips are up to 32 rows (35 loaded rows per 32).
// All loads are unconditional with clamped addresses and zeroed by select (no control flow in the row loop).
//
// replaces, for these units, the autograd backward of nn.Conv2d(groups=C) + nn.BatchNorm2d + ReLU6 / LeakyReLU
// (models/mobilenetv2.py:65-67,79-81, models/mbv2_yolo.py:22-24).
#include <stdlib.h>

#include "common.h"

namespace mny {

struct DwbGeom {
    int N, H, W, C;
    int TH, nHS;
    int64_t nstrips;
    int cg_total, cgb;
    int xcd;            // XCD-contiguous strip order (workgroup b runs on XCD b % 8): column / row halos shared with the neighbours hit the same L2
    int dz;             // RED only: store  in_scale o dX o act_P'(z)  instead of dX (mny_dw_bnbwd_red_dz: the producer's low-rank BN backward, csrc/lrbwd.hip)
};

// AM: activation of THIS unit: 0 = none (act' = 1), 1 = min(max(z, slope z), hi) family, 2 = hswish
// XF: view of the input: 0 = as is, 1 = scale/shift + ReLU6, 2 = scale/shift + hswish, 3 = scale/shift + max(z, slope z) (leaky / relu / none)
// RED: the input X is itself the raw output of a conv+BN+act unit P whose ONLY consumer is this depthwise unit, so the dX
// written here is P's complete output gradient: the kernel also leaves P's BN-backward sums (sum dz, sum dz * xhat per channel,
// dz = dX * act_P'(in_scale * x + in_shift), xhat = (x - in_mean) * in_invstd) as per-block partial rows [gridDim.x][2][C] —
// what a separate mny_bn_bwd_reduce pass over (dX, X) would produce, without re-reading either.  The accumulators and the
// parked raw centre input live in LDS (the kernel sits at the 168-VGPR limit of 3 waves per SIMD).
template <typename T, int AM, int XF, bool RED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void dw_bnbwd_s1k3_kernel(
    const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    const float* __restrict__ coef, const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int in_act, const float* __restrict__ w, const T* __restrict__ addend, T* __restrict__ dx, float* __restrict__ parts,
    const float* __restrict__ in_mean, const float* __restrict__ in_invstd, float* __restrict__ in_red, DwbGeom gm) {
    __shared__ float4 red[256];
    // per-channel constants (9 filter taps + 7 coefficient vectors [+ mean, invstd of the input's unit]) live in LDS,
    // [18][cgb] float4, read where used: keeping them in registers put the kernel at ~200 VGPRs = 2 waves/SIMD
    extern __shared__ __attribute__((aligned(16))) float4 cst[];
    float4* xcs = cst + 18 * gm.cgb;                     // RED: [256] raw centre input of the previous row, [256] s1, [256] s2
    const int tid = threadIdx.x;
    const int cgl = tid % gm.cgb, pix = tid / gm.cgb, ppb = blockDim.x / gm.cgb;
    const int cg = blockIdx.y * gm.cgb + cgl;
    const bool cvalid = cg < gm.cg_total;
    const int c = cg * 4;

    float4 wacc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wacc[t] = f4zero();

    if (pix == 0 && cvalid) {
#pragma unroll
        for (int t = 0; t < 9; ++t) cst[t * gm.cgb + cgl] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
        cst[9 * gm.cgb + cgl] = ld4(scale + c);
        cst[10 * gm.cgb + cgl] = ld4(shift + c);
        cst[11 * gm.cgb + cgl] = ld4(coef + c);
        cst[12 * gm.cgb + cgl] = ld4(coef + gm.C + c);
        cst[13 * gm.cgb + cgl] = ld4(coef + 2 * gm.C + c);
        cst[14 * gm.cgb + cgl] = (XF != 0 && in_scale) ? ld4(in_scale + c) : f4one();
        cst[15 * gm.cgb + cgl] = (XF != 0 && in_scale) ? ld4(in_shift + c) : f4zero();
        if (RED) { cst[16 * gm.cgb + cgl] = ld4(in_mean + c); cst[17 * gm.cgb + cgl] = ld4(in_invstd + c); }
    }
    if (RED) { xcs[256 + tid] = f4zero(); xcs[512 + tid] = f4zero(); }
    __syncthreads();
    if (cvalid) {
#define WG(t) my[(t) * gm.cgb]
        const float slope = act_slope(act), hi = act_hi(act);
        const float xslope = act_slope(in_act);

        // dY = ca * G * act'(sc*Y + sh) + cb * Y + cc on one register pair (packed fma; the activation derivative is a select)
        // act'(z) as selects between constants (v_cndmask; a select between computed values turned into branches)
        auto dact = [&](float z) {
            if (AM == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
            return (z > 0.f ? 1.f : slope) * (z < hi ? 1.f : 0.f);
        };
        auto dy2 = [&](v2f gv, v2f yv, v2f s, v2f h, v2f a, v2f b, v2f cterm) {
            v2f d = gv;
            if (AM != 0) { const v2f z = __builtin_elementwise_fma(yv, s, h); d = gv * v2f{dact(z.x), dact(z.y)}; }
            return __builtin_elementwise_fma(a, d, __builtin_elementwise_fma(b, yv, cterm));
        };
        auto xf2 = [&](v2f v, v2f s, v2f h) {
            if (XF == 0) return v;
            const v2f z = __builtin_elementwise_fma(v, s, h);
            if (XF == 1) return v2f{__builtin_amdgcn_fmed3f(z.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.y, 0.f, 6.f)};      // ReLU6
            if (XF == 3) {                                                   // leaky / relu / scale-only: max(z, slope z), no upper clip
                const v2f t = z * v2f{xslope, xslope};
                return v2f{fmaxf(z.x, t.x), fmaxf(z.y, t.y)};
            }
            const v2f t = z + v2f{3.f, 3.f};
            return z * v2f{__builtin_amdgcn_fmed3f(t.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(t.y, 0.f, 6.f)} * v2f{1.f / 6.f, 1.f / 6.f};
        };

        const int gxd = gridDim.x;
        const int lb = (gm.xcd && (gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        F4P wp[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wp[t] = f4p(wacc[t]);
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < gm.nstrips; strip += (int64_t)gxd * ppb) {
            const int wo = (int)(strip % gm.W);
            const int hs = (int)((strip / gm.W) % gm.nHS);
            const int n = (int)(strip / ((int64_t)gm.W * gm.nHS));
            const int h0 = hs * gm.TH;
            const int h1 = min(h0 + gm.TH, gm.H);
            const int64_t img = (int64_t)n * gm.H * gm.W * gm.C + c;
            int col[3];
            float cokf[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int wi = wo - 1 + q; cokf[q] = (wi >= 0 && wi < gm.W) ? 1.f : 0.f; col[q] = min(max(wi, 0), gm.W - 1); }

            F4P P0 = f4p0(), P1 = f4p0();
            // weight gradient, round 3: dW[kr][kq] += a[i][w] * dY[i - kr + 1][w - kq + 1] summed over the INPUT pixels (i, w) this strip
            // owns — the activated input is needed at the thread's own column only (one load and one transform per row instead of
            // three: 7 instead of 9 tensor loads per row), the three columns of dY it meets are the ones the data gradient already
            // rebuilds.  State: dY of rows r-1 and r-2 (three columns each) + the input of row r-1.
            F4P dyp1[3] = {f4p0(), f4p0(), f4p0()}, dyp2[3] = {f4p0(), f4p0(), f4p0()};
            F4P aprev = f4p0();
            for (int r = h0 - 1; r <= h1; ++r) {
                int lo = cgl;
                asm volatile("" : "+v"(lo));                       // opaque per iteration: keeps the LDS constant reads IN the loop
                const float4* my = cst + lo;                      // (hoisted, they are 64 VGPRs live across it)
                const int rc = min(max(r, 0), gm.H - 1);
                const float rokf = (r >= 0 && r < gm.H) ? 1.f : 0.f;
                const int64_t rowoff = img + (int64_t)rc * gm.W * gm.C;
                float4 gv[3], yv[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int64_t o = rowoff + (int64_t)col[q] * gm.C;
                    gv[q] = ld4(g + o); yv[q] = ld4(y + o);
                }
                const float4 xc4 = ld4(x + rowoff + (int64_t)col[1] * gm.C);
                F4P dyr[3];
                const F4P sc = f4p(my[9 * gm.cgb]), sh = f4p(my[10 * gm.cgb]), ca = f4p(my[11 * gm.cgb]), cb = f4p(my[12 * gm.cgb]), cc = f4p(my[13 * gm.cgb]);
                const F4P xsc = f4p(my[14 * gm.cgb]), xsh = f4p(my[15 * gm.cgb]);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float m = rokf * cokf[q];                // dY is 0 outside the image
                    const v2f m2 = v2f{m, m};
                    const F4P G4 = f4p(gv[q]), Y4 = f4p(yv[q]);
                    dyr[q].lo = dy2(G4.lo, Y4.lo, sc.lo, sh.lo, ca.lo, cb.lo, cc.lo) * m2;
                    dyr[q].hi = dy2(G4.hi, Y4.hi, sc.hi, sh.hi, ca.hi, cb.hi, cc.hi) * m2;
                }
                // activated input of row r at the thread's own column; it contributes (as row i = r, one step later) only if the strip owns it
                F4P ar;
                {
                    const float own = (r >= h0 && r < h1) ? 1.f : 0.f;
                    const F4P X4 = f4p(xc4);
                    ar.lo = xf2(X4.lo, xsc.lo, xsh.lo) * v2f{own, own};
                    ar.hi = xf2(X4.hi, xsc.hi, xsh.hi) * v2f{own, own};
                }
                // data gradient: dX[i][w] += sum_kq dY[r][w+1-kq] * wt[i-r+1][kq]  (dyr[q] sits at column w-1+q -> q = 2-kq)
                F4P P2 = f4p0();
#pragma unroll
                for (int kq = 0; kq < 3; ++kq) {
                    pfma(P0, dyr[2 - kq], f4p(WG(0 + kq)));
                    pfma(P1, dyr[2 - kq], f4p(WG(3 + kq)));
                    pfma(P2, dyr[2 - kq], f4p(WG(6 + kq)));
                }
                if (r - 1 >= h0) {                                  // row r-1 is complete (r-1 < h1 by the loop bound)
                    const int64_t o = img + ((int64_t)(r - 1) * gm.W + wo) * gm.C;
                    float4 out = f4u(P0);
                    if (addend) add4(out, ld4(addend + o));
                    if (!RED || !gm.dz) st4_stream(dx + o, out);
                    if (RED) {                                      // BN-backward sums of the unit that produced X (raw x of row r-1 was parked below)
                        const float4 xc = xcs[tid], mu = my[16 * gm.cgb], is = my[17 * gm.cgb];
                        const float4 gq = stored4<T>(out);
                        auto pact = [&](float xv1, float s, float h) {
                            const float z = fmaf(xv1, s, h);
                            if (XF == 1) return (z > 0.f ? 1.f : 0.f) * (z < 6.f ? 1.f : 0.f);
                            if (XF == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
                            return z > 0.f ? 1.f : xslope;
                        };
                        float4 dz;
                        dz.x = gq.x * pact(xc.x, xsc.lo.x, xsh.lo.x); dz.y = gq.y * pact(xc.y, xsc.lo.y, xsh.lo.y);
                        dz.z = gq.z * pact(xc.z, xsc.hi.x, xsh.hi.x); dz.w = gq.w * pact(xc.w, xsc.hi.y, xsh.hi.y);
                        if (gm.dz)                                  // the unit in front takes its gradient pre-multiplied: ca = gamma * invstd = in_scale
                            st4_stream(dx + o, make_float4(dz.x * xsc.lo.x, dz.y * xsc.lo.y, dz.z * xsc.hi.x, dz.w * xsc.hi.y));
                        float4 a1 = xcs[256 + tid], a2 = xcs[512 + tid];
                        add4(a1, dz);
                        fma4(a2, dz, make_float4((xc.x - mu.x) * is.x, (xc.y - mu.y) * is.y, (xc.z - mu.z) * is.z, (xc.w - mu.w) * is.w));
                        xcs[256 + tid] = a1; xcs[512 + tid] = a2;
                    }
                }
                if (RED) xcs[tid] = xc4;                            // raw centre input of row r, for the store of the next iteration
                P0 = P1; P1 = P2;
                // weight gradient: input row i = r-1 (own column) against dY rows i+1 = r (kr = 0), i (kr = 1), i-1 (kr = 2), columns w+1-kq
#pragma unroll
                for (int kq = 0; kq < 3; ++kq) {
                    pfma(wp[0 + kq], aprev, dyr[2 - kq]);
                    pfma(wp[3 + kq], aprev, dyp1[2 - kq]);
                    pfma(wp[6 + kq], aprev, dyp2[2 - kq]);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) { dyp2[q] = dyp1[q]; dyp1[q] = dyr[q]; }
                aprev = ar;
            }
            // (input row h1-1 was processed in the last iteration, r = h1: nothing is left over)
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) wacc[t] = f4u(wp[t]);
#undef WG
    }

    // deterministic block reduction of the 9 tap accumulators over the `ppb` pixel slots
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        red[tid] = wacc[t];
        __syncthreads();
        if (pix == 0 && cvalid) {
            float4 a = f4zero();
            for (int p = 0; p < ppb; ++p) add4(a, red[p * gm.cgb + cgl]);
            float* dst = parts + (int64_t)blockIdx.x * gm.C * 9;
            dst[(c + 0) * 9 + t] = a.x; dst[(c + 1) * 9 + t] = a.y;
            dst[(c + 2) * 9 + t] = a.z; dst[(c + 3) * 9 + t] = a.w;
        }
    }
    if (RED) {                                           // per-block partial row of the producer unit's BN-backward sums, fixed order
        __syncthreads();
        if (pix == 0 && cvalid) {
            float4 a = f4zero(), b = f4zero();
            for (int p = 0; p < ppb; ++p) { add4(a, xcs[256 + p * gm.cgb + cgl]); add4(b, xcs[512 + p * gm.cgb + cgl]); }
            float* dst = in_red + (int64_t)blockIdx.x * 2 * gm.C;
            st4(dst + c, a);
            st4(dst + gm.C + c, b);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward of a depthwise 3x3 STRIDE-2 conv+BN+activation unit (the four down-sampling units of MobileNetV2).
// Unfused: bn_bwd_apply (read G, Y; write dY) + dw_bwd_weight (read X at 4x the size, dY) + dw_bwd_data (read dY, write dX at
// 4x the size) = 13 output-sized passes; here G, Y and X are read once and dX written once (10), in one launch instead of three.
// Thread = 4 channels x one input-quad column j (input columns 2j, 2j+1 = output column j), walking down quad rows i:
//   dY[i+1][j], dY[i+1][j+1] are rebuilt from (G, Y) each step, row i is carried over from the previous one;
//   dX of the 2x2 input quad from dY[i..i+1][j..j+1] with statically known taps (see dw_bwd_data_s2k3_kernel);
//   dW += dY[i][j] * a[2i-1..2i+1][2j-1..2j+1]: input rows 2i, 2i+1 are loaded (three columns), row 2i-1 is carried over.
// Same LDS-resident per-channel constants, masks instead of selects and XCD-contiguous strips as the stride-1 kernel.
// RDZ (round 6, fp32 storage, ReLU6 input view): the input is the raw output of a wide expand unit P consumed only here and P runs the low-rank BN
// backward (csrc/lrbwd.hip): the kernel stores  in_scale o dX o act_P'(z)  instead of dX and leaves P's BN-backward sums (sum dz, sum dz * xhat) as
// partial rows in_red[gridDim.x][2][C] — mny_dw_bnbwd_red_dz for the stride-2 unit.
template <typename T, int AM, int XF, bool RDZ = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void dw_bnbwd_s2k3_kernel(
    const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    const float* __restrict__ coef, const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int in_act, const float* __restrict__ w, const T* __restrict__ addend, T* __restrict__ dx, float* __restrict__ parts, DwbGeom gm,
    const float* __restrict__ in_mean = nullptr, const float* __restrict__ in_invstd = nullptr, float* __restrict__ in_red = nullptr) {
    __shared__ float4 red[256];
    extern __shared__ __attribute__((aligned(16))) float4 cst[];
    const int tid = threadIdx.x;
    const int cgl = tid % gm.cgb, pix = tid / gm.cgb, ppb = blockDim.x / gm.cgb;
    const int cg = blockIdx.y * gm.cgb + cgl;
    const bool cvalid = cg < gm.cg_total;
    const int c = cg * 4;
    const int Ho = (gm.H - 1) / 2 + 1, Wo = (gm.W - 1) / 2 + 1;        // = quad rows / columns

    F4P wp[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wp[t] = f4p0();
    if (pix == 0 && cvalid) {
#pragma unroll
        for (int t = 0; t < 9; ++t) cst[t * gm.cgb + cgl] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
        cst[9 * gm.cgb + cgl] = ld4(scale + c);
        cst[10 * gm.cgb + cgl] = ld4(shift + c);
        cst[11 * gm.cgb + cgl] = ld4(coef + c);
        cst[12 * gm.cgb + cgl] = ld4(coef + gm.C + c);
        cst[13 * gm.cgb + cgl] = ld4(coef + 2 * gm.C + c);
        cst[14 * gm.cgb + cgl] = (XF != 0 && in_scale) ? ld4(in_scale + c) : f4one();
        cst[15 * gm.cgb + cgl] = (XF != 0 && in_scale) ? ld4(in_shift + c) : f4zero();
        if (RDZ) { cst[16 * gm.cgb + cgl] = ld4(in_mean + c); cst[17 * gm.cgb + cgl] = ld4(in_invstd + c); }
    }
    F4P rs1 = f4p0(), rs2 = f4p0();                      // RDZ: the producer's BN-backward sums of this thread's channels
    __syncthreads();
    if (cvalid) {
#define WG(t) f4p(my[(t) * gm.cgb])
        const float slope = act_slope(act), hi = act_hi(act);
        const float xslope = act_slope(in_act);
        auto dact = [&](float z) {
            if (AM == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
            return (z > 0.f ? 1.f : slope) * (z < hi ? 1.f : 0.f);
        };
        auto dy2 = [&](v2f gv, v2f yv, v2f s, v2f h, v2f a, v2f b, v2f cterm) {
            v2f d = gv;
            if (AM != 0) { const v2f z = __builtin_elementwise_fma(yv, s, h); d = gv * v2f{dact(z.x), dact(z.y)}; }
            return __builtin_elementwise_fma(a, d, __builtin_elementwise_fma(b, yv, cterm));
        };
        auto xf2 = [&](v2f v, v2f s, v2f h) {
            if (XF == 0) return v;
            const v2f z = __builtin_elementwise_fma(v, s, h);
            if (XF == 1) return v2f{__builtin_amdgcn_fmed3f(z.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.y, 0.f, 6.f)};
            if (XF == 3) { const v2f t = z * v2f{xslope, xslope}; return v2f{fmaxf(z.x, t.x), fmaxf(z.y, t.y)}; }
            const v2f t = z + v2f{3.f, 3.f};
            return z * v2f{__builtin_amdgcn_fmed3f(t.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(t.y, 0.f, 6.f)} * v2f{1.f / 6.f, 1.f / 6.f};
        };
        const int gxd = gridDim.x;
        const int lb = (gm.xcd && (gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        const int64_t pitch = (int64_t)gm.W * gm.C, opitch = (int64_t)Wo * gm.C;
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < gm.nstrips; strip += (int64_t)gxd * ppb) {
            const int j = (int)(strip % Wo);
            const int hs = (int)((strip / Wo) % gm.nHS);
            const int n = (int)(strip / ((int64_t)Wo * gm.nHS));
            const int i0 = hs * gm.TH, i1 = min(i0 + gm.TH, Ho);
            const T* xn = x + (int64_t)n * gm.H * pitch + c;
            const int64_t on = (int64_t)n * Ho * opitch + c;
            int xoff[3];
            float xm[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int wi = 2 * j - 1 + q; xm[q] = (wi >= 0 && wi < gm.W) ? 1.f : 0.f; xoff[q] = min(max(wi, 0), gm.W - 1) * gm.C; }
            const float jm1 = (j + 1 < Wo) ? 1.f : 0.f;
            const int joff0 = j * gm.C, joff1 = min(j + 1, Wo - 1) * gm.C;

            // dY at output row `ho`, columns j and j+1 (zero outside the output)
            auto dy_row = [&](int ho, const float4* my, F4P& d0, F4P& d1) {
                const float rm = (ho < Ho) ? 1.f : 0.f;
                const int64_t ro = on + (int64_t)min(ho, Ho - 1) * opitch;
                const float4 g0 = ld4(g + ro + joff0), y0 = ld4(y + ro + joff0), g1 = ld4(g + ro + joff1), y1 = ld4(y + ro + joff1);
                const F4P sc = f4p(my[9 * gm.cgb]), sh = f4p(my[10 * gm.cgb]), ca = f4p(my[11 * gm.cgb]), cb = f4p(my[12 * gm.cgb]), cc = f4p(my[13 * gm.cgb]);
                const F4P G0 = f4p(g0), Y0 = f4p(y0), G1 = f4p(g1), Y1 = f4p(y1);
                const v2f m0 = v2f{rm, rm}, m1 = v2f{rm * jm1, rm * jm1};
                d0.lo = dy2(G0.lo, Y0.lo, sc.lo, sh.lo, ca.lo, cb.lo, cc.lo) * m0; d0.hi = dy2(G0.hi, Y0.hi, sc.hi, sh.hi, ca.hi, cb.hi, cc.hi) * m0;
                d1.lo = dy2(G1.lo, Y1.lo, sc.lo, sh.lo, ca.lo, cb.lo, cc.lo) * m1; d1.hi = dy2(G1.hi, Y1.hi, sc.hi, sh.hi, ca.hi, cb.hi, cc.hi) * m1;
            };
            // activated input row hi, columns 2j, 2j+1 — the thread's OWN input quad (zero outside the image).  Round 3: the weight gradient is
            // summed over the input pixels a thread owns (each meets the one to four dY values whose window covers it, with the taps the
            // data gradient uses), so no halo column 2j-1 and no halo row 2i-1 of the 4x-sized input tensor is read any more: 4 input
            // loads per quad row instead of 6 (+ a carried row).
            auto x_row = [&](int hi, const float4* my, F4P (&a)[2], float4 (&raw)[2]) {
                const float rm = (hi >= 0 && hi < gm.H) ? 1.f : 0.f;
                const T* p = xn + (int64_t)min(max(hi, 0), gm.H - 1) * pitch;
#pragma unroll
                for (int q = 0; q < 2; ++q) raw[q] = ld4(p + xoff[q + 1]);
                const F4P xsc = f4p(my[14 * gm.cgb]), xsh = f4p(my[15 * gm.cgb]);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float m = rm * xm[q + 1];
                    const F4P X4 = f4p(raw[q]);
                    a[q].lo = xf2(X4.lo, xsc.lo, xsh.lo) * v2f{m, m};
                    a[q].hi = xf2(X4.hi, xsc.hi, xsh.hi) * v2f{m, m};
                }
            };
            F4P d00, d01;
            {
                int lo = cgl;
                asm volatile("" : "+v"(lo));
                const float4* my = cst + lo;
                dy_row(i0, my, d00, d01);
            }
            for (int i = i0; i < i1; ++i) {
                int lo = cgl;
                asm volatile("" : "+v"(lo));                       // keeps the LDS constant reads inside the loop (see the stride-1 kernel)
                const float4* my = cst + lo;
                F4P d10, d11, a0[2], a1[2];
                float4 raw0[2], raw1[2];
                dy_row(i + 1, my, d10, d11);
                x_row(2 * i, my, a0, raw0);
                x_row(2 * i + 1, my, a1, raw1);
                // weight gradient: input (2i, 2j) <-> tap 4 of dY[i][j]; (2i, 2j+1) <-> taps 5 / 3 of dY[i][j] / dY[i][j+1]; (2i+1, 2j) <-> 7 / 1 of
                // dY[i][j] / dY[i+1][j]; (2i+1, 2j+1) <-> 8 / 6 / 2 / 0 of dY[i][j] / dY[i][j+1] / dY[i+1][j] / dY[i+1][j+1]
                pfma(wp[4], a0[0], d00);
                pfma(wp[5], a0[1], d00); pfma(wp[3], a0[1], d01);
                pfma(wp[7], a1[0], d00); pfma(wp[1], a1[0], d10);
                pfma(wp[8], a1[1], d00); pfma(wp[6], a1[1], d01); pfma(wp[2], a1[1], d10); pfma(wp[0], a1[1], d11);
                // data gradient of the quad (2i..2i+1, 2j..2j+1): taps as in dw_bwd_data_s2k3_kernel
                F4P o00 = f4p0(), o01 = f4p0(), o10 = f4p0(), o11 = f4p0();
                pfma(o00, d00, WG(4));
                pfma(o01, d00, WG(5)); pfma(o01, d01, WG(3));
                pfma(o10, d00, WG(7)); pfma(o10, d10, WG(1));
                pfma(o11, d00, WG(8)); pfma(o11, d01, WG(6)); pfma(o11, d10, WG(2)); pfma(o11, d11, WG(0));
                const int h0 = 2 * i, w0 = 2 * j;
                const int64_t base = (int64_t)n * gm.H * pitch + (int64_t)h0 * pitch + (int64_t)w0 * gm.C + c;
                const bool hv = h0 + 1 < gm.H, wv2 = w0 + 1 < gm.W;
                float4 f00 = f4u(o00), f01 = f4u(o01), f10 = f4u(o10), f11 = f4u(o11);
                if (addend) {
                    add4(f00, ld4(addend + base));
                    if (wv2) add4(f01, ld4(addend + base + gm.C));
                    if (hv) add4(f10, ld4(addend + base + pitch));
                    if (hv && wv2) add4(f11, ld4(addend + base + pitch + gm.C));
                }
                if constexpr (RDZ) {
                    // dz = dX * relu6'(z) of the producer, its sums, and the stored value in_scale o dz (pixels outside the image: dz = 0 by the masks below)
                    const F4P xsc = f4p(my[14 * gm.cgb]), xsh = f4p(my[15 * gm.cgb]), mu = f4p(my[16 * gm.cgb]), is = f4p(my[17 * gm.cgb]);
                    auto one = [&](float4& f, const float4 rawv, float ok) {
                        const F4P R = f4p(rawv), F = f4p(f);
                        const v2f z0 = __builtin_elementwise_fma(R.lo, xsc.lo, xsh.lo), z1 = __builtin_elementwise_fma(R.hi, xsc.hi, xsh.hi);
                        const v2f m0 = v2f{(z0.x > 0.f && z0.x < 6.f) ? ok : 0.f, (z0.y > 0.f && z0.y < 6.f) ? ok : 0.f};
                        const v2f m1 = v2f{(z1.x > 0.f && z1.x < 6.f) ? ok : 0.f, (z1.y > 0.f && z1.y < 6.f) ? ok : 0.f};
                        const v2f d0 = F.lo * m0, d1 = F.hi * m1;
                        rs1.lo += d0; rs1.hi += d1;
                        rs2.lo = __builtin_elementwise_fma(d0, (R.lo - mu.lo) * is.lo, rs2.lo);
                        rs2.hi = __builtin_elementwise_fma(d1, (R.hi - mu.hi) * is.hi, rs2.hi);
                        const v2f o0 = d0 * xsc.lo, o1 = d1 * xsc.hi;
                        f = make_float4(o0.x, o0.y, o1.x, o1.y);
                    };
                    one(f00, raw0[0], 1.f);
                    one(f01, raw0[1], wv2 ? 1.f : 0.f);
                    one(f10, raw1[0], hv ? 1.f : 0.f);
                    one(f11, raw1[1], (hv && wv2) ? 1.f : 0.f);
                }
                st4_stream(dx + base, f00);
                if (wv2) st4_stream(dx + base + gm.C, f01);
                if (hv) st4_stream(dx + base + pitch, f10);
                if (hv && wv2) st4_stream(dx + base + pitch + gm.C, f11);
                d00 = d10; d01 = d11;
            }
        }
#undef WG
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        red[tid] = f4u(wp[t]);
        __syncthreads();
        if (pix == 0 && cvalid) {
            float4 a = f4zero();
            for (int p = 0; p < ppb; ++p) add4(a, red[p * gm.cgb + cgl]);
            float* dst = parts + (int64_t)blockIdx.x * gm.C * 9;
            dst[(c + 0) * 9 + t] = a.x; dst[(c + 1) * 9 + t] = a.y;
            dst[(c + 2) * 9 + t] = a.z; dst[(c + 3) * 9 + t] = a.w;
        }
    }
    if constexpr (RDZ) {                                 // per-block partial row of the producer's sums, fixed order over the pixel slots
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __syncthreads();
            red[tid] = f4u(k == 0 ? rs1 : rs2);
            __syncthreads();
            if (pix == 0 && cvalid) {
                float4 a = f4zero();
                for (int p = 0; p < ppb; ++p) add4(a, red[p * gm.cgb + cgl]);
                st4(in_red + (int64_t)blockIdx.x * 2 * gm.C + k * gm.C + c, a);
            }
        }
    }
}

// geometry of the stride-2 kernel: strips of TH quad rows over (N, quad columns)
static int dwb2_geom(DwbGeom& g, CgLayout& L, int& gx, int N, int H, int W, int C) {
    MNY_REQUIRE(C % 4 == 0 && C > 0, "dw_bnbwd_s2: C=%d must be a positive multiple of 4", C);
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "dw_bnbwd_s2: empty tensor");
    g.N = N; g.H = H; g.W = W; g.C = C; g.dz = 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    static const int th2 = getenv("MNY_DWB2_TH") ? atoi(getenv("MNY_DWB2_TH")) : 16;     // quad rows per strip (the 4 launches of a step: 8: 2.33, 16: 2.30, 32: 2.42, 64: 2.50 ms)
    const int ns = (int)cdiv(Ho, th2);
    g.TH = (int)cdiv(Ho, ns);
    g.nHS = (int)cdiv(Ho, g.TH);
    g.nstrips = (int64_t)N * Wo * g.nHS;
    L = make_stencil_layout(C);
    g.cg_total = L.cg_total; g.cgb = L.cgb;
    int64_t want = cdiv(g.nstrips, L.ppb);
    static const int res2 = getenv("MNY_DWB_RES") ? atoi(getenv("MNY_DWB_RES")) : 768;       // resident workgroups (3 per CU at <= 168 VGPRs)
    int cap = res2 / L.chunks > 0 ? res2 / L.chunks : 1;
    g.xcd = 1;
    if (cap > 8) cap &= ~7;
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < cap ? want : cap);
    return MNY_OK;
}

template <typename T>
static int dw_bnbwd_s2_impl(const T* g, const T* y, const float* scale, const float* shift, int act, const float* coef,
                            const T* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                            const T* addend, T* dx, float* dw, float* ws, int N, int H, int W, int C, void* stream) {
    MNY_REQUIRE(g && y && scale && shift && coef && x && w && dx && ws, "dw_bnbwd_s2: null pointer");
    MNY_REQUIRE(act != MNY_ACT_HSIGMOID && in_act != MNY_ACT_HSIGMOID, "dw_bnbwd_s2: h-sigmoid views are not supported");
    DwbGeom gm; CgLayout L; int gx;
    int rc = dwb2_geom(gm, L, gx, N, H, W, C);
    if (rc) return rc;
    dim3 grid(gx, L.chunks), block(L.threads);
    hipStream_t st = (hipStream_t)stream;
    const int am = act == MNY_ACT_NONE ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
    const int xf = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : (in_act == MNY_ACT_RELU6 ? 1 : 3));
    const size_t lds = (size_t)16 * L.cgb * sizeof(float4);
#define MNY_L2(A_, X_) hipLaunchKernelGGL((dw_bnbwd_s2k3_kernel<T, A_, X_>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, \
                                          in_act, w, addend, dx, ws, gm)
    switch (am * 4 + xf) {
        case 0: MNY_L2(0, 0); break; case 1: MNY_L2(0, 1); break; case 2: MNY_L2(0, 2); break; case 3: MNY_L2(0, 3); break;
        case 4: MNY_L2(1, 0); break; case 5: MNY_L2(1, 1); break; case 6: MNY_L2(1, 2); break; case 7: MNY_L2(1, 3); break;
        case 8: MNY_L2(2, 0); break; case 9: MNY_L2(2, 1); break; case 10: MNY_L2(2, 2); break; default: MNY_L2(2, 3); break;
    }
#undef MNY_L2
    rc = check_launch("dw_bnbwd_s2k3_kernel");
    if (rc || !dw) return rc;
    return launch_reduce_parts(ws, gx, C * 9, dw, st);
}

// stride-2 unit behind a wide expand unit on the low-rank BN backward: dx = in_scale o dX o relu6'(z), + the producer's sums (fp32 storage, ReLU6 view)
static int dw_bnbwd_s2_red_dz_impl(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                   const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                   const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C, void* stream) {
    MNY_REQUIRE(g && y && scale && shift && coef && x && w && dx && ws && in_scale && in_shift && in_mean && in_invstd && in_red, "dw_bnbwd_s2_red_dz: null pointer");
    MNY_REQUIRE(in_act == MNY_ACT_RELU6 && act != MNY_ACT_HSIGMOID, "dw_bnbwd_s2_red_dz: the producer's activation must be ReLU6 (got %d)", in_act);
    DwbGeom gm; CgLayout L; int gx;
    int rc = dwb2_geom(gm, L, gx, N, H, W, C);
    if (rc) return rc;
    dim3 grid(gx, L.chunks), block(L.threads);
    hipStream_t st = (hipStream_t)stream;
    const int am = act == MNY_ACT_NONE ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
    const size_t lds = (size_t)18 * L.cgb * sizeof(float4);
#define MNY_L2R(A_) hipLaunchKernelGGL((dw_bnbwd_s2k3_kernel<float, A_, 1, true>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, \
                                       in_act, w, addend, dx, ws, gm, in_mean, in_invstd, in_red)
    switch (am) { case 0: MNY_L2R(0); break; case 1: MNY_L2R(1); break; default: MNY_L2R(2); break; }
#undef MNY_L2R
    rc = check_launch("dw_bnbwd_s2k3_kernel<red_dz>");
    if (rc || !dw) return rc;
    return launch_reduce_parts(ws, gx, C * 9, dw, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward of a depthwise 5x5 STRIDE-2 conv+BN+activation unit (MobileNetV3: Block(5, 24 -> 72 -> 40, s2) and
// Block(5, 160 -> 672 -> 160, s2), models/mobilenetv3.py:88,100).  Unfused these ran bn_bwd_apply + dw5_wgrad + dw_bwd_data_s2k5
// (+ the producer's bn_bwd_reduce); here (G, Y, X) are read once and dX written once, in one launch.
// The register form of the 3x3 stride-2 kernel above carries over because a stride-2 unit's input quad (rows 2i, 2i+1 x columns 2j, 2j+1)
// meets only a 3 x 3 window of dY — rows i-1..i+1, columns j-1..j+1 — whatever the filter size; what grows is the tap count: the even
// input row 2i meets dY rows i-1 / i / i+1 through filter rows 4 / 2 / 0, the odd row 2i+1 meets rows i / i+1 through 3 / 1 (the same
// for columns), so the quad uses each of the 25 taps exactly once: 25 packed FMAs for dX and 25 for dW per step.  Thread = TWO channels
// x one quad column (25 accumulator pairs = 50 VGPRs; four channels would be 100 + a 36-register window: two waves per SIMD), walking
// down quad rows: dY row i+1 is rebuilt from (G, Y) at three columns per step, rows i-1 and i are carried.  Taps and per-channel
// constants in LDS as pairs.  RED: the input is the raw output of a conv+BN+act unit consumed only here — its BN-backward sums
// (sum dz, sum dz * xhat) leave with dX as partial rows in_red[gridDim.x][2][C].
template <typename T> __device__ __forceinline__ v2f ld2(const T* p);
template <> __device__ __forceinline__ v2f ld2<float>(const float* p) { return *reinterpret_cast<const v2f*>(p); }
template <> __device__ __forceinline__ v2f ld2<bf16_t>(const bf16_t* p) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return v2f{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ void st2_stream(float* p, v2f v) { __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(p)); }
__device__ __forceinline__ void st2_stream(bf16_t* p, v2f v) { __builtin_nontemporal_store(pack_bf16x2(v.x, v.y), reinterpret_cast<uint32_t*>(p)); }

constexpr int kDw5Consts = 25 + 9;      // 25 taps, scale, shift, ca, cb, cc, input scale / shift, producer mean / invstd

template <typename T, int AM, int XF, bool RED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void dw_bnbwd_s2k5_kernel(
    const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    const float* __restrict__ coef, const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int in_act, const float* __restrict__ w, const T* __restrict__ addend, T* __restrict__ dx, float* __restrict__ parts, DwbGeom gm,
    const float* __restrict__ in_mean, const float* __restrict__ in_invstd, float* __restrict__ in_red) {
    __shared__ v2f red[256];
    extern __shared__ __attribute__((aligned(16))) v2f cst2[];          // [kDw5Consts][cgb]
    const int tid = threadIdx.x;
    const int cgl = tid % gm.cgb, pix = tid / gm.cgb, ppb = blockDim.x / gm.cgb;
    const int cg = blockIdx.y * gm.cgb + cgl;                            // channel PAIR
    const bool cvalid = cg < gm.cg_total;
    const int c = cg * 2;
    const int Ho = (gm.H - 1) / 2 + 1, Wo = (gm.W - 1) / 2 + 1;          // = quad rows / columns (pad 2: (H + 4 - 5) / 2 + 1)

    v2f wacc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) wacc[t] = v2f{0.f, 0.f};
    if (pix == 0 && cvalid) {
        auto put = [&](int k, const float* src, float fill) { cst2[k * gm.cgb + cgl] = src ? v2f{src[c], src[c + 1]} : v2f{fill, fill}; };
#pragma unroll
        for (int t = 0; t < 25; ++t) cst2[t * gm.cgb + cgl] = v2f{w[c * 25 + t], w[(c + 1) * 25 + t]};
        put(25, scale, 1.f); put(26, shift, 0.f);
        put(27, coef, 0.f); put(28, coef + gm.C, 0.f); put(29, coef + 2 * gm.C, 0.f);
        put(30, XF != 0 ? in_scale : nullptr, 1.f); put(31, XF != 0 ? in_shift : nullptr, 0.f);
        put(32, RED ? in_mean : nullptr, 0.f); put(33, RED ? in_invstd : nullptr, 1.f);
    }
    v2f rs1 = v2f{0.f, 0.f}, rs2 = v2f{0.f, 0.f};
    __syncthreads();
    if (cvalid) {
        const float slope = act_slope(act), hi = act_hi(act);
        const float xslope = act_slope(in_act);
        auto dact = [&](float z) {
            if (AM == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
            return (z > 0.f ? 1.f : slope) * (z < hi ? 1.f : 0.f);
        };
        auto xf2 = [&](v2f v, v2f s, v2f h) {
            if (XF == 0) return v;
            const v2f z = __builtin_elementwise_fma(v, s, h);
            if (XF == 1) return v2f{__builtin_amdgcn_fmed3f(z.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.y, 0.f, 6.f)};
            if (XF == 3) { const v2f t = z * v2f{xslope, xslope}; return v2f{fmaxf(z.x, t.x), fmaxf(z.y, t.y)}; }
            const v2f t = z + v2f{3.f, 3.f};
            return z * v2f{__builtin_amdgcn_fmed3f(t.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(t.y, 0.f, 6.f)} * v2f{1.f / 6.f, 1.f / 6.f};
        };
        auto pact = [&](float z) {                             // derivative of the producer's activation (RED)
            if (XF == 1) return (z > 0.f ? 1.f : 0.f) * (z < 6.f ? 1.f : 0.f);
            if (XF == 2) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
            return z > 0.f ? 1.f : xslope;
        };
        const int gxd = gridDim.x;
        const int lb = (gm.xcd && (gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        const int64_t pitch = (int64_t)gm.W * gm.C, opitch = (int64_t)Wo * gm.C;
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < gm.nstrips; strip += (int64_t)gxd * ppb) {
            const int j = (int)(strip % Wo);
            const int hs = (int)((strip / Wo) % gm.nHS);
            const int n = (int)(strip / ((int64_t)Wo * gm.nHS));
            const int i0 = hs * gm.TH, i1 = min(i0 + gm.TH, Ho);
            // 64-bit image bases once per strip; inside an image 32-bit element offsets, advanced by addition (an image plane stays below 2^31 elements:
            // checked by the launcher) — the row loop is bound by instruction issue, 64-bit multiplies per row were 24 of its 226 vector instructions
            const int64_t ximg = (int64_t)n * gm.H * pitch + c;
            const T* xn = x + ximg;
            const T* an = addend ? addend + ximg : nullptr;
            T* dn = dx + ximg;
            const T* gn = g + (int64_t)n * Ho * opitch + c;
            const T* yn = y + (int64_t)n * Ho * opitch + c;
            const int pitch32 = (int)pitch, opitch32 = (int)opitch;
            int goff[3];
            float cm[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int wo = j - 1 + q; cm[q] = (wo >= 0 && wo < Wo) ? 1.f : 0.f; goff[q] = min(max(wo, 0), Wo - 1) * gm.C; }
            const bool wv2 = 2 * j + 1 < gm.W;
            const int xoff0 = 2 * j * gm.C, xoff1 = min(2 * j + 1, gm.W - 1) * gm.C;
            const int olast = (Ho - 1) * opitch32, xlast = (gm.H - 1) * pitch32;

            // dY of output row `ho` (element offset `ro` of its clamped row) at columns j-1, j, j+1 (zero outside the output)
            auto dy_row = [&](int ho, int ro, const v2f* my, v2f (&d)[3]) {
                const float rm = (ho >= 0 && ho < Ho) ? 1.f : 0.f;
                v2f gv[3], yv[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) { gv[q] = ld2<T>(gn + ro + goff[q]); yv[q] = ld2<T>(yn + ro + goff[q]); }
                const v2f sc = my[25 * gm.cgb], sh = my[26 * gm.cgb], ca = my[27 * gm.cgb], cb = my[28 * gm.cgb], cc = my[29 * gm.cgb];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    v2f dd = gv[q];
                    if (AM != 0) { const v2f z = __builtin_elementwise_fma(yv[q], sc, sh); dd = gv[q] * v2f{dact(z.x), dact(z.y)}; }
                    const float m = rm * cm[q];
                    d[q] = __builtin_elementwise_fma(ca, dd, __builtin_elementwise_fma(cb, yv[q], cc)) * v2f{m, m};
                }
            };
            v2f D[3][3];                                         // D[dr][dq] = dY[i - 1 + dr][j - 1 + dq]
            {
                int lo = cgl;
                asm volatile("" : "+v"(lo));
                const v2f* my = cst2 + lo;
                dy_row(i0 - 1, max(i0 - 1, 0) * opitch32, my, D[0]);
                dy_row(i0, i0 * opitch32, my, D[1]);
            }
            int ro_next = min((i0 + 1) * opitch32, olast);       // row i + 1 of the step
            int xo0 = 2 * i0 * pitch32;                          // input row 2 i
            for (int i = i0; i < i1; ++i) {
                int lo = cgl;
                asm volatile("" : "+v"(lo));                       // keeps the LDS constant reads inside the loop
                const v2f* my = cst2 + lo;
                dy_row(i + 1, ro_next, my, D[2]);
                ro_next = min(ro_next + opitch32, olast);
                // the thread's own input quad, raw and activated (zero outside the image)
                v2f raw[2][2], A[2][2];
                const bool hv = 2 * i + 1 < gm.H;
                const int xo1 = min(xo0 + pitch32, xlast);
                {
                    raw[0][0] = ld2<T>(xn + xo0 + xoff0); raw[0][1] = ld2<T>(xn + xo0 + xoff1);
                    raw[1][0] = ld2<T>(xn + xo1 + xoff0); raw[1][1] = ld2<T>(xn + xo1 + xoff1);
                    const v2f xsc = my[30 * gm.cgb], xsh = my[31 * gm.cgb];
                    const float m01 = wv2 ? 1.f : 0.f, m10 = hv ? 1.f : 0.f, m11 = (hv && wv2) ? 1.f : 0.f;
                    A[0][0] = xf2(raw[0][0], xsc, xsh);
                    A[0][1] = xf2(raw[0][1], xsc, xsh) * v2f{m01, m01};
                    A[1][0] = xf2(raw[1][0], xsc, xsh) * v2f{m10, m10};
                    A[1][1] = xf2(raw[1][1], xsc, xsh) * v2f{m11, m11};
                }
                v2f o[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) o[a][b] = v2f{0.f, 0.f};
                // input (2i + a, 2j + b) meets dY[i - 1 + dr][j - 1 + dq] through tap (kh, kw) = (4 + a - 2 dr, 4 + b - 2 dq): even rows / columns
                // dr, dq = 0..2, odd ones 1..2
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int dr = a; dr < 3; ++dr)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int dq = b; dq < 3; ++dq) {
                                const int u = (4 + a - 2 * dr) * 5 + (4 + b - 2 * dq);
                                const v2f d = D[dr][dq];
                                o[a][b] = __builtin_elementwise_fma(d, my[u * gm.cgb], o[a][b]);
                                wacc[u] = __builtin_elementwise_fma(A[a][b], d, wacc[u]);
                            }
                const int base = xo0 + xoff0;                     // element offset of the quad's first pixel inside the image
                if (an) {
                    o[0][0] += ld2<T>(an + base);
                    if (wv2) o[0][1] += ld2<T>(an + base + gm.C);
                    if (hv) o[1][0] += ld2<T>(an + base + pitch32);
                    if (hv && wv2) o[1][1] += ld2<T>(an + base + pitch32 + gm.C);
                }
                st2_stream(dn + base, o[0][0]);
                if (wv2) st2_stream(dn + base + gm.C, o[0][1]);
                if (hv) st2_stream(dn + base + pitch32, o[1][0]);
                if (hv && wv2) st2_stream(dn + base + pitch32 + gm.C, o[1][1]);
                xo0 += 2 * pitch32;
                if constexpr (RED) {
                    const v2f xsc = my[30 * gm.cgb], xsh = my[31 * gm.cgb], mu = my[32 * gm.cgb], is = my[33 * gm.cgb];
                    auto one = [&](v2f ov, v2f rv, float ok) {
                        const v2f gq = v2f{stored<T>(ov.x), stored<T>(ov.y)};            // the gradient as the consumer of dX reads it back
                        const v2f z = __builtin_elementwise_fma(rv, xsc, xsh);
                        const v2f dz = gq * v2f{pact(z.x) * ok, pact(z.y) * ok};
                        rs1 += dz;
                        rs2 = __builtin_elementwise_fma(dz, (rv - mu) * is, rs2);
                    };
                    one(o[0][0], raw[0][0], 1.f);
                    one(o[0][1], raw[0][1], wv2 ? 1.f : 0.f);
                    one(o[1][0], raw[1][0], hv ? 1.f : 0.f);
                    one(o[1][1], raw[1][1], (hv && wv2) ? 1.f : 0.f);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) { D[0][q] = D[1][q]; D[1][q] = D[2][q]; }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        __syncthreads();
        red[tid] = wacc[t];
        __syncthreads();
        if (pix == 0 && cvalid) {
            v2f a = v2f{0.f, 0.f};
            for (int p = 0; p < ppb; ++p) a += red[p * gm.cgb + cgl];
            float* dst = parts + (int64_t)blockIdx.x * gm.C * 25;
            dst[c * 25 + t] = a.x; dst[(c + 1) * 25 + t] = a.y;
        }
    }
    if constexpr (RED) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            __syncthreads();
            red[tid] = k == 0 ? rs1 : rs2;
            __syncthreads();
            if (pix == 0 && cvalid) {
                v2f a = v2f{0.f, 0.f};
                for (int p = 0; p < ppb; ++p) a += red[p * gm.cgb + cgl];
                *reinterpret_cast<v2f*>(in_red + (int64_t)blockIdx.x * 2 * gm.C + k * gm.C + c) = a;
            }
        }
    }
}

// geometry of the 5x5 stride-2 kernel: channel PAIRS, <= 64 pairs per workgroup (grid.y chunks), strips of TH quad rows
static int dwb5_geom(DwbGeom& g, int& chunks, int& threads, int& gx, int N, int H, int W, int C) {
    MNY_REQUIRE(C % 2 == 0 && C > 0, "dw_bnbwd_s2k5: C=%d must be a positive even number", C);
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "dw_bnbwd_s2k5: empty tensor");
    g.N = N; g.H = H; g.W = W; g.C = C; g.dz = 0;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    static const int th5 = getenv("MNY_DWB5_TH") ? atoi(getenv("MNY_DWB5_TH")) : 16;
    const int ns = (int)cdiv(Ho, th5);
    g.TH = (int)cdiv(Ho, ns);
    g.nHS = (int)cdiv(Ho, g.TH);
    g.nstrips = (int64_t)N * Wo * g.nHS;
    g.cg_total = C / 2;
    static const int cgb5 = getenv("MNY_DWB5_CGB") ? atoi(getenv("MNY_DWB5_CGB")) : 64;
    chunks = (int)cdiv(g.cg_total, cgb5);
    g.cgb = (int)cdiv(g.cg_total, chunks);
    const int ppb = 256 / g.cgb > 0 ? 256 / g.cgb : 1;
    threads = g.cgb * ppb;
    int64_t want = cdiv(g.nstrips, ppb);
    static const int res5 = getenv("MNY_DWB5_RES") ? atoi(getenv("MNY_DWB5_RES")) : 768;
    int cap = res5 / chunks > 0 ? res5 / chunks : 1;
    g.xcd = 1;
    if (cap > 8) cap &= ~7;
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < cap ? want : cap);
    return MNY_OK;
}

template <typename T>
static int dw_bnbwd_s2k5_impl(const T* g, const T* y, const float* scale, const float* shift, int act, const float* coef,
                              const T* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                              const float* w, const T* addend, T* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C, void* stream) {
    MNY_REQUIRE(g && y && scale && shift && coef && x && w && dx && ws, "dw_bnbwd_s2k5: null pointer");
    MNY_REQUIRE(act != MNY_ACT_HSIGMOID && in_act != MNY_ACT_HSIGMOID, "dw_bnbwd_s2k5: h-sigmoid views are not supported");
    MNY_REQUIRE(!in_red || (in_mean && in_invstd && in_scale && in_shift), "dw_bnbwd_s2k5: the producer's sums need its mean / invstd and view");
    MNY_REQUIRE((int64_t)H * W * C < ((int64_t)1 << 31), "dw_bnbwd_s2k5: an image plane of %d x %d x %d elements exceeds 32-bit offsets", H, W, C);
    DwbGeom gm; int chunks, threads, gx;
    int rc = dwb5_geom(gm, chunks, threads, gx, N, H, W, C);
    if (rc) return rc;
    dim3 grid(gx, chunks), block(threads);
    hipStream_t st = (hipStream_t)stream;
    const int am = act == MNY_ACT_NONE ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
    const int xf = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : (in_act == MNY_ACT_RELU6 ? 1 : 3));
    const size_t lds = (size_t)kDw5Consts * gm.cgb * sizeof(v2f);
#define MNY_L5(A_, X_)                                                                                                                              \
    do {                                                                                                                                            \
        if (in_red) hipLaunchKernelGGL((dw_bnbwd_s2k5_kernel<T, A_, X_, true>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale,  \
                                       in_shift, in_act, w, addend, dx, ws, gm, in_mean, in_invstd, in_red);                                        \
        else hipLaunchKernelGGL((dw_bnbwd_s2k5_kernel<T, A_, X_, false>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale,        \
                                in_shift, in_act, w, addend, dx, ws, gm, in_mean, in_invstd, in_red);                                               \
    } while (0)
    switch (am * 4 + xf) {
        case 0: MNY_L5(0, 0); break; case 1: MNY_L5(0, 1); break; case 2: MNY_L5(0, 2); break; case 3: MNY_L5(0, 3); break;
        case 4: MNY_L5(1, 0); break; case 5: MNY_L5(1, 1); break; case 6: MNY_L5(1, 2); break; case 7: MNY_L5(1, 3); break;
        case 8: MNY_L5(2, 0); break; case 9: MNY_L5(2, 1); break; case 10: MNY_L5(2, 2); break; default: MNY_L5(2, 3); break;
    }
#undef MNY_L5
    rc = check_launch("dw_bnbwd_s2k5_kernel");
    if (rc || !dw) return rc;
    return launch_reduce_parts(ws, gx, C * 25, dw, st);
}

static int dwb_geom(DwbGeom& g, CgLayout& L, int& gx, int N, int H, int W, int C) {
    MNY_REQUIRE(C % 4 == 0 && C > 0, "dw_bnbwd: C=%d must be a positive multiple of 4", C);
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "dw_bnbwd: empty tensor");
    g.N = N; g.H = H; g.W = W; g.C = C; g.dz = 0;
    static const int th = getenv("MNY_DWB_TH") ? atoi(getenv("MNY_DWB_TH")) : 32;   // strip height (3 halo rows per strip): 16 -> 32: 5.32 -> 5.22 ms
    const int ns = (int)cdiv(H, th);
    g.TH = (int)cdiv(H, ns);
    g.nHS = (int)cdiv(H, g.TH);
    g.nstrips = (int64_t)N * W * g.nHS;
    L = make_stencil_layout(C);                            // <= 64 channel groups per block: >= 4 columns, <= 16 KB of LDS constants
    g.cg_total = L.cg_total; g.cgb = L.cgb;
    int64_t want = cdiv(g.nstrips, L.ppb);
    static const int res1 = getenv("MNY_DWB_RES") ? atoi(getenv("MNY_DWB_RES")) : 768;
    int cap = res1 / L.chunks > 0 ? res1 / L.chunks : 1;       // 3 resident workgroups per CU (<= 168 VGPRs): one full wave of blocks
    static const int xcd_env = getenv("MNY_DWB_XCD") ? atoi(getenv("MNY_DWB_XCD")) : 1;
    g.xcd = xcd_env;
    if (cap > 8) cap &= ~7;
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < cap ? want : cap);
    return MNY_OK;
}

template <typename T>
static int dw_bnbwd_impl(const T* g, const T* y, const float* scale, const float* shift, int act, const float* coef,
                         const T* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                         const T* addend, T* dx, float* dw, float* ws, int N, int H, int W, int C, int K, int stride, void* stream,
                         const float* in_mean = nullptr, const float* in_invstd = nullptr, float* in_red = nullptr, bool store_dz = false) {
    MNY_REQUIRE(g && y && scale && shift && coef && x && w && dx && ws, "dw_bnbwd: null pointer");
    MNY_REQUIRE(!store_dz || (in_red && !dwt_use(K, sizeof(T) == 2 ? 1 : 0, 1, C)), "dw_bnbwd_red_dz: register form with producer sums only (mny_dw_bnbwd_red_dz_supported)");
    MNY_REQUIRE(!in_red || (in_mean && in_invstd && in_scale && in_shift), "dw_bnbwd_red: the input must be a BN unit's raw output (scale, shift, mean, invstd)");
    MNY_REQUIRE(stride == 1 && (K == 3 || (K == 5 && dwt_use(5, 0, 0, C))), "dw_bnbwd: only 3x3 / 5x5 stride 1 are fused (got K=%d stride=%d); use bn_bwd_apply + dw_bwd_*", K, stride);
    MNY_REQUIRE(act != MNY_ACT_HSIGMOID && in_act != MNY_ACT_HSIGMOID, "dw_bnbwd: h-sigmoid views are not supported");
    if (dwt_use(K, sizeof(T) == 2 ? 1 : 0, in_red != nullptr ? 1 : 0, C))      // tile form (dwtile.hip): every 5x5 unit, the 3x3 units it is faster on
        return dwt_launch(sizeof(T) == 2 ? 1 : 0, g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, dw, ws, N, H, W, C, K, stream,
                          in_mean, in_invstd, in_red);
    DwbGeom gm; CgLayout L; int gx;
    int rc = dwb_geom(gm, L, gx, N, H, W, C);
    if (rc) return rc;
    gm.dz = store_dz ? 1 : 0;
    dim3 grid(gx, L.chunks), block(L.threads);
    hipStream_t st = (hipStream_t)stream;
    const int am = act == MNY_ACT_NONE ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
    const int xf = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : (in_act == MNY_ACT_RELU6 ? 1 : 3));
    const size_t lds = (size_t)(18 * L.cgb + (in_red ? 3 * 256 : 0)) * sizeof(float4);
#define MNY_L(A_, X_) do { if (in_red) hipLaunchKernelGGL((dw_bnbwd_s1k3_kernel<T, A_, X_, true>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, \
                                         in_act, w, addend, dx, ws, in_mean, in_invstd, in_red, gm); \
        else hipLaunchKernelGGL((dw_bnbwd_s1k3_kernel<T, A_, X_, false>), grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, \
                                         in_act, w, addend, dx, ws, in_mean, in_invstd, in_red, gm); } while (0)
    switch (am * 4 + xf) {
        case 0: MNY_L(0, 0); break; case 1: MNY_L(0, 1); break; case 2: MNY_L(0, 2); break; case 3: MNY_L(0, 3); break;
        case 4: MNY_L(1, 0); break; case 5: MNY_L(1, 1); break; case 6: MNY_L(1, 2); break; case 7: MNY_L(1, 3); break;
        case 8: MNY_L(2, 0); break; case 9: MNY_L(2, 1); break; case 10: MNY_L(2, 2); break; default: MNY_L(2, 3); break;
    }
#undef MNY_L
    rc = check_launch("dw_bnbwd_s1k3_kernel");
    if (rc || !dw) return rc;                       // dw == NULL: partials only (combined later by mny_reduce_batch)
    return launch_reduce_parts(ws, gx, C * 9, dw, st);
}

}  // namespace mny

using namespace mny;

extern "C" int mny_dw_bnbwd_supported(int K, int stride) { return (stride == 1 && (K == 3 || (K == 5 && dwt_use(5, 0, 0, 0)))) ? 1 : 0; }

extern "C" int mny_dw_bnbwd_s2_parts(int N, int H, int W, int C) {
    DwbGeom g; CgLayout L; int gx;
    if (dwb2_geom(g, L, gx, N, H, W, C)) return MNY_EINVAL;
    return gx;
}
extern "C" int mny_dw_bnbwd_s2(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                               const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                               const float* addend, float* dx, float* dw, float* ws, int N, int H, int W, int C, void* stream) {
    return dw_bnbwd_s2_impl<float>(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, dw, ws, N, H, W, C, stream);
}
extern "C" int mny_dw_bnbwd_s2_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                                    const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                                    const void* addend, void* dx, float* dw, float* ws, int N, int H, int W, int C, void* stream) {
    return dw_bnbwd_s2_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, (const bf16_t*)x, in_scale, in_shift, in_act, w,
                                    (const bf16_t*)addend, (bf16_t*)dx, dw, ws, N, H, W, C, stream);
}

// 5x5 stride-2 unit (fp32 / bf16 storage): ws [mny_dw_bnbwd_s2k5_parts()][C*25]; in_red != NULL: + the producer's BN-backward sums [parts][2][C]
extern "C" int mny_dw_bnbwd_s2k5_parts(int N, int H, int W, int C) {
    DwbGeom g; int chunks, threads, gx;
    if (dwb5_geom(g, chunks, threads, gx, N, H, W, C)) return MNY_EINVAL;
    return gx;
}
extern "C" int mny_dw_bnbwd_s2k5(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                 const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                 const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C,
                                 void* stream) {
    return dw_bnbwd_s2k5_impl<float>(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, in_mean, in_invstd, w, addend, dx, dw, ws, in_red,
                                     N, H, W, C, stream);
}
extern "C" int mny_dw_bnbwd_s2k5_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                                      const void* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                      const float* w, const void* addend, void* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C,
                                      void* stream) {
    return dw_bnbwd_s2k5_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, (const bf16_t*)x, in_scale, in_shift, in_act, in_mean,
                                      in_invstd, w, (const bf16_t*)addend, (bf16_t*)dx, dw, ws, in_red, N, H, W, C, stream);
}

extern "C" int mny_dw_bnbwd_parts_k(int N, int H, int W, int C, int K, int flags) {
    if (K != 3 && K != 5) return MNY_EINVAL;
    if (dwt_use(K, flags & 1, (flags >> 1) & 1, C)) return dwt_parts(N, H, W, C, K, flags & 1);
    if (K != 3) return MNY_EINVAL;
    DwbGeom g; CgLayout L; int gx;
    if (dwb_geom(g, L, gx, N, H, W, C)) return MNY_EINVAL;
    return gx;
}
extern "C" int mny_dw_bnbwd_parts(int N, int H, int W, int C) { return mny_dw_bnbwd_parts_k(N, H, W, C, 3, 0); }     // (fp32 storage, 3x3: the register form)

extern "C" int mny_dw_bnbwd(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                            const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                            const float* addend, float* dx, float* dw, float* ws, int N, int H, int W, int C, int K, int stride,
                            void* stream) {
    return dw_bnbwd_impl<float>(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, dw, ws, N, H, W, C, K, stride, stream);
}

extern "C" int mny_dw_bnbwd_red(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C,
                                int K, int stride, void* stream) {
    MNY_REQUIRE(in_red && in_mean && in_invstd, "dw_bnbwd_red: null pointer");
    return dw_bnbwd_impl<float>(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, dw, ws, N, H, W, C, K, stride, stream,
                                in_mean, in_invstd, in_red);
}

extern "C" int mny_dw_bnbwd_s2_red_dz(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                      const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                      const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C,
                                      void* stream) {
    return dw_bnbwd_s2_red_dz_impl(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, in_mean, in_invstd, w, addend, dx, dw, ws, in_red, N, H, W, C, stream);
}
// the same, storing  in_scale o dX o act_P'(in_scale x + in_shift)  instead of dX: the producer unit P runs the low-rank BN backward (csrc/lrbwd.hip)
extern "C" int mny_dw_bnbwd_red_dz_supported(int K, int C, int bf16) { return (K == 3 && !dwt_use(3, bf16 ? 1 : 0, 1, C)) ? 1 : 0; }
extern "C" int mny_dw_bnbwd_red_dz(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                   const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                                   const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red, int N, int H, int W, int C,
                                   int K, int stride, void* stream) {
    MNY_REQUIRE(in_red && in_mean && in_invstd, "dw_bnbwd_red_dz: null pointer");
    return dw_bnbwd_impl<float>(g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, dw, ws, N, H, W, C, K, stride, stream,
                                in_mean, in_invstd, in_red, true);
}

extern "C" int mny_dw_bnbwd_red_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                                     const void* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                                     const float* in_invstd, const float* w, const void* addend, void* dx, float* dw, float* ws, float* in_red,
                                     int N, int H, int W, int C, int K, int stride, void* stream) {
    MNY_REQUIRE(in_red && in_mean && in_invstd, "dw_bnbwd_red: null pointer");
    return dw_bnbwd_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, (const bf16_t*)x, in_scale, in_shift, in_act, w,
                                 (const bf16_t*)addend, (bf16_t*)dx, dw, ws, N, H, W, C, K, stride, stream, in_mean, in_invstd, in_red);
}

extern "C" int mny_dw_bnbwd_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                                 const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                                 const void* addend, void* dx, float* dw, float* ws, int N, int H, int W, int C, int K, int stride,
                                 void* stream) {
    return dw_bnbwd_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, (const bf16_t*)x, in_scale, in_shift, in_act, w,
                                 (const bf16_t*)addend, (bf16_t*)dx, dw, ws, N, H, W, C, K, stride, stream);
}
