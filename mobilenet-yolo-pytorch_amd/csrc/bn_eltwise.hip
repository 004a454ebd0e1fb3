// BatchNorm (training + eval) split as "statistics in the producer's epilogue, apply in the consumer's
// load path", plus the elementwise glue of the network (residual add, nearest-2x upsample, axpy).
// All kernels are HBM-bound float4 streams in the shared channel-group thread layout; all reductions
// are two-stage and deterministic (per-block partial rows, fp64 combine).
//
// replaces nn.BatchNorm2d (models/mobilenetv2.py:41-84, models/mbv2_yolo.py:23), ReLU6 / LeakyReLU
// backward, `x + conv(x)` (mobilenetv2.py:89), torch.add (mbv2_yolo.py:103,151) and nn.Upsample (:52).
#include "common.h"

namespace mny {

// fp64 sum of one channel's share (rows slice, slice + 32, ...) of the per-block partial rows [parts][2][C].  The additions run in a fixed
// order — four accumulators taking rows round-robin, the rows after the last full group of four on the first — and the loads of
// sixteen rows are issued before the first addition: with one group in flight the loop was 6 serial HBM/L2 latencies at 768 rows
// (5.7 us per launch, 131 launches per step).
__device__ __forceinline__ void sum_partial_rows(const float* __restrict__ rows, int parts, int C, int c, int slice, double& s, double& q) {
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    s = 0.0; q = 0.0;
    int p = slice;
    const int64_t step = (int64_t)64 * C;                         // 32 rows of 2C floats
    const float* a = rows + (int64_t)p * 2 * C + c;
    for (; p + 480 < parts; p += 512, a += 16 * step) {
        float v[16], u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { v[i] = a[i * step]; u[i] = a[i * step + C]; }
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            s += (double)v[i];      q += (double)u[i];
            s1 += (double)v[i + 1]; q1 += (double)u[i + 1];
            s2 += (double)v[i + 2]; q2 += (double)u[i + 2];
            s3 += (double)v[i + 3]; q3 += (double)u[i + 3];
        }
    }
    for (; p + 96 < parts; p += 128, a += 4 * step) {
        const float v0 = a[0], u0 = a[C], v1 = a[step], u1 = a[step + C], v2 = a[2 * step], u2 = a[2 * step + C], v3 = a[3 * step], u3 = a[3 * step + C];
        s += (double)v0;  q += (double)u0;
        s1 += (double)v1; q1 += (double)u1;
        s2 += (double)v2; q2 += (double)u2;
        s3 += (double)v3; q3 += (double)u3;
    }
    for (; p < parts; p += 32, a += step) {
        s += (double)a[0];
        q += (double)a[C];
    }
    s = (s + s1) + (s2 + s3);
    q = (q + q1) + (q2 + q3);
}

// ---- forward statistics -> scale/shift -----------------------------------------------------------
// block = 32 channels x 8 slices of the partial rows
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ stats, int parts, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* running_mean, float* running_var,
                                                          float* scale, float* shift, float* mean_out, float* invstd_out, int C) {
    __shared__ double red[2][32][32];
    const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        sum_partial_rows(stats, parts, C, c, slice, s, q);
    }
    red[0][slice][cl] = s;
    red[1][slice][cl] = q;
    __syncthreads();
    if (slice == 0 && c < C) {
        s = 0.0; q = 0.0;
        for (int i = 0; i < 32; ++i) { s += red[0][i][cl]; q += red[1][i][cl]; }
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        if (mean_out) mean_out[c] = (float)mean;
        if (invstd_out) invstd_out[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---- backward ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ red_out, int64_t M, int C, int cgb, int cg_total) {
    __shared__ float4 red[256 * 2];
    const int tid = threadIdx.x;
    const int cgl = tid % cgb, pix = tid / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    const bool cvalid = cg < cg_total;
    const int c = cg * 4;
    float4 s1 = f4zero(), s2 = f4zero();
    if (cvalid) {
        const float4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c), is = ld4(invstd + c);
        auto accum = [&](float4 gv, float4 yv) {
            float4 dz;
            dz.x = gv.x * act_bwd(fmaf(yv.x, sc.x, sh.x), act);
            dz.y = gv.y * act_bwd(fmaf(yv.y, sc.y, sh.y), act);
            dz.z = gv.z * act_bwd(fmaf(yv.z, sc.z, sh.z), act);
            dz.w = gv.w * act_bwd(fmaf(yv.w, sc.w, sh.w), act);
            add4(s1, dz);
            s2.x = fmaf(dz.x, (yv.x - mu.x) * is.x, s2.x);
            s2.y = fmaf(dz.y, (yv.y - mu.y) * is.y, s2.y);
            s2.z = fmaf(dz.z, (yv.z - mu.z) * is.z, s2.z);
            s2.w = fmaf(dz.w, (yv.w - mu.w) * is.w, s2.w);
        };
        const int64_t stride = (int64_t)gridDim.x * ppb;
        int64_t m = (int64_t)blockIdx.x * ppb + pix;
        // 16-bit storage: a lane's load is 8 bytes, so the loop is bound by loads in flight, not bytes — keep four rows' worth
        // of loads outstanding (fp32 is already at the HBM ceiling with one)
        constexpr int U = sizeof(T) == 2 ? 4 : 1;
        if (U > 1) {
            for (; m + (U - 1) * stride < M; m += U * stride) {
                float4 gv[U], yv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { gv[u] = ld4(g + (m + u * stride) * C + c); yv[u] = ld4(y + (m + u * stride) * C + c); }
#pragma unroll
                for (int u = 0; u < U; ++u) accum(gv[u], yv[u]);
            }
        }
        for (; m < M; m += stride) accum(ld4(g + m * C + c), ld4(y + m * C + c));
    }
    red[tid * 2] = s1;
    red[tid * 2 + 1] = s2;
    __syncthreads();
    if (pix == 0 && cvalid) {
        float4 a = f4zero(), b = f4zero();
        for (int p = 0; p < ppb; ++p) { add4(a, red[(p * cgb + cgl) * 2]); add4(b, red[(p * cgb + cgl) * 2 + 1]); }
        st4(red_out + (int64_t)blockIdx.x * 2 * C + c, a);
        st4(red_out + (int64_t)blockIdx.x * 2 * C + C + c, b);
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ red_in, int parts, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                              float* coef, int C) {
    __shared__ double red[2][32][32];
    const int cl = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        sum_partial_rows(red_in, parts, C, c, slice, s, q);
    }
    red[0][slice][cl] = s;
    red[1][slice][cl] = q;
    __syncthreads();
    if (slice == 0 && c < C) {
        s = 0.0; q = 0.0;
        for (int i = 0; i < 32; ++i) { s += red[0][i][cl]; q += red[1][i][cl]; }
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        // dy = a*(dz - s/M - yhat*q/M),  yhat = (y-mu)*invstd   ->   dy = ca*dz + cb*y + cc
        const double a = (double)gamma[c] * (double)invstd[c];
        const double b = -a * (double)invstd[c] * q / count;
        coef[c] = (float)a;
        coef[C + c] = (float)b;
        coef[2 * C + c] = (float)(-a * s / count - b * (double)mean[c]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                           const float* __restrict__ coef, T* __restrict__ dy,
                                                           int64_t M, int C, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    const float4 sc = scale ? ld4(scale + c) : f4one(), sh = scale ? ld4(shift + c) : f4zero();
    const float4 ca = coef ? ld4(coef + c) : f4one();
    const float4 cb = coef ? ld4(coef + C + c) : f4zero();
    const float4 cc = coef ? ld4(coef + 2 * C + c) : f4zero();
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        const float4 gv = ld4(g + m * C + c), yv = ld4(y + m * C + c);
        float4 o;
        o.x = fmaf(ca.x, gv.x * act_bwd(fmaf(yv.x, sc.x, sh.x), act), fmaf(cb.x, yv.x, cc.x));
        o.y = fmaf(ca.y, gv.y * act_bwd(fmaf(yv.y, sc.y, sh.y), act), fmaf(cb.y, yv.y, cc.y));
        o.z = fmaf(ca.z, gv.z * act_bwd(fmaf(yv.z, sc.z, sh.z), act), fmaf(cb.z, yv.z, cc.z));
        o.w = fmaf(ca.w, gv.w * act_bwd(fmaf(yv.w, sc.w, sh.w), act), fmaf(cb.w, yv.w, cc.w));
        st4_stream(dy + m * C + c, o);
    }
}

// ---- residual add / upsample ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void add_views_kernel(const T* __restrict__ a, const float* __restrict__ a_scale,
                                                        const float* __restrict__ a_shift, int a_act,
                                                        const T* __restrict__ b, const float* __restrict__ b_scale,
                                                        const float* __restrict__ b_shift, int b_act,
                                                        const T* __restrict__ up, T* __restrict__ out,
                                                        int N, int H, int W, int C, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    const float4 asc = a_scale ? ld4(a_scale + c) : f4one(), ash = a_scale ? ld4(a_shift + c) : f4zero();
    const float4 bsc = b_scale ? ld4(b_scale + c) : f4one(), bsh = b_scale ? ld4(b_shift + c) : f4zero();
    const int64_t M = (int64_t)N * H * W;
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        float4 o = xform4(ld4(a + m * C + c), asc, ash, a_act);
        if (b) add4(o, xform4(ld4(b + m * C + c), bsc, bsh, b_act));
        if (up) {
            const int wi = (int)(m % W), hi = (int)((m / W) % H);
            const int64_t n = m / ((int64_t)W * H);
            add4(o, ld4(up + ((n * (H / 2) + hi / 2) * (W / 2) + wi / 2) * C + c));
        }
        st4_stream(out + m * C + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ src, T* __restrict__ dst, int accumulate,
                                                           int N, int H, int W, int C, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    const int Hh = H / 2, Wh = W / 2;
    const int64_t M = (int64_t)N * Hh * Wh;
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        const int wi = (int)(m % Wh), hi = (int)((m / Wh) % Hh);
        const int64_t n = m / ((int64_t)Wh * Hh);
        const T* s = src + ((n * H + 2 * hi) * W + 2 * wi) * C + c;
        float4 o = accumulate ? ld4(dst + m * C + c) : f4zero();
        add4(o, ld4(s)); add4(o, ld4(s + C));
        add4(o, ld4(s + (int64_t)W * C)); add4(o, ld4(s + (int64_t)W * C + C));
        st4(dst + m * C + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void axpy_kernel(const T* __restrict__ src, const float* __restrict__ alpha,
                                                   T* __restrict__ dst, int accumulate, int64_t n4, int64_t n) {
    const float a = alpha ? alpha[0] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = ld4(src + i * 4);
        float4 d = accumulate ? ld4(dst + i * 4) : f4zero();
        d.x = fmaf(a, v.x, d.x); d.y = fmaf(a, v.y, d.y); d.z = fmaf(a, v.z, d.z); d.w = fmaf(a, v.w, d.w);
        st4(dst + i * 4, d);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        st1(dst + i, (accumulate ? ld1(dst + i) : 0.f) + a * ld1(src + i));
    }
}


// ---- scalar-channel variants (C not a multiple of 4: the 10-channel SE bottleneck of MobileNetV3) ---------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_c1_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               float* __restrict__ red_out, int64_t M, int C, int cb) {
    __shared__ float red[256 * 2];
    const int tid = threadIdx.x;
    const int cl = tid % cb, pix = tid / cb, ppb = blockDim.x / cb;
    const int c = blockIdx.y * cb + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
            const float yv = ld1(y + m * C + c);
            const float dz = ld1(g + m * C + c) * act_bwd(fmaf(yv, sc, sh), act);
            s1 += dz;
            s2 = fmaf(dz, (yv - mu) * is, s2);
        }
    }
    red[tid * 2] = s1; red[tid * 2 + 1] = s2;
    __syncthreads();
    if (pix == 0 && c < C) {
        float a = 0.f, b = 0.f;
        for (int p = 0; p < ppb; ++p) { a += red[(p * cb + cl) * 2]; b += red[(p * cb + cl) * 2 + 1]; }
        red_out[(int64_t)blockIdx.x * 2 * C + c] = a;
        red_out[(int64_t)blockIdx.x * 2 * C + C + c] = b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_c1_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                              const float* __restrict__ coef, T* __restrict__ dy, int64_t total, int C) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
        const float ca = coef ? coef[c] : 1.f, cb = coef ? coef[C + c] : 0.f, cc = coef ? coef[2 * C + c] : 0.f;
        const float yv = ld1(y + e);
        st1(dy + e, fmaf(ca, ld1(g + e) * act_bwd(fmaf(yv, sc, sh), act), fmaf(cb, yv, cc)));
    }
}

// ---- per-pixel gate (MobileNetV3 "SE") and PartAdd --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mul_views_kernel(const T* __restrict__ a, const float* __restrict__ a_scale,
                                                        const float* __restrict__ a_shift, int a_act,
                                                        const T* __restrict__ b, const float* __restrict__ b_scale,
                                                        const float* __restrict__ b_shift, int b_act,
                                                        const T* __restrict__ addend, T* __restrict__ out,
                                                        int64_t M, int C, int cgb, int cg_total, int bwd) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    const float4 asc = a_scale ? ld4(a_scale + c) : f4one(), ash = a_scale ? ld4(a_shift + c) : f4zero();
    const float4 bsc = b_scale ? ld4(b_scale + c) : f4one(), bsh = b_scale ? ld4(b_shift + c) : f4zero();
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        // forward: a and b are both views; backward (bwd=1): a is the raw upstream gradient, b the other operand's view
        const float4 x = bwd ? ld4(a + m * C + c) : xform4(ld4(a + m * C + c), asc, ash, a_act);
        const float4 w = xform4(ld4(b + m * C + c), bsc, bsh, b_act);
        float4 o = addend ? ld4(addend + m * C + c) : f4zero();
        fma4(o, x, w);
        st4(out + m * C + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void partadd_up_kernel(const T* __restrict__ a, const float* __restrict__ a_scale,
                                                         const float* __restrict__ a_shift, int a_act,
                                                         const T* __restrict__ up, T* __restrict__ out,
                                                         int N, int H, int W, int Ca, int Cb, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    const bool in_a = c < Ca;
    const float4 asc = (a_scale && in_a) ? ld4(a_scale + c) : f4one(), ash = (a_scale && in_a) ? ld4(a_shift + c) : f4zero();
    const int64_t M = (int64_t)N * H * W;
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        const int wi = (int)(m % W), hi = (int)((m / W) % H);
        const int64_t n = m / ((int64_t)W * H);
        float4 o = ld4(up + ((n * (H / 2) + hi / 2) * (W / 2) + wi / 2) * Cb + c);
        if (in_a) add4(o, xform4(ld4(a + m * Ca + c), asc, ash, a_act));
        st4(out + m * Cb + c, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void slice_channels_kernel(const T* __restrict__ src, T* __restrict__ dst, int accumulate,
                                                             int64_t M, int Ca, int Cb, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    for (int64_t m = (int64_t)blockIdx.x * ppb + pix; m < M; m += (int64_t)gridDim.x * ppb) {
        float4 o = accumulate ? ld4(dst + m * Ca + c) : f4zero();
        add4(o, ld4(src + m * Cb + c));
        st4(dst + m * Ca + c, o);
    }
}

// ---- storage conversion at the fp32 <-> bf16 boundary (detection heads, which stay fp32) -------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cvt_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t n4, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        st4(dst + i * 4, ld4(src + i * 4));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) st1(dst + n4 * 4 + threadIdx.x, ld1(src + n4 * 4 + threadIdx.x));
}

// partial rows of bn_bwd_reduce (mny_bn_bwd_parts()).  2048 (all 8 resident workgroups per CU) was measured slower:
// 4.87 -> 5.05 ms/step for the reduce itself plus +0.6 ms in the finalize that sums the rows
constexpr int kBnBwdParts = 1024;

// rows [M][C] fp32 (times *alpha) -> [M][Cp] of T with zeroed pad columns: gives the 75-channel head gradients 16-B aligned rows
template <typename T>
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, const float* __restrict__ alpha, T* __restrict__ dst,
                                                       int64_t total, int C, int Cp) {
    const float a = alpha ? alpha[0] : 1.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / Cp;
        const int c = (int)(e - m * Cp);
        st1(dst + e, c < C ? a * src[m * C + c] : 0.f);
    }
}

static inline dim3 rows_grid(int64_t M, const CgLayout& L, int cap) {
    int64_t want = cdiv(M, L.ppb);
    return dim3((unsigned)(want < cap ? want : cap), L.chunks);
}

}  // namespace mny

using namespace mny;

extern "C" int mny_bn_finalize(const float* stats, int parts, int64_t count, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                               float* mean, float* invstd, int C, void* stream) {
    MNY_REQUIRE(stats && gamma && beta && scale && shift && parts > 0 && count > 0 && C > 0, "bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, stats, parts, (double)count,
                       gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd, C);
    return check_launch("bn_finalize_kernel");
}

extern "C" int mny_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  float eps, float* scale, float* shift, int C, void* stream) {
    MNY_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bn_eval_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, scale, shift, C);
    return check_launch("bn_eval_coeffs_kernel");
}

// frozen BatchNorm (module in eval mode, gradients wanted: mbv2_yolo.py:157 returns differentiable losses under model.eval()):
// the statistics are the running ones, constants of the step — mean / invstd for the backward kernels' yhat, and a finalize whose
// data-gradient coefficients are (gamma * invstd, 0, 0): dgamma = sum dz * yhat, dbeta = sum dz as in training.
__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, float eps, float* mean_out, float* invstd_out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mean_out[c] = rm[c];
        invstd_out[c] = 1.f / sqrtf(rv[c] + eps);            // the denominator bn_eval_coeffs_kernel divides gamma by
    }
}

extern "C" int mny_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean_out, float* invstd_out, int C,
                                 void* stream) {
    MNY_REQUIRE(running_mean && running_var && mean_out && invstd_out && C > 0, "bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_mean, running_var, eps, mean_out,
                       invstd_out, C);
    return check_launch("bn_eval_stats_kernel");
}

static void c1_layout(int64_t M, int C, int& cb, int& chunks, int& gx) {
    chunks = (int)cdiv(C, 256);
    cb = (int)cdiv(C, chunks);
    const int ppb = 256 / cb > 0 ? 256 / cb : 1;
    int64_t want = cdiv(M, ppb);
    gx = (int)(want < 1024 ? want : 1024);
}

extern "C" int mny_bn_bwd_parts(int64_t M, int C) {
    if (M <= 0 || C <= 0) return MNY_EINVAL;
    if (C % 4) { int cb, ch, gx; c1_layout(M, C, cb, ch, gx); return gx; }
    CgLayout L = make_cg_layout(C);
    return (int)rows_grid(M, L, kBnBwdParts).x;
}

template <typename T>
static int bn_bwd_reduce_impl(const T* g, const T* y, const float* scale, const float* shift, int act,
                                 const float* mean, const float* invstd, float* red, int64_t M, int C, void* stream) {
    MNY_REQUIRE(g && y && scale && shift && mean && invstd && red, "bn_bwd_reduce: null pointer");
    MNY_REQUIRE(M > 0 && C > 0, "bn_bwd_reduce: bad shape M=%lld C=%d", (long long)M, C);
    if (C % 4) {
        int cb, ch, gx; c1_layout(M, C, cb, ch, gx);
        const int ppb = 256 / cb > 0 ? 256 / cb : 1;
        hipLaunchKernelGGL((bn_bwd_reduce_c1_kernel<T>), dim3(gx, ch), dim3(cb * ppb), 0, (hipStream_t)stream, g, y, scale, shift, act, mean, invstd, red, M, C, cb);
        return check_launch("bn_bwd_reduce_c1_kernel");
    }
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), rows_grid(M, L, kBnBwdParts), dim3(L.threads), 0, (hipStream_t)stream, g, y, scale, shift, act,
                       mean, invstd, red, M, C, L.cgb, L.cg_total);
    return check_launch("bn_bwd_reduce_kernel");
}
extern "C" int mny_bn_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift, int act,
                                 const float* mean, const float* invstd, float* red, int64_t M, int C, void* stream) {
    return bn_bwd_reduce_impl<float>(g, y, scale, shift, act, mean, invstd, red, M, C, stream);
}
extern "C" int mny_bn_bwd_reduce_bf16(const void* g, const void* y, const float* scale, const float* shift, int act,
                                 const float* mean, const float* invstd, float* red, int64_t M, int C, void* stream) {
    return bn_bwd_reduce_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, mean, invstd, red, M, C, stream);
}

extern "C" int mny_bn_bwd_finalize(const float* red, int parts, int64_t count, const float* gamma, const float* mean,
                                   const float* invstd, float* dgamma, float* dbeta, float* coef, int C, void* stream) {
    MNY_REQUIRE(red && gamma && mean && invstd && dgamma && dbeta && coef && parts > 0 && C > 0, "bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, red, parts, (double)count,
                       gamma, mean, invstd, dgamma, dbeta, coef, C);
    return check_launch("bn_bwd_finalize_kernel");
}

extern "C" int mny_bn_bwd_finalize_frozen(const float* red, int parts, int64_t count, const float* gamma, const float* mean,
                                          const float* invstd, float* dgamma, float* dbeta, float* coef, int C, void* stream) {
    (void)count;                                   // same argument list as mny_bn_bwd_finalize; the statistics do not depend on the batch
    MNY_REQUIRE(red && gamma && mean && invstd && dgamma && dbeta && coef && parts > 0 && C > 0, "bn_bwd_finalize_frozen: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, red, parts, (double)INFINITY,
                       gamma, mean, invstd, dgamma, dbeta, coef, C);              // count = inf: cb = -0, cc = -0 exactly
    return check_launch("bn_bwd_finalize_kernel");
}

template <typename T>
static int bn_bwd_apply_impl(const T* g, const T* y, const float* scale, const float* shift, int act, const float* coef,
                                T* dy, int64_t M, int C, void* stream) {
    MNY_REQUIRE(g && y && dy, "bn_bwd_apply: null pointer");
    MNY_REQUIRE(M > 0 && C > 0, "bn_bwd_apply: bad shape");
    if (C % 4) {
        int64_t blocks = cdiv(M * C, 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((bn_bwd_apply_c1_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, y, scale, shift, act, coef, dy, M * C, C);
        return check_launch("bn_bwd_apply_c1_kernel");
    }
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), rows_grid(M, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, g, y, scale, shift, act,
                       coef, dy, M, C, L.cgb, L.cg_total);
    return check_launch("bn_bwd_apply_kernel");
}
extern "C" int mny_bn_bwd_apply(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                                float* dy, int64_t M, int C, void* stream) {
    return bn_bwd_apply_impl<float>(g, y, scale, shift, act, coef, dy, M, C, stream);
}
extern "C" int mny_bn_bwd_apply_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                                void* dy, int64_t M, int C, void* stream) {
    return bn_bwd_apply_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)y, scale, shift, act, coef, (bf16_t*)dy, M, C, stream);
}

template <typename T>
static int add_views_impl(const T* a, const float* a_scale, const float* a_shift, int a_act, const T* b,
                             const float* b_scale, const float* b_shift, int b_act, const T* up, T* out, int N, int H,
                             int W, int C, void* stream) {
    MNY_REQUIRE(a && out, "add_views: null pointer");
    MNY_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "add_views: bad shape");
    MNY_REQUIRE(!up || (H % 2 == 0 && W % 2 == 0), "add_views: upsample operand needs even H,W");
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((add_views_kernel<T>), rows_grid((int64_t)N * H * W, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, a, a_scale,
                       a_shift, a_act, b, b_scale, b_shift, b_act, up, out, N, H, W, C, L.cgb, L.cg_total);
    return check_launch("add_views_kernel");
}
extern "C" int mny_add_views(const float* a, const float* a_scale, const float* a_shift, int a_act, const float* b,
                             const float* b_scale, const float* b_shift, int b_act, const float* up, float* out, int N, int H,
                             int W, int C, void* stream) {
    return add_views_impl<float>(a, a_scale, a_shift, a_act, b, b_scale, b_shift, b_act, up, out, N, H, W, C, stream);
}
extern "C" int mny_add_views_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b,
                             const float* b_scale, const float* b_shift, int b_act, const void* up, void* out, int N, int H,
                             int W, int C, void* stream) {
    return add_views_impl<bf16_t>((const bf16_t*)a, a_scale, a_shift, a_act, (const bf16_t*)b, b_scale, b_shift, b_act, (const bf16_t*)up, (bf16_t*)out, N, H, W, C, stream);
}

template <typename T>
static int upsample_bwd_impl(const T* src, T* dst, int accumulate, int N, int H, int W, int C, void* stream) {
    MNY_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "upsample_bwd: bad arguments");
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((upsample_bwd_kernel<T>), rows_grid((int64_t)N * (H / 2) * (W / 2), L, 8192), dim3(L.threads), 0, (hipStream_t)stream,
                       src, dst, accumulate, N, H, W, C, L.cgb, L.cg_total);
    return check_launch("upsample_bwd_kernel");
}
extern "C" int mny_upsample_bwd(const float* src, float* dst, int accumulate, int N, int H, int W, int C, void* stream) {
    return upsample_bwd_impl<float>(src, dst, accumulate, N, H, W, C, stream);
}
extern "C" int mny_upsample_bwd_bf16(const void* src, void* dst, int accumulate, int N, int H, int W, int C, void* stream) {
    return upsample_bwd_impl<bf16_t>((const bf16_t*)src, (bf16_t*)dst, accumulate, N, H, W, C, stream);
}

template <typename T>
static int axpy_impl(const T* src, const float* alpha, T* dst, int accumulate, int64_t n, void* stream) {
    MNY_REQUIRE(src && dst && n > 0, "axpy: bad arguments");
    const int64_t n4 = n / 4;
    int64_t blocks = cdiv(n4 > 0 ? n4 : 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((axpy_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, alpha, dst, accumulate, n4, n);
    return check_launch("axpy_kernel");
}
extern "C" int mny_axpy(const float* src, const float* alpha, float* dst, int accumulate, int64_t n, void* stream) {
    return axpy_impl<float>(src, alpha, dst, accumulate, n, stream);
}
extern "C" int mny_axpy_bf16(const void* src, const float* alpha, void* dst, int accumulate, int64_t n, void* stream) {
    return axpy_impl<bf16_t>((const bf16_t*)src, alpha, (bf16_t*)dst, accumulate, n, stream);
}

template <typename T>
static int mul_views_impl(const T* a, const float* a_scale, const float* a_shift, int a_act, const T* b,
                             const float* b_scale, const float* b_shift, int b_act, T* out, int64_t M, int C, void* stream) {
    MNY_REQUIRE(a && b && out && M > 0 && C > 0 && C % 4 == 0, "mul_views: bad arguments");
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((mul_views_kernel<T>), rows_grid(M, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, a, a_scale, a_shift, a_act, b, b_scale,
                       b_shift, b_act, (const T*)nullptr, out, M, C, L.cgb, L.cg_total, 0);
    return check_launch("mul_views_kernel");
}
extern "C" int mny_mul_views(const float* a, const float* a_scale, const float* a_shift, int a_act, const float* b,
                             const float* b_scale, const float* b_shift, int b_act, float* out, int64_t M, int C, void* stream) {
    return mul_views_impl<float>(a, a_scale, a_shift, a_act, b, b_scale, b_shift, b_act, out, M, C, stream);
}
extern "C" int mny_mul_views_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b,
                             const float* b_scale, const float* b_shift, int b_act, void* out, int64_t M, int C, void* stream) {
    return mul_views_impl<bf16_t>((const bf16_t*)a, a_scale, a_shift, a_act, (const bf16_t*)b, b_scale, b_shift, b_act, (bf16_t*)out, M, C, stream);
}

template <typename T>
static int mul_views_bwd_impl(const T* g, const T* o, const float* o_scale, const float* o_shift, int o_act,
                                 const T* addend, T* dst, int64_t M, int C, void* stream) {
    MNY_REQUIRE(g && o && dst && M > 0 && C > 0 && C % 4 == 0, "mul_views_bwd: bad arguments");
    CgLayout L = make_cg_layout(C);
    hipLaunchKernelGGL((mul_views_kernel<T>), rows_grid(M, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, g, (const float*)nullptr, (const float*)nullptr, MNY_ACT_NONE, o,
                       o_scale, o_shift, o_act, addend, dst, M, C, L.cgb, L.cg_total, 1);
    return check_launch("mul_views_kernel(bwd)");
}
extern "C" int mny_mul_views_bwd(const float* g, const float* o, const float* o_scale, const float* o_shift, int o_act,
                                 const float* addend, float* dst, int64_t M, int C, void* stream) {
    return mul_views_bwd_impl<float>(g, o, o_scale, o_shift, o_act, addend, dst, M, C, stream);
}
extern "C" int mny_mul_views_bwd_bf16(const void* g, const void* o, const float* o_scale, const float* o_shift, int o_act,
                                 const void* addend, void* dst, int64_t M, int C, void* stream) {
    return mul_views_bwd_impl<bf16_t>((const bf16_t*)g, (const bf16_t*)o, o_scale, o_shift, o_act, (const bf16_t*)addend, (bf16_t*)dst, M, C, stream);
}

template <typename T>
static int partadd_up_impl(const T* a, const float* a_scale, const float* a_shift, int a_act, const T* up, T* out,
                              int N, int H, int W, int Ca, int Cb, void* stream) {
    MNY_REQUIRE(a && up && out && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "partadd_up: bad arguments");
    MNY_REQUIRE(Ca > 0 && Ca <= Cb && Ca % 4 == 0 && Cb % 4 == 0, "partadd_up: need Ca <= Cb, both multiples of 4");
    CgLayout L = make_cg_layout(Cb);
    hipLaunchKernelGGL((partadd_up_kernel<T>), rows_grid((int64_t)N * H * W, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, a, a_scale, a_shift,
                       a_act, up, out, N, H, W, Ca, Cb, L.cgb, L.cg_total);
    return check_launch("partadd_up_kernel");
}
extern "C" int mny_partadd_up(const float* a, const float* a_scale, const float* a_shift, int a_act, const float* up, float* out,
                              int N, int H, int W, int Ca, int Cb, void* stream) {
    return partadd_up_impl<float>(a, a_scale, a_shift, a_act, up, out, N, H, W, Ca, Cb, stream);
}
extern "C" int mny_partadd_up_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* up, void* out,
                              int N, int H, int W, int Ca, int Cb, void* stream) {
    return partadd_up_impl<bf16_t>((const bf16_t*)a, a_scale, a_shift, a_act, (const bf16_t*)up, (bf16_t*)out, N, H, W, Ca, Cb, stream);
}

template <typename T>
static int slice_channels_impl(const T* src, T* dst, int accumulate, int64_t M, int Ca, int Cb, void* stream) {
    MNY_REQUIRE(src && dst && M > 0 && Ca > 0 && Ca <= Cb && Ca % 4 == 0 && Cb % 4 == 0, "slice_channels: bad arguments");
    CgLayout L = make_cg_layout(Ca);
    hipLaunchKernelGGL((slice_channels_kernel<T>), rows_grid(M, L, 8192), dim3(L.threads), 0, (hipStream_t)stream, src, dst, accumulate, M, Ca, Cb,
                       L.cgb, L.cg_total);
    return check_launch("slice_channels_kernel");
}
extern "C" int mny_slice_channels(const float* src, float* dst, int accumulate, int64_t M, int Ca, int Cb, void* stream) {
    return slice_channels_impl<float>(src, dst, accumulate, M, Ca, Cb, stream);
}
extern "C" int mny_slice_channels_bf16(const void* src, void* dst, int accumulate, int64_t M, int Ca, int Cb, void* stream) {
    return slice_channels_impl<bf16_t>((const bf16_t*)src, (bf16_t*)dst, accumulate, M, Ca, Cb, stream);
}

extern "C" int mny_cvt_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    MNY_REQUIRE(src && dst && n > 0, "cvt_f32_bf16: bad arguments");
    int64_t blocks = cdiv(n / 4 > 0 ? n / 4 : 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((cvt_kernel<float, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n / 4, n);
    return check_launch("cvt_kernel");
}

// every fp32 -> bf16 weight shadow of a forward pass in ONE launch: block b converts elements [4096*(b - job.block0), +4096) of
// job block_job[b]
__global__ __launch_bounds__(256) void cvt_batch_kernel(const mny_cvt_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    const mny_cvt_job jb = jobs[block_job[blockIdx.x]];
    const int64_t base = (int64_t)((int)blockIdx.x - jb.block0) * 4096;
    bf16_t* dst = (bf16_t*)jb.dst;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < jb.n) st1(dst + i, jb.src[i]);
    }
}
extern "C" int mny_cvt_batch_f32_bf16(const mny_cvt_job* jobs, const int32_t* block_job, int nblocks, void* stream) {
    MNY_REQUIRE(jobs && block_job && nblocks > 0, "cvt_batch: bad arguments");
    hipLaunchKernelGGL(cvt_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs, block_job);
    return check_launch("cvt_batch_kernel");
}

extern "C" int mny_cvt_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
    MNY_REQUIRE(src && dst && n > 0, "cvt_bf16_f32: bad arguments");
    int64_t blocks = cdiv(n / 4 > 0 ? n / 4 : 1, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((cvt_kernel<bf16_t, float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n / 4, n);
    return check_launch("cvt_kernel");
}

template <typename T>
static int pad_rows_impl(const float* src, const float* alpha, T* dst, int64_t M, int C, int Cp, void* stream) {
    MNY_REQUIRE(src && dst && M > 0 && C > 0 && Cp >= C, "pad_rows: bad arguments");
    int64_t blocks = cdiv(M * Cp, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((pad_rows_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, alpha, dst, M * Cp, C, Cp);
    return check_launch("pad_rows_kernel");
}
extern "C" int mny_pad_rows(const float* src, const float* alpha, float* dst, int64_t M, int C, int Cp, void* stream) {
    return pad_rows_impl<float>(src, alpha, dst, M, C, Cp, stream);
}
extern "C" int mny_pad_rows_bf16(const float* src, const float* alpha, void* dst, int64_t M, int C, int Cp, void* stream) {
    return pad_rows_impl<bf16_t>(src, alpha, (bf16_t*)dst, M, C, Cp, stream);
}
