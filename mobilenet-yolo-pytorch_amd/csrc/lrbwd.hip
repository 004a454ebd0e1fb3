// Low-rank BatchNorm backward of a WIDE expand unit: 1x1 conv K -> C (C >= 2K) + BatchNorm + activation whose output gradient arrives
// already multiplied by act'(z) and by ca = gamma * invstd (the depthwise backward in front of it stores  dzc = ca o G o act'(z)  instead
// of G: mny_dw_bnbwd_red_dz).  autograd of nn.Conv2d(K, 6K, 1) + nn.BatchNorm2d + ReLU6 (models/mobilenetv2.py:73-78) and of
// conv1 + bn1 + nolinear1 (models/mobilenetv3.py:49-51,67).
//
// Un-fused, the unit's backward runs  bn_bwd_apply : dY = ca o dz + cb o Y + cc  (read G, Y, write dY: three C-wide passes) and feeds dY to the
// weight- and the data-gradient GEMMs.  With Y = X W^T (X the viewed K-wide input, W [C][K]) everything except the `ca o dz` term is
// LOW RANK in the thin X:
//     dX = dY W          = dzc W  +  X Q + r          Q = W^T diag(cb) W  [K][K],  r = cc^T W  [K]
//     dW = dY^T X        = dzc^T X  +  cb o (W G) + cc (x) s      G = X^T X  [K][K],  s = colsum(X)  [K]
// so both GEMMs run on the stored dzc (no apply pass, no dY tensor) and the BatchNorm terms are two K-wide corrections:
//     mny_lr_gram   : G, s of the viewed input (one read of the thin X; fp32 matrix cores: the loaded dword is both operands) -> partial rows
//     mny_lr_prep   : after mny_bn_bwd_finalize: Q (symmetric: it is its own NT operand) and r, on the fp32 matrix cores
//     mny_pw_lr_fix : dX += X Q + r  (X through its linear view in the GEMM's A-fragment read; + the BN-backward sums of the unit in front)   [pwgemm.hip]
//     mny_lr_wfix   : dW += cb o (W G) + cc (x) s   in place, after the partial combine
// Same algebra as the thin expand units' mny_pw_bnbwd (pw_bnbwd_finalize_kernel), for K = 64 ... 320 where its one-workgroup finalize and
// K <= 32 stream kernels do not reach.
#include "common.h"
#include "x6.h"

namespace mny {

// ---- Gram matrix + column sums of the viewed input --------------------------------------------------------------------------------
// grid (gx, TJ): block (b, ti) owns rows [b * rpb, (b+1) * rpb) and the 32-row band ti of G; its four waves take every fourth pixel
// pair.  v_mfma_f32_32x32x2_f32: A[i][k] = X[pixel k][32 ti + i], B[k][j] = X[pixel k][32 tj + j] — lane (half = pixel of the pair,
// l32 = channel) loads ONE dword per operand, coalesced 128-byte runs, no LDS on the way in.
template <typename T, int TJ>
__global__ __launch_bounds__(256) void lr_gram_kernel(const T* __restrict__ x, const float* __restrict__ xs, const float* __restrict__ xt, int act,
                                                      float* __restrict__ parts, int64_t M, int K, int64_t rpb) {
    extern __shared__ float red[];                      // [32][32 * TJ] band of G, then [4][32 * TJ] column sums
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int ti = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    const int ci = ti * 32 + l32;
    const bool oki = ci < K;
    const float si = oki ? (xs ? xs[ci] : 1.f) : 0.f, hi_ = (oki && xt) ? xt[ci] : 0.f;
    float sj[TJ], hj[TJ];
    int cj[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
        const int c = t * 32 + l32;
        const bool ok = c < K;
        cj[t] = ok ? c : 0;
        sj[t] = ok ? (xs ? xs[c] : 1.f) : 0.f;          // a channel past K: scale = shift = 0 -> act(0) = 0 for every activation of the family
        hj[t] = (ok && xt) ? xt[c] : 0.f;
    }
    f32x16 acc[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float cs[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) cs[t] = 0.f;
    // a wave takes groups of U consecutive pixel pairs, every fourth group; the NEXT group's loads are requested before this group's products.
    // G is symmetric: band ti forms only the tiles on and right of the diagonal (the mirror image is written below).  The band index is a
    // COMPILE-TIME constant of the loop body (one copy per band, selected once): run-time `t >= ti` tests around the loads and MFMAs of the
    // unrolled tile loop cost more than the skipped tiles saved (K = 64: 31 -> 44 us).
    constexpr int U = TJ <= 3 ? 4 : 2;
    const int64_t npairs = (r1 - r0 + 1) / 2;
    const int64_t ngroups = (npairs + U - 1) / U;
    auto band = [&](auto tic) {
        constexpr int TI = decltype(tic)::value;
        float ra[U], rb[U][TJ];
        auto load = [&](int64_t gidx, float (&a)[U], float (&b)[U][TJ]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int64_t row = r0 + 2 * (gidx * U + u) + half;
                if (row >= r1) row = r1 - 1;             // clamped (a valid address); masked where it is used
                const T* xr = x + row * K;
                a[u] = ld1(xr + (oki ? ci : 0));
#pragma unroll
                for (int t = TI; t < TJ; ++t) b[u][t] = ld1(xr + cj[t]);
            }
        };
        int64_t g = wv;
        if (g < ngroups) load(g, ra, rb);
        for (; g < ngroups; g += 4) {
            float na[U], nb[U][TJ];
            const bool more = g + 4 < ngroups;
            if (more) load(g + 4, na, nb);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool rok = r0 + 2 * (g * U + u) + half < r1;
                const float a = rok ? act_fwd(fmaf(ra[u], si, hi_), act) : 0.f;
#pragma unroll
                for (int t = TI; t < TJ; ++t) {
                    const float b = rok ? act_fwd(fmaf(rb[u][t], sj[t], hj[t]), act) : 0.f;
                    cs[t] += b;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ra[u] = na[u];
#pragma unroll
                    for (int t = TI; t < TJ; ++t) rb[u][t] = nb[u][t];
                }
            }
        }
    };
    if constexpr (TJ <= 5) {
        static_for<0, TJ>([&](auto tic) { if (ti == decltype(tic)::value) band(tic); });
    } else {
        band(std::integral_constant<int, 0>{});          // (K > 160: every tile of every band, no mirror writes)
    }
    // fixed-order sum of the four waves' bands through LDS
    constexpr int BW = 32 * TJ;
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < TJ; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;        // accumulator row of the 32x32 layout
                    float* d = red + i * BW + t * 32 + l32;
                    *d = (w == 0 ? 0.f : *d) + acc[t][r];
                }
        }
        __syncthreads();
    }
    float* prow = parts + (int64_t)blockIdx.x * ((int64_t)K * K + K);
    for (int e = tid; e < 32 * BW; e += 256) {           // the band from the diagonal tile on, and its mirror image below the diagonal
        const int i = e / BW, j = e - i * BW;
        if ((TJ <= 5 && j < ti * 32) || ti * 32 + i >= K || j >= K) continue;
        prow[(int64_t)(ti * 32 + i) * K + j] = red[e];
        if (TJ <= 5 && j >= (ti + 1) * 32) prow[(int64_t)j * K + ti * 32 + i] = red[e];
    }
    if (ti == 0) {                                       // column sums (every tile column is live in band 0): the two pixels of a pair, then the four waves
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            const float v = cs[t] + __shfl_xor(cs[t], 32);
            if (half == 0) red[wv * BW + t * 32 + l32] = v;
        }
        __syncthreads();
        for (int j = tid; j < BW; j += 256)
            if (j < K) prow[(int64_t)K * K + j] = (red[j] + red[BW + j]) + (red[2 * BW + j] + red[3 * BW + j]);
    }
}

static int lr_gram_gx(int64_t M) {                    // partial rows: <= 256 row slices of >= 256 rows
    int64_t g = cdiv(M, 256);
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}

template <typename T>
static int lr_gram_launch(const T* x, const float* xs, const float* xt, int act, float* parts, int64_t M, int K, hipStream_t st) {
    const int TJ = (K + 31) / 32, gx = lr_gram_gx(M);
    int64_t rpb = cdiv(M, gx);
    rpb += rpb & 1;                                      // whole pixel pairs per block
    const dim3 grid(gx, TJ), block(256);
    const size_t lds = (size_t)32 * 32 * TJ * sizeof(float);
#define MNY_G(J) hipLaunchKernelGGL((lr_gram_kernel<T, J>), grid, block, lds, st, x, xs, xt, act, parts, M, K, rpb)
    switch (TJ) {
        case 1: MNY_G(1); break; case 2: MNY_G(2); break; case 3: MNY_G(3); break; case 4: MNY_G(4); break; case 5: MNY_G(5); break;
        case 6: MNY_G(6); break; case 7: MNY_G(7); break; case 8: MNY_G(8); break; case 9: MNY_G(9); break; default: MNY_G(10); break;
    }
#undef MNY_G
    return check_launch("lr_gram_kernel");
}

// ---- Q, r from the finalized coefficients -------------------------------------------------------------------------------------------
// Q = W^T diag(cb) W is a C-deep contraction with K x K outputs: block (ti, tj) < T*T forms one 32x32 tile on the fp32 matrix cores
// (A[i][n] = cb[n] W[n][32 ti + i], B[n][j] = W[n][32 tj + j]: the lane's dword of a W row is the operand), its four waves take every
// fourth channel pair and meet in LDS; the last block forms r = cc^T W with sixteen channel slices per column.  (The first version ran
// fp64 chains over all C channels per thread: 30-110 us per unit on the critical path of the backward pass.)
__global__ __launch_bounds__(1024) void lr_prep_kernel(const float* __restrict__ coef, const float* __restrict__ W, float* __restrict__ Q,
                                                       float* __restrict__ r, int C, int K) {
    __shared__ float red[32][33];
    __shared__ float rs[16][65];
    const int T = (K + 31) / 32, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;      // 16 waves: the contraction is latency-bound, not work-bound
    const float* cb = coef + C;
    const float* cc = coef + 2 * C;
    if ((int)blockIdx.x < T * T) {
        const int ti = (int)blockIdx.x / T, tj = (int)blockIdx.x % T;
        const int half = lane >> 5, l32 = lane & 31;
        const int ci = ti * 32 + l32, cj = tj * 32 + l32;
        const bool oki = ci < K, okj = cj < K;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int npairs = (C + 1) / 2;
        const int cic = oki ? ci : 0, cjc = okj ? cj : 0;
        for (int q = wv; q < npairs; q += 64) {             // four channel pairs per trip, their loads issued together
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = 2 * (q + 16 * u) + half;
                const bool nok = n < C;
                const int nn = nok ? n : 0;
                const float wa = W[(int64_t)nn * K + cic], wb = W[(int64_t)nn * K + cjc], c = cb[nn];
                a[u] = (nok && oki) ? c * wa : 0.f;
                b[u] = (nok && okj) ? wb : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        }
        for (int w = 0; w < 16; ++w) {                      // fixed-order sum of the waves
            if (wv == w) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (e & 3) + 8 * (e >> 2) + 4 * half;
                    red[i][l32] = (w == 0 ? 0.f : red[i][l32]) + acc[e];
                }
            }
            __syncthreads();
        }
        {
            const int i = tid >> 5, j = tid & 31;
            if (ti * 32 + i < K && tj * 32 + j < K) Q[(int64_t)(ti * 32 + i) * K + tj * 32 + j] = red[i][j];
        }
        return;
    }
    // r[j] = sum_n cc[n] W[n][j]: 64 columns x 16 channel slices at a time (four loads in flight per thread), slices summed in a fixed order
    const int jl = tid & 63, sl = tid >> 6;
    for (int j0 = 0; j0 < K; j0 += 64) {
        const int j = j0 + jl;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (j < K)
            for (int n = sl; n < C; n += 64) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int m = n + 16 * u;
                    if (m < C) a[u] = fmaf(cc[m], W[(int64_t)m * K + j], a[u]);
                }
            }
        __syncthreads();
        rs[sl][jl] = (a[0] + a[1]) + (a[2] + a[3]);
        __syncthreads();
        if (tid < 64 && j0 + tid < K) {
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) v += rs[u][tid];
            r[j0 + tid] = v;
        }
    }
}

// ---- dW += cb o (W G) + cc (x) s, in place ------------------------------------------------------------------------------------------
// gs = [K*K Gram | K column sums] of the viewed input (combined partial rows of mny_lr_gram); 16x16 tiles of dW, fp64 chains
__global__ __launch_bounds__(256) void lr_wfix_kernel(float* __restrict__ dW, const float* __restrict__ gs, const float* __restrict__ coef,
                                                      const float* __restrict__ W, int C, int K) {
    __shared__ float sW[16][17], sG[16][17];
    const int TK = (K + 15) / 16, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int n0 = ((int)blockIdx.x / TK) * 16, k0 = ((int)blockIdx.x % TK) * 16;
    double a = 0.0;
    for (int j0 = 0; j0 < K; j0 += 16) {
        __syncthreads();
        sW[ty][tx] = (n0 + ty < C && j0 + tx < K) ? W[(int64_t)(n0 + ty) * K + j0 + tx] : 0.f;
        sG[ty][tx] = (j0 + ty < K && k0 + tx < K) ? gs[(int64_t)(j0 + ty) * K + k0 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) a += (double)sW[ty][j] * (double)sG[j][tx];
    }
    const int n = n0 + ty, k = k0 + tx;
    if (n < C && k < K) {
        const int64_t e = (int64_t)n * K + k;
        dW[e] = (float)((double)dW[e] + (double)coef[C + n] * a + (double)coef[2 * C + n] * (double)gs[(int64_t)K * K + k]);
    }
}

}  // namespace mny

using namespace mny;

// which (M, K, C) the low-rank form serves: a thin input (K a multiple of 8, <= 320: the Gram band fits LDS) feeding an output at least twice as wide
extern "C" int mny_lr_supported(int64_t M, int K, int C) {
    static const bool off = getenv("MNY_NO_LR") != nullptr && atoi(getenv("MNY_NO_LR")) != 0;
    return (!off && M > 0 && K >= 16 && K <= 320 && (K & 7) == 0 && C >= 2 * K && C <= 4096) ? 1 : 0;
}
extern "C" int mny_lr_gram_parts(int64_t M, int K) { return (M <= 0 || K <= 0 || K > 320) ? MNY_EINVAL : lr_gram_gx(M); }

extern "C" int mny_lr_gram(const float* x, const float* in_scale, const float* in_shift, int in_act, float* parts, int64_t M, int K, void* stream) {
    MNY_REQUIRE(x && parts && M > 0 && K > 0 && K <= 320, "lr_gram: bad arguments (K=%d)", K);
    MNY_REQUIRE(in_act >= MNY_ACT_NONE && in_act <= MNY_ACT_HSIGMOID && (!in_scale) == (!in_shift), "lr_gram: bad view");
    return lr_gram_launch<float>(x, in_scale, in_shift, in_act, parts, M, K, (hipStream_t)stream);
}
extern "C" int mny_lr_gram_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, float* parts, int64_t M, int K, void* stream) {
    MNY_REQUIRE(x && parts && M > 0 && K > 0 && K <= 320, "lr_gram: bad arguments (K=%d)", K);
    MNY_REQUIRE(in_act >= MNY_ACT_NONE && in_act <= MNY_ACT_HSIGMOID && (!in_scale) == (!in_shift), "lr_gram: bad view");
    return lr_gram_launch<bf16_t>((const bf16_t*)x, in_scale, in_shift, in_act, parts, M, K, (hipStream_t)stream);
}

extern "C" int mny_lr_prep(const float* coef, const float* w, float* q, float* r, int C, int K, void* stream) {
    MNY_REQUIRE(coef && w && q && r && C > 0 && K > 0, "lr_prep: bad arguments");
    const int T = (K + 31) / 32;
    hipLaunchKernelGGL(lr_prep_kernel, dim3(T * T + 1), dim3(1024), 0, (hipStream_t)stream, coef, w, q, r, C, K);
    return check_launch("lr_prep_kernel");
}

extern "C" int mny_lr_wfix(float* dw, const float* gram_sums, const float* coef, const float* w, int C, int K, void* stream) {
    MNY_REQUIRE(dw && gram_sums && coef && w && C > 0 && K > 0, "lr_wfix: bad arguments");
    const int TK = (K + 15) / 16, TC = (C + 15) / 16;
    hipLaunchKernelGGL(lr_wfix_kernel, dim3(TC * TK), dim3(256), 0, (hipStream_t)stream, dw, gram_sums, coef, w, C, K);
    return check_launch("lr_wfix_kernel");
}
