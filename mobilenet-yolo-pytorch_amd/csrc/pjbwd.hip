// Backward of a thin "project" conv unit (1x1 conv Ki -> No + BN, no activation: the linear bottleneck that closes an inverted-residual
// block, models/mobilenetv2.py:69-70,83-84) as ONE pass, for No in {16, 24, 32} and Ki a multiple of 16 up to 192.
//
// Unfused the unit's backward is three launches over the same tensors:
//     mny_bn_bwd_apply    : reads G, Y (M x No)                    -> writes dY (M x No)
//     mny_pw_dgrad_bnred  : reads dY, D (M x Ki, for the BN sums)  -> writes G_d = dY W (M x Ki) + the BN-backward sums of the unit in front
//     mny_pw_wgrad        : reads dY, D                            -> dW = dY^T act(BN(D))
// The wide tensor D (the raw output of the depthwise unit in front) is read twice and the thin dY written once and read twice.  Here a
// wave owns 16-pixel tiles: dY = ca o G + cb o Y + cc goes to LDS (16 x No), D is loaded ONCE in the register layout both matrix
// products want, G_d = dY W leaves as it is formed (with the sums sum dz, sum dz * d_hat, dz = G_d * act'(BN(D)) of the unit in front),
// and dW += dY^T act(BN(D)) accumulates on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products).
//
// One register layout for both products: lane (l16, lg) holds D[pixel 4r + lg][channel 16b + l16] in register (b, r).  The data-gradient
// product runs with its output rows permuted (row i = 4 lg + r of the tile stands for pixel 4 r + lg — its A operand simply reads dY of
// the permuted pixel), so the accumulator element (b, r) of a lane IS the gradient of the D element it holds; and register (b, s) of the
// activated D is the B operand of k-step s of the weight-gradient product (k = pixel 4 s + lg).  No transposes, no second load.
// fp32 storage.  Partial rows: dW [gridDim.x][No * Ki], sums [gridDim.x][2][Ki] (mny_bn_bwd_finalize's layout).
#include "common.h"

namespace mny {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PjArgs {
    const float* g; const float* y; const float* coef;                       // unit output gradient, raw output, (ca, cb, cc)[No]
    const float* d; const float* d_scale; const float* d_shift; const float* d_mean; const float* d_invstd; int d_act;
    const float* w; float* gd; float* dw_parts; float* red; int64_t M;
};

template <int NO, int KI>
struct PjCfg {
    static constexpr int NB = KI / 16, NS = NO / 4, NIB = (NO + 15) / 16, NQ = NO / 4, S = NO + 4;       // S: LDS row stride of the dY tile (conflict-free operand reads)
    static constexpr int DY_F4 = 4 * NO;                                     // float4 elements of a 16 x NO tile
    static constexpr int NPASS = (DY_F4 + 63) / 64;
    static constexpr size_t LDS = (size_t)(NO * KI + 4 * KI + 4 * 16 * S) * sizeof(float);
};

template <int NO, int KI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void pj_bwd_kernel(PjArgs p) {
    using Cf = PjCfg<NO, KI>;
    constexpr int NB = Cf::NB, NS = Cf::NS, NIB = Cf::NIB, NQ = Cf::NQ, S = Cf::S, NPASS = Cf::NPASS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sW = lds;                                   // [NO][KI]
    float* sC = sW + NO * KI;                          // [4][KI]: d_scale, d_shift, d_mean, d_invstd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
    float* sdY = sC + 4 * KI + wave * 16 * S;          // this wave's dY tile [16][S]
    for (int e = tid; e < NO * KI; e += 256) sW[e] = p.w[e];
    for (int e = tid; e < KI; e += 256) { sC[e] = p.d_scale[e]; sC[KI + e] = p.d_shift[e]; sC[2 * KI + e] = p.d_mean[e]; sC[3 * KI + e] = p.d_invstd[e]; }
    __syncthreads();
    const float slope = act_slope(p.d_act), hi = act_hi(p.d_act);
    // the lane's share of the dY tile: float4 element idx = lane + 64 pass -> (pixel, channel quad); BN-backward coefficients of its quad
    int t_px[NPASS], t_q[NPASS];
    float4 ca[NPASS], cb[NPASS], cc[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int idx = lane + 64 * ps;
        const bool ok = idx < Cf::DY_F4;
        t_px[ps] = ok ? idx / NQ : -1;
        t_q[ps] = ok ? idx % NQ : 0;
        ca[ps] = ld4(p.coef + 4 * t_q[ps]); cb[ps] = ld4(p.coef + NO + 4 * t_q[ps]); cc[ps] = ld4(p.coef + 2 * NO + 4 * t_q[ps]);
    }
    const int pix_a = 4 * (l16 & 3) + (l16 >> 2);      // the pixel that row i = l16 of the data-gradient product stands for: pi(4 lg + r) = 4 r + lg

    f32x4 accW[NIB][NB];
    float s1[NB], s2[NB];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int b = 0; b < NB; ++b) accW[ib][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NB; ++b) { s1[b] = 0.f; s2[b] = 0.f; }

    const int64_t ntiles = (p.M + 15) / 16;
    constexpr bool PF = NB <= 6;                       // narrow inputs: the NEXT tile's D is requested while this one is processed (4 NB more registers)
    float dnext[PF ? NB : 1][4];
    auto load_d = [&](int64_t tt, float (&dst)[PF ? NB : 1][4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = tt * 16 + 4 * r + lg;
            const float* row = p.d + (m < p.M ? m : p.M - 1) * KI + l16;
#pragma unroll
            for (int b = 0; b < (PF ? NB : 1); ++b) dst[b][r] = row[16 * b];
        }
    };
    if (PF && (int64_t)blockIdx.x * 4 + wave < ntiles) load_d((int64_t)blockIdx.x * 4 + wave, dnext);
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t m0 = t * 16;
        // ---- D in the shared register layout (requested first: the long loads) ----
        float dreg[NB][4];
        bool pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = m0 + 4 * r + lg;
            pv[r] = m < p.M;
            if (!PF) {
                const float* row = p.d + (pv[r] ? m : p.M - 1) * KI + l16;
#pragma unroll
                for (int b = 0; b < NB; ++b) dreg[b][r] = row[16 * b];
            }
        }
        if (PF) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) dreg[b][r] = dnext[PF ? b : 0][r];
            const int64_t tn = t + (int64_t)gridDim.x * 4;
            if (tn < ntiles) load_d(tn, dnext);
        }
        // ---- dY tile -> LDS ----
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            if (t_px[ps] >= 0) {
                const int64_t m = m0 + t_px[ps];
                float4 v = f4zero();
                if (m < p.M) {
                    const float4 g4 = ld4(p.g + m * NO + 4 * t_q[ps]), y4 = ld4(p.y + m * NO + 4 * t_q[ps]);
                    v.x = fmaf(ca[ps].x, g4.x, fmaf(cb[ps].x, y4.x, cc[ps].x)); v.y = fmaf(ca[ps].y, g4.y, fmaf(cb[ps].y, y4.y, cc[ps].y));
                    v.z = fmaf(ca[ps].z, g4.z, fmaf(cb[ps].z, y4.z, cc[ps].z)); v.w = fmaf(ca[ps].w, g4.w, fmaf(cb[ps].w, y4.w, cc[ps].w));
                }
                *reinterpret_cast<float4*>(&sdY[t_px[ps] * S + 4 * t_q[ps]]) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();               // (LDS operations of one wave execute in order; this only stops the compiler from reordering them)
        // A operand of the data-gradient product: dY[pixel pi(l16)][4 s + lg]
        float aop[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) aop[s] = sdY[pix_a * S + 4 * s + lg];
        // ---- G_d = dY W, the sums of the unit in front, act(BN(D)) in place ----
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[s], sW[(4 * s + lg) * KI + 16 * b + l16], acc, 0, 0, 0);
            const int ch = 16 * b + l16;
            const float dsc = sC[ch], dsh = sC[KI + ch], dmu = sC[2 * KI + ch], dis = sC[3 * KI + ch];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gd = acc[r], dv = dreg[b][r];
                const float z = fmaf(dv, dsc, dsh);
                const float dz = gd * ((z > 0.f ? 1.f : slope) * (z < hi ? 1.f : 0.f));
                if (pv[r]) {
                    s1[b] += dz;
                    s2[b] = fmaf(dz, (dv - dmu) * dis, s2[b]);
                    p.gd[(m0 + 4 * r + lg) * KI + ch] = gd;
                }
                dreg[b][r] = pv[r] ? fminf(fmaxf(z, slope * z), hi) : 0.f;
            }
        }
        // ---- dW += dY^T act(BN(D)): k-step s = pixels 4 s + lg, B operand = register (b, s) ----
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
                const int no = 16 * ib + l16;
                const float a = no < NO ? sdY[(4 * s + lg) * S + (no < NO ? no : 0)] : 0.f;
#pragma unroll
                for (int b = 0; b < NB; ++b) accW[ib][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dreg[b][s], accW[ib][b], 0, 0, 0);
            }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- partial rows: the four waves folded in wave order ----
    __syncthreads();
    float* fold = lds;                                 // [NO][KI] + [2][KI]  (sW / sC are done)
    for (int e = tid; e < NO * KI + 2 * KI; e += 256) fold[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int no = 16 * ib + 4 * lg + r;
                        if (no < NO) fold[no * KI + 16 * b + l16] += accW[ib][b][r];
                    }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float a = s1[b], c2 = s2[b];
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                c2 += __shfl_xor(c2, 16); c2 += __shfl_xor(c2, 32);
                if (lg == 0) { fold[NO * KI + 16 * b + l16] += a; fold[NO * KI + KI + 16 * b + l16] += c2; }
            }
        }
        __syncthreads();
    }
    float* dwp = p.dw_parts + (int64_t)blockIdx.x * NO * KI;
    for (int e = tid; e < NO * KI; e += 256) dwp[e] = fold[e];
    float* rp = p.red + (int64_t)blockIdx.x * 2 * KI;
    for (int e = tid; e < 2 * KI; e += 256) rp[e] = fold[NO * KI + e];
}

static bool pj_shape_ok(int64_t M, int Ki, int No) {
    return M >= 16 && (No == 16 || No == 24 || No == 32) && (Ki == 32 || Ki == 96 || Ki == 144 || Ki == 192);
}
static int pj_grid(int64_t M, int Ki) {
    const int64_t want = cdiv(cdiv(M, 16), 4);
    const int cap = Ki == 32 ? 1024 : 512;             // resident workgroups: 88 VGPRs at Ki = 32 (four per CU), 160-250 above (two per CU)
    int gx = (int)(want < cap ? want : cap);
    return gx < 1 ? 1 : gx;
}

template <int NO, int KI>
static int pj_launch(const PjArgs& a, int grid, hipStream_t st) {
    const size_t lds = PjCfg<NO, KI>::LDS;
    hipLaunchKernelGGL((pj_bwd_kernel<NO, KI>), dim3(grid), dim3(256), lds, st, a);
    return check_launch("pj_bwd_kernel");
}

}  // namespace mny

using namespace mny;

extern "C" int mny_pj_bwd_supported(int64_t M, int Ki, int No, int d_act) {
    static const bool off = getenv("MNY_NO_PJBWD") != nullptr;
    return (!off && pj_shape_ok(M, Ki, No) && d_act != MNY_ACT_HSWISH && d_act != MNY_ACT_HSIGMOID) ? 1 : 0;
}

extern "C" int mny_pj_bwd_parts(int64_t M, int Ki, int No) { return pj_shape_ok(M, Ki, No) ? pj_grid(M, Ki) : MNY_EINVAL; }

extern "C" int mny_pj_bwd(const float* g, const float* y, const float* coef, const float* d, const float* d_scale, const float* d_shift,
                          const float* d_mean, const float* d_invstd, int d_act, const float* w, float* gd, float* dw, float* dw_ws, float* red,
                          int64_t M, int Ki, int No, void* stream) {
    MNY_REQUIRE(g && y && coef && d && d_scale && d_shift && d_mean && d_invstd && w && gd && dw_ws && red, "pj_bwd: null pointer");
    MNY_REQUIRE(pj_shape_ok(M, Ki, No) && d_act != MNY_ACT_HSWISH && d_act != MNY_ACT_HSIGMOID, "pj_bwd: M=%lld Ki=%d No=%d act %d not supported",
                (long long)M, Ki, No, d_act);
    PjArgs a{g, y, coef, d, d_scale, d_shift, d_mean, d_invstd, d_act, w, gd, dw_ws, red, M};
    const int grid = pj_grid(M, Ki);
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define MNY_PJ(NO_, KI_) rc = pj_launch<NO_, KI_>(a, grid, st)
    if (No == 16) { if (Ki == 32) MNY_PJ(16, 32); else if (Ki == 96) MNY_PJ(16, 96); else if (Ki == 144) MNY_PJ(16, 144); else MNY_PJ(16, 192); }
    else if (No == 24) { if (Ki == 32) MNY_PJ(24, 32); else if (Ki == 96) MNY_PJ(24, 96); else if (Ki == 144) MNY_PJ(24, 144); else MNY_PJ(24, 192); }
    else { if (Ki == 32) MNY_PJ(32, 32); else if (Ki == 96) MNY_PJ(32, 96); else if (Ki == 144) MNY_PJ(32, 144); else MNY_PJ(32, 192); }
#undef MNY_PJ
    if (rc || !dw) return rc;                          // dw == NULL: partial rows only (combined later by mny_reduce_batch)
    return launch_reduce_parts(dw_ws, grid, No * Ki, dw, st);
}
