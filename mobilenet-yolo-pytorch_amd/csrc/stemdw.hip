// Backward of "stem conv + BN + act  ->  depthwise 3x3 stride-1 conv + BN + act" (the first two units of MobileNetV2:
// models/mobilenetv2.py:40 conv_3x3_bn(3, 32, 2) and the depthwise half of the first InvertedResidual, :65-67) as ONE pass.
//
// Unfused (round 3) the pair costs mny_dw_bnbwd_red (reads G_d, D, S; writes the stem's output gradient G_s, 1.0 GB at bs 256 / 352^2)
// + mny_bn_bwd_finalize + mny_stem_bnwgrad (reads G_s, S and the image again): 7.4 GB, 1.41 ms.  G_s has ONE consumer — the stem's
// weight gradient — and that gradient is a bilinear form of things this pass already holds:
//     dY_s = ca o dz_s + cb o s + cc        (BN backward of the stem; dz_s = G_s * act'(BN(s)), s = W p, p = the 27-value image patch)
//     dW_s = dY_s^T P = ca o (dz_s^T P) + cb o (W P^T P) + cc (x) colsum(P)
// with ca = gamma * invstd known BEFORE the sums (s1 = sum dz_s, s2 = sum dz_s * s_hat) that cb and cc need.  So the pass forms dz_s per
// pixel, accumulates P1 = dz_s^T P (32 x 27), the patch second-moment matrix P^T P (27 x 27), colsum(P) and (s1, s2) — partial rows in
// exactly the layout of the fused expand-unit backward (pwgemm.hip: [P1 | Gram | s1 | s2 | s3]) — and pw_bnbwd_finalize_kernel (fp64)
// turns them into dW_s, dgamma_s, dbeta_s.  G_s is never written and S is read once: 3.4 GB.
//
// Workgroup = 8 channel groups (4 channels each) x 32 column slots: 30 owned columns + one halo column each side, walking down a strip of
// rows.  A thread loads ONLY its column (G_d, D, S: rows r+1, r+2 in flight while row r is processed), rebuilds the depthwise unit's dY
// once and trades it with its neighbours through an LDS ring of four rows (which is also the depthwise weight gradient's dY history);
// the data gradient of the depthwise conv is the scatter of dwbwd.hip (row r-1 complete after row r).  The finished row of dz_s and the
// three image rows of its patches go to LDS and the matrix cores take the bilinear sums: the loaded dword of the image IS both operands
// of the patch second-moment product (v_mfma_f32_16x16x4_f32: A[i][k] = P[k][i], B[k][j] = P[k][j], k = the pixel).
// fp32 storage, Cout = 32, activations of the ReLU / ReLU6 / leaky family (no h-swish: MobileNetV3's stem keeps the unfused kernels).
#include "common.h"

namespace mny {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only (exdw.hip's): __syncthreads() also waits for every outstanding global load of the wave
// (s_waitcnt vmcnt(0)) — here that is the rows requested two iterations ahead, i.e. the prefetch would be drained at every row.
// Nothing crosses waves through global memory inside the row loop.
__device__ __forceinline__ void sd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int SD_C = 32, SD_CG = 8, SD_SLOTS = 32, SD_OWN = 30, SD_K = 27, SD_DZS = 48;     // SD_DZS: floats per pixel row of the dz_s tile (bank-conflict-free MFMA reads)
constexpr int SD_IMGW = 68;                                                                 // staged image row: 2 * 32 + 1 columns, padded

struct SdGeom {
    int N, H, W, Ho, Wo;
    int TH, nHS, nWT;
    int64_t ntiles;
};

struct SdArgs {
    const float* gd; const float* d; const float* d_scale; const float* d_shift; int d_act; const float* d_coef;
    const float* s; const float* s_scale; const float* s_shift; const float* s_mean; const float* s_invstd; int s_act;
    const float* x; const float* w_dw; float* dw_parts; float* partial; SdGeom g;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void stemdw_bwd_kernel(SdArgs p) {
    __shared__ float4 cst[18 * SD_CG];                       // 9 taps, d scale / shift, ca, cb, cc, s scale / shift / mean / invstd, (pad)
    __shared__ float4 ring[4][256];                          // dY rows of the depthwise unit
    __shared__ __attribute__((aligned(16))) float dzs[2][SD_SLOTS * SD_DZS];   // the finished row of dz_s: [slot][channel], double-buffered:
    __shared__ float img[2][3 * 3 * SD_IMGW];                // image rows of that row's patches: [ci][kh][column]    its sums run one row behind
    __shared__ float4 red[256 * 2];
    const SdGeom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
    const int cgl = tid & (SD_CG - 1), slot = tid >> 3;
    const int c = 4 * cgl;
    if (slot == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
            cst[t * SD_CG + cgl] = make_float4(p.w_dw[(c + 0) * 9 + t], p.w_dw[(c + 1) * 9 + t], p.w_dw[(c + 2) * 9 + t], p.w_dw[(c + 3) * 9 + t]);
        cst[9 * SD_CG + cgl] = ld4(p.d_scale + c);  cst[10 * SD_CG + cgl] = ld4(p.d_shift + c);
        cst[11 * SD_CG + cgl] = ld4(p.d_coef + c);  cst[12 * SD_CG + cgl] = ld4(p.d_coef + SD_C + c);  cst[13 * SD_CG + cgl] = ld4(p.d_coef + 2 * SD_C + c);
        cst[14 * SD_CG + cgl] = ld4(p.s_scale + c); cst[15 * SD_CG + cgl] = ld4(p.s_shift + c);
        cst[16 * SD_CG + cgl] = ld4(p.s_mean + c);  cst[17 * SD_CG + cgl] = ld4(p.s_invstd + c);
    }
    __syncthreads();
    const float dslope = act_slope(p.d_act), dhi = act_hi(p.d_act);
    const float sslope = act_slope(p.s_act), shi = act_hi(p.s_act);
    auto dact = [&](float z) { return (z > 0.f ? 1.f : dslope) * (z < dhi ? 1.f : 0.f); };
    auto dy2 = [&](v2f gv, v2f yv, v2f s, v2f h, v2f a, v2f b, v2f cterm) {
        const v2f z = __builtin_elementwise_fma(yv, s, h);
        const v2f d = gv * v2f{dact(z.x), dact(z.y)};
        return __builtin_elementwise_fma(a, d, __builtin_elementwise_fma(b, yv, cterm));
    };

    // matrix-core accumulators of this wave: P1 blocks (channel block a, patch block b) and the patch second-moment blocks (0,0) (0,1) (1,1)
    f32x4 accP[2][2], accG[3];
    float cs[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) accP[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 3; ++b) accG[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the lane's two patch elements t = l16, 16 + l16 -> (ci, kh, kw) -> offset inside the staged image rows; t >= 27 does not exist
    int poff[2];
    bool pok[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int t = 16 * b + l16;
        pok[b] = t < SD_K;
        const int tc = pok[b] ? t : 0;
        const int ci = tc / 9, kh = (tc % 9) / 3, kw = tc % 3;
        poff[b] = (ci * 3 + kh) * SD_IMGW + kw;
    }

    F4P wp[9], s1 = f4p0(), s2 = f4p0();
#pragma unroll
    for (int t = 0; t < 9; ++t) wp[t] = f4p0();
    const int gxd = gridDim.x;
    const int lb = ((gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t rowpitch = (int64_t)g.Wo * SD_C;
    const int64_t plane = (int64_t)g.H * g.W;
    // image staging: element e = tid + 256 u of the 9 x 65 block (row rr = ci * 3 + kh, column col) — all of it thread-constant
    int st_off[3], st_col[3], st_kh[3], st_lds[3];
    bool st_ok[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = tid + 256 * u;
        const int rr = e / 65, col = e - rr * 65;
        st_ok[u] = rr < 9;
        st_kh[u] = rr % 3;
        st_col[u] = col;
        st_off[u] = (int)((rr / 3) * plane) + st_kh[u] * g.W + col;       // + (2 hp - 1) * W + wi0: the (row, tile) part
        st_lds[u] = rr * SD_IMGW + col;
    }
    // this wave's eight pixels (slots 8 wave .. 8 wave + 7) of a finished row: two k-steps of four
    auto mma_row = [&](int buf, int w0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = 8 * wave + 4 * ks + lg;
            const bool owned = px >= 1 && px <= SD_OWN && w0 + px - 1 < g.Wo;
            const float a0 = dzs[buf][px * SD_DZS + l16], a1 = dzs[buf][px * SD_DZS + 16 + l16];
            float b0 = img[buf][poff[0] + 2 * px], b1 = img[buf][poff[1] + 2 * px];
            b0 = (owned && pok[0]) ? b0 : 0.f;
            b1 = (owned && pok[1]) ? b1 : 0.f;
            cs[0] += b0; cs[1] += b1;
            accP[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, accP[0][0], 0, 0, 0);
            accP[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, accP[0][1], 0, 0, 0);
            accP[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, accP[1][0], 0, 0, 0);
            accP[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, accP[1][1], 0, 0, 0);
            accG[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, b0, accG[0], 0, 0, 0);
            accG[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, b1, accG[1], 0, 0, 0);
            accG[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, b1, accG[2], 0, 0, 0);
        }
    };
    for (int64_t tile = lb; tile < g.ntiles; tile += gxd) {                     // workgroup-uniform: every thread meets every barrier
        const int wt = (int)(tile % g.nWT);
        const int hs = (int)((tile / g.nWT) % g.nHS);
        const int n = (int)(tile / ((int64_t)g.nWT * g.nHS));
        const int h0 = hs * g.TH;
        const int h1 = min(h0 + g.TH, g.Ho);
        const int w0 = wt * SD_OWN;
        const int wo = w0 + slot - 1;                                           // slot 0 / 31: the halo columns
        const bool colok = wo >= 0 && wo < g.Wo;
        const bool inter = slot >= 1 && slot <= SD_OWN && colok;                // this thread owns column wo
        const float cokf = colok ? 1.f : 0.f;
        const int64_t colbase = (int64_t)n * g.Ho * rowpitch + (int64_t)min(max(wo, 0), g.Wo - 1) * SD_C + c;
        const float* xn = p.x + (int64_t)n * 3 * plane;
        const int wi0 = 2 * (w0 - 1) - 1;                                       // image column of staged column 0

        float4 g0, y0, x0, g1, y1, x1;
        auto ldrow = [&](int r, float4& gq, float4& yq, float4& xq) {
            const int64_t o = colbase + (int64_t)min(max(r, 0), g.Ho - 1) * rowpitch;
            gq = ld4(p.gd + o); yq = ld4(p.d + o); xq = ld4(p.s + o);
        };
        ldrow(h0 - 1, g0, y0, x0);
        ldrow(h0, g1, y1, x1);

        F4P P0 = f4p0(), P1 = f4p0();
        F4P aprev = f4p0();
        float4 spark = f4zero();                                                // raw stem output of row r-1 at this column
        for (int r = h0 - 1; r <= h1; ++r) {
            int lo = cgl;
            asm volatile("" : "+v"(lo));                       // opaque per iteration: keeps the LDS constant reads IN the loop
            const float4* my = cst + lo;
            const float4 gc = g0, yc = y0, xc4 = x0;
            g0 = g1; y0 = y1; x0 = x1;
            if (r + 2 <= h1) ldrow(r + 2, g1, y1, x1);
            // image rows of stem row r-1 (its patches): 3 channels x 3 rows x 65 columns, zeros outside the image
            const int hp = r - 1;
            const bool mrow = hp >= h0;                         // row r-1 is a finished, owned row of dz_s (hp < h1 by the loop bound)
            float iv[3];
            if (mrow) {
                const float* xr = xn + (int64_t)(2 * hp - 1) * g.W + wi0;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int hi = 2 * hp - 1 + st_kh[u], wi = wi0 + st_col[u];
                    iv[u] = (st_ok[u] && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? xr[st_off[u]] : 0.f;
                }
            }
            const float rokf = (r >= 0 && r < g.Ho) ? 1.f : 0.f;
            const F4P sc = f4p(my[9 * SD_CG]), sh = f4p(my[10 * SD_CG]), ca = f4p(my[11 * SD_CG]), cb = f4p(my[12 * SD_CG]), cc = f4p(my[13 * SD_CG]);
            F4P dyo;
            {
                const float m = rokf * cokf;                   // dY is 0 outside the image
                const v2f m2 = v2f{m, m};
                const F4P G4 = f4p(gc), Y4 = f4p(yc);
                dyo.lo = dy2(G4.lo, Y4.lo, sc.lo, sh.lo, ca.lo, cb.lo, cc.lo) * m2;
                dyo.hi = dy2(G4.hi, Y4.hi, sc.hi, sh.hi, ca.hi, cb.hi, cc.hi) * m2;
            }
            float4* row = &ring[r & 3][0];
            row[tid] = f4u(dyo);
            sd_barrier();
            if (r - 2 >= h0) mma_row((r - 1) & 1, w0);          // the bilinear sums of the row finished one iteration ago (buffers written before this barrier)
            float4 dz4 = f4zero();
            if (inter) {
                F4P dyr[3];
                dyr[0] = f4p(row[tid - SD_CG]);
                dyr[1] = dyo;
                dyr[2] = f4p(row[tid + SD_CG]);
                const F4P xsc = f4p(my[14 * SD_CG]), xsh = f4p(my[15 * SD_CG]);
                // activated stem output of row r at this column (the depthwise conv's input), counted once: by the strip that owns the row
                F4P ar;
                {
                    const float own = (r >= h0 && r < h1) ? 1.f : 0.f;
                    const F4P X4 = f4p(xc4);
                    const v2f z0 = __builtin_elementwise_fma(X4.lo, xsc.lo, xsh.lo), z1 = __builtin_elementwise_fma(X4.hi, xsc.hi, xsh.hi);
                    const v2f t0 = z0 * v2f{sslope, sslope}, t1 = z1 * v2f{sslope, sslope};
                    ar.lo = v2f{fminf(fmaxf(z0.x, t0.x), shi), fminf(fmaxf(z0.y, t0.y), shi)} * v2f{own, own};
                    ar.hi = v2f{fminf(fmaxf(z1.x, t1.x), shi), fminf(fmaxf(z1.y, t1.y), shi)} * v2f{own, own};
                }
                F4P P2 = f4p0();
#define WG(t) f4p(my[(t) * SD_CG])
#pragma unroll
                for (int kq = 0; kq < 3; ++kq) {
                    pfma(P0, dyr[2 - kq], WG(0 + kq));
                    pfma(P1, dyr[2 - kq], WG(3 + kq));
                    pfma(P2, dyr[2 - kq], WG(6 + kq));
                }
#undef WG
                if (mrow) {                                     // G_s of row r-1 is complete: dz_s, the stem's sums
                    const float4 gs = f4u(P0);
                    const float4 mu = my[16 * SD_CG], is = my[17 * SD_CG];
                    auto pact = [&](float sv, float s_, float h_) {
                        const float z = fmaf(sv, s_, h_);
                        return (z > 0.f ? 1.f : sslope) * (z < shi ? 1.f : 0.f);
                    };
                    dz4.x = gs.x * pact(spark.x, xsc.lo.x, xsh.lo.x); dz4.y = gs.y * pact(spark.y, xsc.lo.y, xsh.lo.y);
                    dz4.z = gs.z * pact(spark.z, xsc.hi.x, xsh.hi.x); dz4.w = gs.w * pact(spark.w, xsc.hi.y, xsh.hi.y);
                    const F4P dzp = f4p(dz4);
                    s1.lo += dzp.lo; s1.hi += dzp.hi;
                    const v2f h0v = (v2f{spark.x, spark.y} - v2f{mu.x, mu.y}) * v2f{is.x, is.y};
                    const v2f h1v = (v2f{spark.z, spark.w} - v2f{mu.z, mu.w}) * v2f{is.z, is.w};
                    s2.lo = __builtin_elementwise_fma(dzp.lo, h0v, s2.lo); s2.hi = __builtin_elementwise_fma(dzp.hi, h1v, s2.hi);
                }
                spark = xc4;
                P0 = P1; P1 = P2;
                if (r > h0) {                                   // aprev = stem row r-1 >= h0 (before that it is zero and the ring rows are stale)
                    const float4* r1 = &ring[(r - 1) & 3][tid];
                    const float4* r2 = &ring[(r - 2) & 3][tid];
#pragma unroll
                    for (int kq = 0; kq < 3; ++kq) {
                        pfma(wp[0 + kq], aprev, dyr[2 - kq]);
                        pfma(wp[3 + kq], aprev, f4p(r1[(1 - kq) * SD_CG]));
                        pfma(wp[6 + kq], aprev, f4p(r2[(1 - kq) * SD_CG]));
                    }
                }
                aprev = ar;
            }
            if (mrow) {                                         // workgroup-uniform
                *reinterpret_cast<float4*>(&dzs[r & 1][slot * SD_DZS + c]) = dz4;    // zeros from the halo / past-the-edge slots
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (st_ok[u]) img[r & 1][st_lds[u]] = iv[u];
            }
        }
        sd_barrier();                                           // the last row of the strip
        mma_row(h1 & 1, w0);
    }

    // ---- the block's partial rows ------------------------------------------------------------------------------------------------
    // depthwise weight gradient: fixed-order sum over the column slots (halo slots hold zeros)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        red[tid] = f4u(wp[t]);
        __syncthreads();
        if (slot == 0) {
            float4 a = f4zero();
            for (int q = 0; q < SD_SLOTS; ++q) add4(a, red[q * SD_CG + cgl]);
            float* dst = p.dw_parts + (int64_t)blockIdx.x * SD_C * 9;
            dst[(c + 0) * 9 + t] = a.x; dst[(c + 1) * 9 + t] = a.y; dst[(c + 2) * 9 + t] = a.z; dst[(c + 3) * 9 + t] = a.w;
        }
    }
    // stem: [P1 (32 x 27) | P^T P (27 x 27) | s1 (32) | s2 (32) | colsum(P) (27)], the four waves' matrices folded in wave order
    float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(SD_C, SD_K);
    float* fold = reinterpret_cast<float*>(&ring[0][0]);        // 32 x 32 (P1) + 32 x 32 (P^T P) + 32 (colsum) floats
    __syncthreads();
    for (int e = tid; e < 2 * 32 * 32 + 32; e += 256) fold[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (w == wave) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fold[(16 * a + 4 * lg + r) * 32 + 16 * b + l16] += accP[a][b][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fold[1024 + (4 * lg + r) * 32 + l16] += accG[0][r];
                fold[1024 + (4 * lg + r) * 32 + 16 + l16] += accG[1][r];
                fold[1024 + (16 + 4 * lg + r) * 32 + 16 + l16] += accG[2][r];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float t = cs[b];
                t += __shfl_xor(t, 16);
                t += __shfl_xor(t, 32);
                if (lg == 0) fold[2048 + 16 * b + l16] += t;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < SD_C * SD_K; e += 256) dst[e] = fold[(e / SD_K) * 32 + e % SD_K];
    for (int e = tid; e < SD_K * SD_K; e += 256) {
        const int i = e / SD_K, j = e % SD_K;
        dst[SD_C * SD_K + e] = (i / 16 > j / 16) ? fold[1024 + j * 32 + i] : fold[1024 + i * 32 + j];     // the block below the diagonal mirrors (0,1)
    }
    if (tid < SD_K) dst[SD_C * SD_K + SD_K * SD_K + 2 * SD_C + tid] = fold[2048 + tid];
    red[tid * 2] = f4u(s1); red[tid * 2 + 1] = f4u(s2);
    __syncthreads();
    if (slot == 0) {
        float4 a = f4zero(), b = f4zero();
        for (int q = 0; q < SD_SLOTS; ++q) { add4(a, red[(q * SD_CG + cgl) * 2]); add4(b, red[(q * SD_CG + cgl) * 2 + 1]); }
        st4(dst + SD_C * SD_K + SD_K * SD_K + c, a);
        st4(dst + SD_C * SD_K + SD_K * SD_K + SD_C + c, b);
    }
}

static bool sd_shape_ok(int N, int H, int W, int Cout) { return N > 0 && H >= 8 && W >= 8 && Cout == SD_C; }

static void sd_geom(SdGeom& g, int& gx, int N, int H, int W) {
    g.N = N; g.H = H; g.W = W;
    g.Ho = (H + 2 - 3) / 2 + 1; g.Wo = (W + 2 - 3) / 2 + 1;
    const int ns = (int)cdiv(g.Ho, 32);
    g.TH = (int)cdiv(g.Ho, ns);
    g.nHS = (int)cdiv(g.Ho, g.TH);
    g.nWT = (int)cdiv(g.Wo, SD_OWN);
    g.ntiles = (int64_t)N * g.nHS * g.nWT;
    int64_t want = g.ntiles;
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < 512 ? want : 512);                    // two resident workgroups per CU (<= 256 VGPRs)
}

}  // namespace mny

using namespace mny;

extern "C" int mny_stemdw_supported(int N, int H, int W, int Cout, int s_act, int d_act) {
    static const bool off = getenv("MNY_NO_STEMDW") != nullptr;
    const bool acts = s_act != MNY_ACT_HSWISH && s_act != MNY_ACT_HSIGMOID && d_act != MNY_ACT_HSWISH && d_act != MNY_ACT_HSIGMOID;
    return (!off && acts && sd_shape_ok(N, H, W, Cout)) ? 1 : 0;
}

extern "C" int mny_stemdw_bwd_parts(int N, int H, int W, int Cout) {
    if (!sd_shape_ok(N, H, W, Cout)) return MNY_EINVAL;
    SdGeom g; int gx;
    sd_geom(g, gx, N, H, W);
    return gx;
}

extern "C" size_t mny_stemdw_bwd_ws_floats(int N, int H, int W, int Cout) {
    if (!sd_shape_ok(N, H, W, Cout)) return 0;
    SdGeom g; int gx;
    sd_geom(g, gx, N, H, W);
    return (size_t)(gx + 1) * bnw_stride(SD_C, SD_K) + (size_t)SD_C * SD_K + (size_t)SD_K * SD_K + 64;      // partial rows + reduced row + B1 + Q + bias
}

extern "C" int mny_stemdw_bwd(const float* gd, const float* d, const float* d_scale, const float* d_shift, int d_act, const float* d_coef,
                              const float* s, const float* s_scale, const float* s_shift, const float* s_mean, const float* s_invstd,
                              const float* s_gamma, int s_act, const float* x_nchw, const float* w_stem, const float* w_dw,
                              float* dw_stem, float* dgamma_s, float* dbeta_s, float* dw_dw, float* dw_ws, float* ws,
                              int N, int H, int W, int Cout, void* stream) {
    MNY_REQUIRE(gd && d && d_scale && d_shift && d_coef && s && s_scale && s_shift && s_mean && s_invstd && s_gamma && x_nchw && w_stem && w_dw &&
                dw_stem && dgamma_s && dbeta_s && dw_ws && ws, "stemdw_bwd: null pointer");
    MNY_REQUIRE(sd_shape_ok(N, H, W, Cout) && s_act != MNY_ACT_HSWISH && s_act != MNY_ACT_HSIGMOID && d_act != MNY_ACT_HSWISH && d_act != MNY_ACT_HSIGMOID,
                "stemdw_bwd: N=%d H=%d W=%d Cout=%d acts %d/%d not supported", N, H, W, Cout, s_act, d_act);
    SdArgs a{gd, d, d_scale, d_shift, d_act, d_coef, s, s_scale, s_shift, s_mean, s_invstd, s_act, x_nchw, w_dw, dw_ws, ws, {}};
    int gx;
    sd_geom(a.g, gx, N, H, W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stemdw_bwd_kernel, dim3(gx), dim3(256), 0, st, a);
    int rc = check_launch("stemdw_bwd_kernel");
    if (rc) return rc;
    const int64_t stride = bnw_stride(SD_C, SD_K);
    float* red = ws + (size_t)gx * stride;
    float* B1 = red + stride;
    float* Q = B1 + (size_t)SD_C * SD_K;
    float* bias = Q + (size_t)SD_K * SD_K;
    rc = pw_bnbwd_finalize_launch(ws, gx, red, w_stem, s_gamma, s_mean, s_invstd, (int64_t)N * a.g.Ho * a.g.Wo, SD_C, SD_K, dw_stem, dgamma_s, dbeta_s, B1, Q, bias, st);
    if (rc || !dw_dw) return rc;                            // dw_dw == NULL: partial rows only (combined later by mny_reduce_batch)
    return launch_reduce_parts(dw_ws, gx, SD_C * 9, dw_dw, st);
}
