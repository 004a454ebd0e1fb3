// The front half of an inverted-residual block as ONE unit: expand 1x1 conv (K = 16 / 24 / 32 input channels -> C = 6K) + BN + ReLU6
// + depthwise 3x3 STRIDE-2 conv + BN, with the 6x-wide expand output NEVER written to HBM (mobilenetv2.py:73-85, the down-sampling
// blocks 16->96 @176^2, 24->144 @88^2, 32->192 @44^2 of the 352x352 network).
//
// Why: the expand output is the largest tensor of the network (3.05 GB at bs 256 for 16->96 @176^2).  Materialised, it is written
// once and read three times (depthwise forward, depthwise backward, the expand unit's own BN backward), and its gradient is written
// once and read twice: 21 GB of HBM traffic for ONE unit, 5.0 ms of a 40.6 ms step (round-3 profile).  But the reduction that
// produces it is 16 channels deep, and behind a stride-2 depthwise conv everything that really has to cross HBM is 4x smaller (the
// depthwise output Z and its gradient) or 6x thinner (the block input X and its gradient).  So every pass recomputes
// a = relu6(sc * (X W^T) + sh) where it needs it — on the matrix cores: v_mfma_f32_16x16x4_f32 is exact fp32 and the SAME fmaf chain
// (k ascending) as pw_thin_kernel's (pwthin.hip), so the recomputed values are bit-identical to what mny_pw_fwd would have stored
// (tests/test_gpu_exdw.py compares Z bit for bit with the materialised kernels):
//
//   forward   exdw_stats2    : column sums / sums of squares of Y = X W^T (BN batch statistics), no store            (reads X)
//             exdw_fwd2_s2   : a in an LDS ring of input rows -> Z = dw3x3_s2(a), Z statistics                        (reads X, writes Z)
//   backward  exdw_bwd1v2_s2 : Y again; dZ rebuilt from (G_z, Z) -> dW_dw, G_a = dw^T(dZ), dz = G_a * relu6'(z) -> BN sums of the expand
//                              unit, P1 = dz^T X, Gram = X^T X, colsum(X), and dz (ca o W)^T -> dX (ca = gamma * invstd is known
//                              BEFORE the BN-backward sums)                                                           (reads X, G_z, Z; writes dX)
//             finalize       : the expand unit's dgamma / dbeta / dW, Q = W^T diag(cb) W, bias = cc . W               (pwgemm.hip, fp64)
//             exdw_dxfix     : dX += X Q^T + bias (+ addend); optionally the BN-backward sums of the unit in front     (thin tensors only)
//
// 512-thread workgroups split into PRODUCER waves (matrix cores) and CONSUMER waves (stencil threads: 4 channels x one column, lanes =
// channel quads first, so every global store is a run of 16-B vectors) that run the same barrier sequence in separate loops (two
// disjoint register live sets).  Measured (LAB_NOTES.md R4.1): fp32 MFMA time and vector-ALU time ADD on this part, so these kernels
// are bound by total instruction issue, not by a pipe or by HBM; a first generation on the vector ALU alone (X rows broadcast from LDS,
// 4 x K weights per thread in registers) ran LDS- and ALU-bound together at half the rate and is gone.
#include "common.h"

namespace mny {

constexpr int kExTH = 6;            // forward: output rows (= input row pairs) per work item
constexpr int kExTHB = 4;           // backward: input row PAIRS per work item (LDS: three y / dz row-pair buffers next to the X tile)

struct ExGeom {
    int N, H, W, K, C, Ho, Wo;
    int nq, ppb;                    // channel quads (C/4), output columns per item (256 / nq)
    int nHS, nCT;                   // row strips / column tiles per image
    int items;
};

static bool ex_shape_ok(int N, int H, int W, int K, int C, int stride) {
    return N > 0 && stride == 2 && (K == 16 || K == 24 || K == 32) && C == 6 * K && H >= 4 && W >= 4 && (H & 1) == 0 && (W & 1) == 0;
}

static ExGeom ex_geom(int N, int H, int W, int K, int C, int th = kExTH) {
    ExGeom g;
    g.N = N; g.H = H; g.W = W; g.K = K; g.C = C; g.Ho = H / 2; g.Wo = W / 2;
    g.nq = C / 4; g.ppb = 256 / g.nq;
    g.nHS = (int)cdiv(g.Ho, th); g.nCT = (int)cdiv(g.Wo, g.ppb);
    g.items = N * g.nHS * g.nCT;
    return g;
}

static int ex_grid(const ExGeom& g, int per_cu) {
    static const int env = getenv("MNY_EXDW_GRID") ? atoi(getenv("MNY_EXDW_GRID")) : 0;
    int cap = env > 0 ? env : 256 * per_cu;
    int gx = g.items < cap ? g.items : cap;
    if (gx > 8) gx &= ~7;           // a multiple of 8: the XCD-contiguous item order below needs it
    return gx;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + barrier: it waits for EVERY outstanding memory operation
// of the wave (s_waitcnt vmcnt(0)), i.e. for the round trip of the streaming stores and of the prefetched loads in flight — with one
// barrier per output row that was ~3 000 cycles per row (forward, 16->96 @176^2: 0.74 ms against a 0.25 ms matrix-pipe bound).  Nothing
// that crosses waves in these kernels goes through global memory, so the barrier only needs the LDS counter.
__device__ __forceinline__ void ex_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the producer's view of X applied once per staged element: XF = 0 as is, 1 = scale / shift + min(max(z, slope z), hi) (pw_thin's XF = 1)
template <int XF>
__device__ __forceinline__ float4 ex_xf(float4 v, float4 sc, float4 sh, float slope, float hi) {
    if (XF == 1) {
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        v.x = fminf(fmaxf(v.x, slope * v.x), hi); v.y = fminf(fmaxf(v.y, slope * v.y), hi);
        v.z = fminf(fmaxf(v.z, slope * v.z), hi); v.w = fminf(fmaxf(v.w, slope * v.w), hi);
    }
    return v;
}

// XCD-contiguous logical block index: hardware places workgroup b on XCD b % 8, so blocks b, b + 8, ... get consecutive items
__device__ __forceinline__ int ex_lb() {
    const int gx = gridDim.x;
    return (gx & 7) == 0 ? (int)(blockIdx.x & 7) * (gx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
}

// ---------------------------------------------------------------------------------------------------------------------
// statistics of Y = view(X) W^T without writing Y: partial rows [gridDim.x][2][C] (sum, sum of squares), fed to mny_bn_finalize
// ---------------------------------------------------------------------------------------------------------------------
struct ExStatArgs {
    const float* x; const float* in_scale; const float* in_shift; int in_act;
    const float* w; float* parts; int64_t M; int C; int nq; int ppb; int64_t ntiles;
};
constexpr int kExStatRows = 128;          // rows per tile (LDS: 2 x 128 x (K + 4) floats = 36 KB at K = 32)

// ---------------------------------------------------------------------------------------------------------------------
// forward: Z = dw3x3_s2(relu6(e_scale * (view(X) W^T) + e_shift)), Z statistics
// ---------------------------------------------------------------------------------------------------------------------
struct ExFwdArgs {
    const float* x; const float* in_scale; const float* in_shift; int in_act;
    const float* w; const float* e_scale; const float* e_shift; const float* w_dw;
    float* z; float* parts; ExGeom g;
};

// On the matrix cores a 16-pixel x 16-channel tile of Y costs K/4 instructions whose A operand (one b32 LDS read per lane) is shared by
// all channel tiles and whose B operand (the weights of one channel tile: K/4 registers) never leaves the register file.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- statistics ------------------------------------------------------------------------------------------------------
template <int K, int XF>
__global__ __launch_bounds__(256) void exdw_stats2_kernel(ExStatArgs p) {
    constexpr int KQ = K / 4, KP = K + 4, ST = 256 / KQ * KQ, PS = ST / KQ, C = 6 * K, NT = C / 16;
    constexpr int NS = (kExStatRows * KQ + ST - 1) / ST;
    __shared__ __attribute__((aligned(16))) float xs[2][kExStatRows * KP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    float wb[NT][KQ];                                    // B operands: W[16 t + l16][4 s + lg]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int sI = 0; sI < KQ; ++sI) wb[t][sI] = p.w[(int64_t)(16 * t + l16) * K + 4 * sI + lg];
    const int kq_s = tid % KQ, row_s = tid / KQ;
    const bool stager = tid < ST;
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    float4 stg[NS];
    auto fetch = [&](int64_t tile) {
        const int64_t base = tile * kExStatRows;
        const int64_t left = p.M - 1 - base;
        const int lim = (int)(left < kExStatRows - 1 ? left : kExStatRows - 1);
        const float* ta = p.x + base * K;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int tr = row_s + i * PS;
            tr = tr < lim ? tr : lim;
            stg[i] = ld4(at_bytes(ta, (unsigned)(tr * K + 4 * kq_s) * 4u));
        }
    };
    auto park = [&](int buf, int64_t tile) {              // rows past M are parked as zeros: they add nothing to either sum
        const int64_t left = p.M - tile * kExStatRows;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int tr = row_s + i * PS;
            if (stager && tr < kExStatRows) *reinterpret_cast<float4*>(&xs[buf][tr * KP + 4 * kq_s]) = tr < left ? ex_xf<XF>(stg[i], xsc, xsh, slope, hi) : f4zero();
        }
    };
    float s1[NT], s2[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { s1[t] = 0.f; s2[t] = 0.f; }
    int64_t tile = blockIdx.x;
    int buf = 0;
    if (tile < p.ntiles) { fetch(tile); park(0, tile); }
    ex_barrier();
    for (; tile < p.ntiles; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < p.ntiles;
        if (has_next) fetch(tile + gridDim.x);
#pragma unroll
        for (int gi = 0; gi < kExStatRows / 64; ++gi) {   // this wave's 16-row groups of the tile
            const float* xp = &xs[buf][(16 * (wave + 4 * gi) + l16) * KP + lg];
            float a[KQ];
#pragma unroll
            for (int sI = 0; sI < KQ; ++sI) a[sI] = xp[4 * sI];
            f32x4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sI = 0; sI < KQ; ++sI)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sI], wb[t][sI], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                s1[t] += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
                s2[t] = fmaf(acc[t][0], acc[t][0], s2[t]); s2[t] = fmaf(acc[t][1], acc[t][1], s2[t]);
                s2[t] = fmaf(acc[t][2], acc[t][2], s2[t]); s2[t] = fmaf(acc[t][3], acc[t][3], s2[t]);
            }
        }
        if (has_next) park(buf ^ 1, tile + gridDim.x);
        ex_barrier();
        buf ^= 1;
    }
    // lane (l16, lg) of wave w holds the sums of channel 16 t + l16 over its pixels: 16 rows per channel -> one partial row per block
    float* red = &xs[0][0];                               // [16][2][C]
    ex_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        red[((wave * 4 + lg) * 2 + 0) * C + 16 * t + l16] = s1[t];
        red[((wave * 4 + lg) * 2 + 1) * C + 16 * t + l16] = s2[t];
    }
    ex_barrier();
    for (int e = tid; e < 2 * C; e += 256) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i * 2 * C + e];
        p.parts[(int64_t)blockIdx.x * 2 * C + e] = a;
    }
}

// ---- statistics, Gram form ----------------------------------------------------------------------------------------------
// sum_c = w_c . colsum(X) and sum_c^2 = w_c^T (X^T X) w_c: the statistics of Y need the K x K second-moment matrix of the (viewed) input
// and its column sums, not Y — M K^2 products instead of M K C, and the pass becomes a plain read of the thin X.  Lane (l16, lg) of a wave
// loads X[pixel 4u + lg][channel 16a + l16] as a dword (a wave instruction = 4 pixels x 64 contiguous bytes at K = 16): that register IS
// both operands of v_mfma_f32_16x16x4_f32 for the block (a, b) of X^T X (A[i][k] = X[k][i], B[k][j] = X[k][j], k = the pixel), so a
// 64-pixel chunk is 16 KT loads (all in flight before the first use), the view transform, 16 column-sum adds and 16 KT(KT+1)/2 MFMAs.
// A workgroup folds its four waves' matrices in fp64, forms its partial (sum, sum of squares) per output channel in fp64 and writes ONE
// partial row [2][C] like every other statistics producer — mny_bn_finalize does not know the difference.
template <int K, int XF>
__global__ __launch_bounds__(256) void exdw_gram_stats_kernel(ExStatArgs p) {
    constexpr int KT = (K + 15) / 16, KP = 16 * KT, C = 6 * K, NB = KT * (KT + 1) / 2, NE = KP * KP + KP;
    __shared__ float sG[4][NE];
    __shared__ double sD[NE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, lg = lane >> 4;
    float sc[KT], sh[KT];
    bool chok[KT];
    int chc[KT];
#pragma unroll
    for (int a = 0; a < KT; ++a) {
        const int ch = 16 * a + l16;
        chok[a] = ch < K;
        chc[a] = chok[a] ? ch : K - 1;
        sc[a] = (XF && p.in_scale) ? p.in_scale[chc[a]] : 1.f;
        sh[a] = (XF && p.in_scale) ? p.in_shift[chc[a]] : 0.f;
    }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    f32x4 acc[NB];
    float cs[KT];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < KT; ++a) cs[a] = 0.f;
    const int64_t nchunks = cdiv(p.M, 64);
    const int64_t per = cdiv(nchunks, (int64_t)gridDim.x);
    const int64_t c_begin = (int64_t)ex_lb() * per;
    const int64_t c_end = c_begin + per < nchunks ? c_begin + per : nchunks;
    for (int64_t c = c_begin + wave; c < c_end; c += 4) {
        float v[16][KT];
        const int64_t pix0 = c * 64 + lg;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t pix = pix0 + 4 * u;
            const float* row = p.x + (pix < p.M ? pix : p.M - 1) * K;
#pragma unroll
            for (int a = 0; a < KT; ++a) v[u][a] = row[chc[a]];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool ok = pix0 + 4 * u < p.M;
#pragma unroll
            for (int a = 0; a < KT; ++a) {
                float t = v[u][a];
                if (XF) { t = fmaf(t, sc[a], sh[a]); t = fminf(fmaxf(t, slope * t), hi); }
                t = (ok && chok[a]) ? t : 0.f;
                v[u][a] = t;
                cs[a] += t;
            }
            int blk = 0;
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int b = a; b < KT; ++b, ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u][a], v[u][b], acc[blk], 0, 0, 0);
        }
    }
    // accumulator of block (a, b): lane (l16, lg), element r = (X^T X)[16 a + 4 lg + r][16 b + l16]
    float* gw = sG[wave];
    {
        int blk = 0;
#pragma unroll
        for (int a = 0; a < KT; ++a)
#pragma unroll
            for (int b = a; b < KT; ++b, ++blk)
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[(16 * a + 4 * lg + r) * KP + 16 * b + l16] = acc[blk][r];
    }
#pragma unroll
    for (int a = 0; a < KT; ++a) {
        float t = cs[a];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (lg == 0) gw[KP * KP + 16 * a + l16] = t;
    }
    __syncthreads();
    for (int e = tid; e < NE; e += 256) {                  // the four waves in fixed order; blocks below the diagonal mirror the ones above
        int src = e;
        if (e < KP * KP) {
            const int i = e / KP, j = e % KP;
            if (i / 16 > j / 16) src = j * KP + i;
        }
        sD[e] = ((double)sG[0][src] + (double)sG[1][src]) + ((double)sG[2][src] + (double)sG[3][src]);
    }
    __syncthreads();
    for (int ch = tid; ch < C; ch += 256) {
        float wr[K];
#pragma unroll
        for (int k = 0; k < K; ++k) wr[k] = p.w[(int64_t)ch * K + k];
        double s = 0.0, q = 0.0;
#pragma unroll 4
        for (int i = 0; i < K; ++i) {
            double gi = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) gi += sD[i * KP + j] * (double)wr[j];
            q += (double)wr[i] * gi;
            s += (double)wr[i] * sD[KP * KP + i];
        }
        p.parts[((int64_t)blockIdx.x * 2 + 0) * C + ch] = (float)s;
        p.parts[((int64_t)blockIdx.x * 2 + 1) * C + ch] = (float)q;
    }
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// A workgroup owns TH = 6 output rows x PPB output columns of one image.  The X tile ((2 TH + 1) x (2 PPB + 1) pixels) is staged once;
// the activated expand output a = relu6(sc * Y + sh) lives in a ring of SIX input rows [slot][column][C], slot(row) = (row + 1) % 6:
// while the stencil threads (4 channels x one output column, dw3_fwd_kernel's tap order) read rows 2i-1..2i+1 of output row i, the
// matrix cores fill rows 2i+2, 2i+3 for the next one — one barrier per output row.  A row pair always sits in two ADJACENT slots, so a
// finished 16 x 16 tile goes to LDS with one address and four immediate offsets.  Matrix work is dealt to the four waves as contiguous
// runs of (channel tile, 16-pixel group) units: a wave touches at most three channel tiles (3 x K/4 weight registers).  Zero padding of
// the ACTIVATED tensor (input row / column -1) is applied by the stencil threads as tap masks, the matrix epilogue has no bounds checks.
template <int K> struct ExF {
    static constexpr int KQ = K / 4, KP = K + 4, C = 6 * K, CP = C + 4, NQ = C / 4, NT = C / 16, PPB = 256 / NQ, NCOLS = 2 * PPB + 1;
    static constexpr int TH = kExTH, ROWS = 2 * TH + 1, XPIX = ROWS * NCOLS;
    static constexpr int NG = (2 * NCOLS + 15) / 16;                      // 16-pixel groups of a row pair
    static constexpr int U = NT * NG, SL = 3;                             // units per row pair; channel-tile slots per wave
    static constexpr int XS_FLOATS = (XPIX + 16) * KP;                    // + the pixels a padded last group reads past the tile
    static constexpr int RING_FLOATS = 6 * NCOLS * CP;
    static constexpr size_t LDS = (size_t)(XS_FLOATS + RING_FLOATS) * 4;
    static constexpr int span(int w) { return (U * (w + 1) / 4 - 1) / NG - (U * w / 4) / NG + 1; }     // channel tiles wave w's run touches
    static_assert(span(0) <= SL && span(1) <= SL && span(2) <= SL && span(3) <= SL, "a wave's run of units must fit its channel-tile slots");
    static_assert(XS_FLOATS >= 256 * 8, "the end-of-kernel statistics fold lives in the X tile");
};

// 512 threads: waves 0-3 drive the matrix cores (producers of ring rows), waves 4-7 are the stencil threads (consumers) — the two
// kinds of work overlap inside a workgroup instead of waiting for a co-resident one to be in the other phase (the 256-thread form, every
// wave doing both in turn with two waves per SIMD, was latency-bound: 0.88 ms at 16->96 @176^2; this form 0.57).
template <int K, int XF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8))) void exdw_fwd2_s2_kernel(ExFwdArgs p) {
    using F = ExF<K>;
    constexpr int KQ = F::KQ, KP = F::KP, C = F::C, CP = F::CP, NQ = F::NQ, PPB = F::PPB, NCOLS = F::NCOLS, NG = F::NG, SL = F::SL, TH = F::TH;
    constexpr int ST = 512 / KQ * KQ, PS = ST / KQ, NS = (F::XPIX * KQ + ST - 1) / ST;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                         // [ROWS][NCOLS][KP]
    float* ring = lds + F::XS_FLOATS;                        // [6][NCOLS][CP]
    const ExGeom& g = p.g;
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < F::XS_FLOATS; i += 512) xs[i] = 0.f;           // pads and slack stay zero (finite operands for the padded groups)
    const int kq_s = tid % KQ, pix_s = tid / KQ;
    const bool stager = tid < ST;
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    float4 stg[NS];
    auto decode = [&](int item, int& n, int& i0, int& j0) {
        const int ct = item % g.nCT; const int t = item / g.nCT;
        const int hs = t % g.nHS; n = t / g.nHS;
        i0 = hs * TH; j0 = ct * PPB;
    };
    auto fetch = [&](int item) {
        int n, i0, j0;
        decode(item, n, i0, j0);
        const int r0 = 2 * i0 - 1, c0 = 2 * j0 - 1;
        const float* xn = p.x + (int64_t)n * g.H * g.W * K;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pix = min(pix_s + i * PS, F::XPIX - 1);
            const int rr = pix / NCOLS, cc = pix - rr * NCOLS;
            const int gr = min(max(r0 + rr, 0), g.H - 1), gc = min(max(c0 + cc, 0), g.W - 1);
            stg[i] = ld4(xn + ((int64_t)gr * g.W + gc) * K + 4 * kq_s);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pix = pix_s + i * PS;
            if (stager && pix < F::XPIX) *reinterpret_cast<float4*>(&xs[pix * KP + 4 * kq_s]) = ex_xf<XF>(stg[i], xsc, xsh, slope, hi);
        }
    };
    const int gx = gridDim.x;
    int item = ex_lb();
    if (item < g.items) fetch(item);

    // The two roles run the SAME barrier sequence (item top x 2, prologue, one per output row) in separate loops, so that the register
    // allocator sees two disjoint live sets (weights + accumulators | taps + window) instead of their union.
    if (wave < 4) {
        // ---- producers: wave W's run of units [UB, UE) of u = tile * NG + group, up to SL channel tiles; code specialised per wave so
        // that no predicate surrounds an MFMA (the first cut chose the units at run time: 49 M branches and 129 M scalar instructions per launch)
        auto run = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            constexpr int UB = F::U * W / 4, UE = F::U * (W + 1) / 4, T0 = UB / NG;
            const int lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
            float wb[SL][KQ], esc_s[SL], esh_s[SL];
#pragma unroll
            for (int j = 0; j < SL; ++j) {
                const int ch = min(16 * (T0 + j) + l16, C - 1);
#pragma unroll
                for (int sI = 0; sI < KQ; ++sI) wb[j][sI] = p.w[(int64_t)ch * K + 4 * sI + lg];
                esc_s[j] = p.e_scale[ch]; esh_s[j] = p.e_shift[ch];
            }
            auto g0 = [](int j) constexpr { const int t = T0 + j; return (UB > t * NG ? UB : t * NG) - t * NG; };
            auto g1 = [](int j) constexpr { const int t = T0 + j; return (UE < (t + 1) * NG ? UE : (t + 1) * NG) - t * NG; };
            // activated expand output of a row PAIR (FULL: 2 NCOLS pixels) or of one row (NCOLS pixels), row-major from X-tile row lr0
            // -> ring slots s0, s0 + 1 (adjacent).  All A fragments first, then every unit's K/4-deep chain one step at a time, then the
            // finished tiles are activated and parked.
            auto mfma_rows = [&](auto full_tag, int lr0, int s0) {
                constexpr bool FULL = decltype(full_tag)::value;
                constexpr int COUNT = FULL ? 2 * NCOLS : NCOLS;
                constexpr int NGC = (COUNT + 15) / 16;                   // groups that hold pixels of this pass
                const float* xp = xs + (lr0 * NCOLS + l16) * KP + lg;
                float a[NG][KQ];
#pragma unroll
                for (int gi = 0; gi < NGC; ++gi)
#pragma unroll
                    for (int sI = 0; sI < KQ; ++sI) a[gi][sI] = xp[16 * gi * KP + 4 * sI];
                f32x4 acc[SL][NG];
#pragma unroll
                for (int sI = 0; sI < KQ; ++sI)
                    static_for<0, SL>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        static_for<0, NGC>([&](auto gc) {
                            constexpr int gi = decltype(gc)::value;
                            if constexpr (gi >= g0(j) && gi < g1(j)) {
                                if (sI == 0) acc[j][gi] = f32x4{0.f, 0.f, 0.f, 0.f};
                                acc[j][gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi][sI], wb[j][sI], acc[j][gi], 0, 0, 0);
                            }
                        });
                    });
                static_for<0, SL>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    float* rp = ring + (s0 * NCOLS + 4 * lg) * CP + 16 * (T0 + j) + l16;
                    static_for<0, NGC>([&](auto gc) {
                        constexpr int gi = decltype(gc)::value;
                        if constexpr (gi >= g0(j) && gi < g1(j)) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const float av = __builtin_amdgcn_fmed3f(fmaf(acc[j][gi][v], esc_s[j], esh_s[j]), 0.f, 6.f);
                                // a pair's last group may run past its pixels (into a slot being read); a single row's spill-over lands on
                                // the next slot's first pixels, which hold the same values already
                                if (!FULL || 16 * gi + 15 < COUNT || 16 * gi + 4 * lg + v < COUNT) rp[(16 * gi + v) * CP] = av;
                            }
                        }
                    });
                });
            };
            for (; item < g.items; item += gx) {
                ex_barrier();                                // every wave is done with the previous tile (X and ring)
                park();
                ex_barrier();
                if (item + gx < g.items) fetch(item + gx);
                int n, i0, j0;
                decode(item, n, i0, j0);
                const int nrows = min(i0 + TH, g.Ho) - i0;
                mfma_rows(std::true_type{}, 1, 2);           // X-tile rows 1, 2 -> slots 2, 3; row 0 -> slot 1
                mfma_rows(std::false_type{}, 0, 1);
                ex_barrier();
                for (int li = 0; li < nrows; ++li) {
                    if (li + 1 < nrows) mfma_rows(std::true_type{}, 2 * li + 3, (2 * li + 4) % 6);
                    ex_barrier();
                }
            }
        };
        if (wave == 0) run(std::integral_constant<int, 0>{});
        else if (wave == 1) run(std::integral_constant<int, 1>{});
        else if (wave == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
        if (p.parts == nullptr) return;
        ex_barrier();
        ex_barrier();
        return;
    }
    // ---- consumers: stencil thread = 4 channels x one output column
    const int st = tid - 256;
    const int q = st % NQ, pp = st / NQ;
    const bool worker = pp < PPB;
    const int c = 4 * q;
    F4P wt[9];
    {
        float raw[36];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float4 v = ld4(p.w_dw + (int64_t)c * 9 + 4 * i);
            raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) { wt[t].lo = v2f{raw[t], raw[9 + t]}; wt[t].hi = v2f{raw[18 + t], raw[27 + t]}; }
    }
    F4P acc1 = f4p0(), acc2 = f4p0();
    for (; item < g.items; item += gx) {
        ex_barrier();
        park();
        ex_barrier();
        if (item + gx < g.items) fetch(item + gx);
        int n, i0, j0;
        decode(item, n, i0, j0);
        const int nrows = min(i0 + TH, g.Ho) - i0;
        ex_barrier();
        const int j = j0 + pp;
        float* zo = p.z + (((int64_t)n * g.Ho + i0) * g.Wo + min(j, g.Wo - 1)) * C + c;
        const int64_t opitch = (int64_t)g.Wo * C;
        const float ml = j > 0 ? 1.f : 0.f;                  // input column 2j-1 = -1: zero padding of the activated tensor
        for (int li = 0; li < nrows; ++li) {
            if (worker && j < g.Wo) {
                const float mt = (li > 0 || i0 > 0) ? 1.f : 0.f;         // input row 2i-1 = -1
                const float* r0p = ring + (((2 * li + 1) % 6) * NCOLS + 2 * pp) * CP + c;
                const float* r1p = ring + (((2 * li + 2) % 6) * NCOLS + 2 * pp) * CP + c;
                const float* r2p = ring + (((2 * li + 3) % 6) * NCOLS + 2 * pp) * CP + c;
                F4P t0[3], t1[3], t2[3];
#pragma unroll
                for (int qc = 0; qc < 3; ++qc) { t0[qc] = f4p(ld4(r0p + qc * CP)); t1[qc] = f4p(ld4(r1p + qc * CP)); t2[qc] = f4p(ld4(r2p + qc * CP)); }
                const v2f mtl = v2f{mt * ml, mt * ml}, mt2 = v2f{mt, mt}, ml2 = v2f{ml, ml};
                t0[0].lo *= mtl; t0[0].hi *= mtl; t0[1].lo *= mt2; t0[1].hi *= mt2; t0[2].lo *= mt2; t0[2].hi *= mt2;
                t1[0].lo *= ml2; t1[0].hi *= ml2; t2[0].lo *= ml2; t2[0].hi *= ml2;
                F4P o = f4p0();
#pragma unroll
                for (int qc = 0; qc < 3; ++qc) { pfma(o, t0[qc], wt[qc]); pfma(o, t1[qc], wt[3 + qc]); pfma(o, t2[qc], wt[6 + qc]); }
                st4_stream(zo, f4u(o));
                zo += opitch;
                acc1.lo += o.lo; acc1.hi += o.hi;
                pfma(acc2, o, o);
            }
            ex_barrier();
        }
    }
    if (p.parts == nullptr) return;
    ex_barrier();
    float4* red = reinterpret_cast<float4*>(lds);
    red[st * 2 + 0] = worker ? f4u(acc1) : f4zero();
    red[st * 2 + 1] = worker ? f4u(acc2) : f4zero();
    ex_barrier();
    if (pp == 0) {
        float4 a = f4zero(), b = f4zero();
        for (int i = 0; i < PPB; ++i) { add4(a, red[(i * NQ + q) * 2]); add4(b, red[(i * NQ + q) * 2 + 1]); }
        float* dst = p.parts + (int64_t)blockIdx.x * 2 * C;
        st4(dst + c, a);
        st4(dst + C + c, b);
    }
}

template <int K>
static int ex_fwd2_launch(const ExFwdArgs& a, bool xf, int grid, hipStream_t st) {
    const size_t lds = ExF<K>::LDS;
    if (!allow_lds((const void*)exdw_fwd2_s2_kernel<K, 0>, lds) || !allow_lds((const void*)exdw_fwd2_s2_kernel<K, 1>, lds)) {
        set_error("exdw_fwd: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    if (xf) hipLaunchKernelGGL((exdw_fwd2_s2_kernel<K, 1>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((exdw_fwd2_s2_kernel<K, 0>), dim3(grid), dim3(512), lds, st, a);
    return check_launch("exdw_fwd2_s2_kernel");
}

}  // namespace mny

using namespace mny;

extern "C" int mny_exdw_supported(int N, int H, int W, int K, int C, int stride) {
    static const bool off = getenv("MNY_NO_EXDW") != nullptr;      // A/B switch: the materialised path
    return (!off && ex_shape_ok(N, H, W, K, C, stride)) ? 1 : 0;
}

// MNY_EXDW_STATS=direct (read at every call: a test switches it): the first form, which recomputes Y on the matrix cores and sums it
static bool ex_stats_gram() { const char* e = getenv("MNY_EXDW_STATS"); return !(e && e[0] == 'd'); }
extern "C" int mny_exdw_stat_parts(int64_t M, int K, int C) {
    (void)K; (void)C;
    const int64_t tiles = cdiv(M, kExStatRows);
    return (int)(tiles < 768 ? tiles : 768);               // both forms run this many workgroups, one partial row each
}

extern "C" int mny_exdw_stats(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* stats,
                              int64_t M, int K, int C, void* stream) {
    MNY_REQUIRE(x && w && stats, "exdw_stats: null pointer");
    MNY_REQUIRE(M > 0 && (K == 16 || K == 24 || K == 32) && C == 6 * K, "exdw_stats: K=%d C=%d not supported", K, C);
    MNY_REQUIRE(!in_scale == !in_shift, "exdw_stats: scale and shift come together");
    MNY_REQUIRE(in_act <= MNY_ACT_RELU, "exdw_stats: unsupported input activation %d", in_act);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    ExStatArgs a{x, in_scale, in_shift, in_act, w, stats, M, C, C / 4, 256 / (C / 4), cdiv(M, kExStatRows)};
    const int grid = mny_exdw_stat_parts(M, K, C);
    hipStream_t st = (hipStream_t)stream;
#define MNY_EXS(K_) do { if (xf) hipLaunchKernelGGL((exdw_stats2_kernel<K_, 1>), dim3(grid), dim3(256), 0, st, a); \
                         else hipLaunchKernelGGL((exdw_stats2_kernel<K_, 0>), dim3(grid), dim3(256), 0, st, a); } while (0)
#define MNY_EXG(K_) do { if (xf) hipLaunchKernelGGL((exdw_gram_stats_kernel<K_, 1>), dim3(grid), dim3(256), 0, st, a); \
                         else hipLaunchKernelGGL((exdw_gram_stats_kernel<K_, 0>), dim3(grid), dim3(256), 0, st, a); } while (0)
    if (ex_stats_gram()) {
        if (K == 16) MNY_EXG(16); else if (K == 24) MNY_EXG(24); else MNY_EXG(32);
        return check_launch("exdw_gram_stats_kernel");
    }
    if (K == 16) MNY_EXS(16); else if (K == 24) MNY_EXS(24); else MNY_EXS(32);
#undef MNY_EXS
#undef MNY_EXG
    return check_launch("exdw_stats2_kernel");
}

extern "C" int mny_exdw_fwd_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_grid(ex_geom(N, H, W, K, C, kExTH), 2);
}

extern "C" int mny_exdw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                            const float* e_scale, const float* e_shift, const float* w_dw, float* z, float* z_stats,
                            int N, int H, int W, int K, int C, int stride, void* stream) {
    MNY_REQUIRE(x && w_exp && e_scale && e_shift && w_dw && z, "exdw_fwd: null pointer");
    MNY_REQUIRE(ex_shape_ok(N, H, W, K, C, stride), "exdw_fwd: N=%d H=%d W=%d K=%d C=%d stride=%d not supported", N, H, W, K, C, stride);
    MNY_REQUIRE(!in_scale == !in_shift, "exdw_fwd: scale and shift come together");
    MNY_REQUIRE(in_act <= MNY_ACT_RELU, "exdw_fwd: unsupported input activation %d", in_act);
    const ExGeom g = ex_geom(N, H, W, K, C, kExTH);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    ExFwdArgs a{x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, w_dw, z, z_stats, g};
    const int grid = ex_grid(g, 2);
    hipStream_t st = (hipStream_t)stream;
    return K == 16 ? ex_fwd2_launch<16>(a, xf, grid, st) : (K == 24 ? ex_fwd2_launch<24>(a, xf, grid, st) : ex_fwd2_launch<32>(a, xf, grid, st));
}


// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
namespace mny {

struct ExBwdArgs {
    const float* gz; const float* z; const float* z_scale; const float* z_shift; const float* z_coef;
    const float* x; const float* in_scale; const float* in_shift; int in_act;
    const float* w; const float* e_scale; const float* e_shift; const float* e_mean; const float* e_invstd; const float* w_dw;
    const float* e_gamma;
    float* partial; float* dw_parts;      // per-block rows: P1 | Gram | s1 | s2 | s3 (bnw_stride) and the depthwise weight gradient [C*9]
    float* dx;                            // receives dz (ca o W)^T; exdw_dxfix_kernel completes it
    ExGeom g;
};

// ---- backward pass ------------------------------------------------------------------------------------------------------------
// 512 threads.  Waves 0-3 (producers) own the matrix cores: per input row pair s they recompute the raw expand output Y of the pair
// (K/4-deep chains per 16-pixel x 16-channel tile) into T[s % 3], and — one pair behind — contract the pair's dz (left in T by the
// consumers) with X: P1 += dz^T X, Gram += X^T X.  Waves 4-7 (consumers, thread = 4 channels x one input-quad column j = input columns 2j, 2j+1,
// dw_bnbwd_s2k3_kernel's mapping) turn Y into a / relu6' / yhat, rebuild dZ, gather G_a, accumulate dW_dw and the BN sums, and overwrite the Y quad with
// dz IN PLACE.  Three row-pair buffers rotate (being produced | being consumed | being contracted): one barrier per row pair, and
// the matrix pipe, the vector ALU and the global-memory latency of (G_z, Z) overlap inside one workgroup.  Work is dealt to the
// producer waves statically (code specialised per wave: no predicates around the MFMAs, one basic block per row pair).
template <int K> struct ExB1 {
    static constexpr int KQ = K / 4, KP = K + 4, C = 6 * K, CP = C + 4, NQ = C / 4, PPB = 256 / NQ, NCOLS = 2 * PPB, NPX = 2 * NCOLS;
    static constexpr int NT = C / 16, KT = (K + 15) / 16, NG = (NPX + 15) / 16, U = NT * NG, SL = 3;
    static constexpr int XPIX = 2 * kExTHB * NCOLS;
    static constexpr int XS_FLOATS = (XPIX + 16) * KP;                    // + the pixels a padded last group reads past the tile
    static constexpr int T_FLOATS = NPX * CP;
    static constexpr int CST_F4 = 18 * NQ;
    static constexpr size_t LDS = (size_t)(XS_FLOATS + 3 * T_FLOATS) * 4 + (size_t)CST_F4 * 16;
    static constexpr int ub(int w) { return U * w / 4; }
    static constexpr int ue(int w) { return U * (w + 1) / 4; }
    static constexpr int span(int w) { return (ue(w) - 1) / NG - ub(w) / NG + 1; }
    static_assert(span(0) <= SL && span(1) <= SL && span(2) <= SL && span(3) <= SL, "a wave's run of units must fit its channel-tile slots");
    static_assert(ue(0) >= NG, "wave 0 must own every pixel group of channel tile 0 (it also accumulates Gram and colsum)");
    static_assert(3 * T_FLOATS >= C * K + K * K, "the end-of-kernel fold of P1 / Gram lives in the row-pair buffers");
    static_assert(XS_FLOATS >= 256 * 8 + 64 * KT, "the end-of-kernel folds of the per-thread sums live in the X tile");
};

template <int K, int XF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 8))) void exdw_bwd1v2_s2_kernel(ExBwdArgs p) {
    using B = ExB1<K>;
    constexpr int KQ = B::KQ, KP = B::KP, C = B::C, CP = B::CP, NQ = B::NQ, PPB = B::PPB, NCOLS = B::NCOLS, NPX = B::NPX, KT = B::KT,
                  NG = B::NG, SL = B::SL, NT = B::NT;
    constexpr int ST = 512 / KQ * KQ, PS = ST / KQ, NSB = (B::XPIX * KQ + ST - 1) / ST;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                                     // [2 THB][NCOLS][KP] (+ slack)
    float* T = lds + B::XS_FLOATS;                                       // [3][NPX][CP]
    float4* cst = reinterpret_cast<float4*>(T + 3 * B::T_FLOATS);        // [18][NQ]: 9 taps, z scale / shift, ca, cb, cc, e scale / shift / mean / invstd
    const ExGeom& g = p.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    for (int i = tid; i < B::XS_FLOATS + 3 * B::T_FLOATS; i += 512) lds[i] = 0.f;
    if (tid < NQ) {
        const int c = 4 * tid;
#pragma unroll
        for (int t = 0; t < 9; ++t) cst[t * NQ + tid] = make_float4(p.w_dw[(c + 0) * 9 + t], p.w_dw[(c + 1) * 9 + t], p.w_dw[(c + 2) * 9 + t], p.w_dw[(c + 3) * 9 + t]);
        cst[9 * NQ + tid] = ld4(p.z_scale + c);   cst[10 * NQ + tid] = ld4(p.z_shift + c);
        cst[11 * NQ + tid] = ld4(p.z_coef + c);   cst[12 * NQ + tid] = ld4(p.z_coef + C + c);   cst[13 * NQ + tid] = ld4(p.z_coef + 2 * C + c);
        cst[14 * NQ + tid] = ld4(p.e_scale + c);  cst[15 * NQ + tid] = ld4(p.e_shift + c);
        cst[16 * NQ + tid] = ld4(p.e_mean + c);   cst[17 * NQ + tid] = ld4(p.e_invstd + c);
    }
    const int kq_s = tid % KQ, pix_s = tid / KQ;
    const bool stager = tid < ST;
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    const int Ho = g.Ho, Wo = g.Wo;
    auto decode = [&](int item, int& n, int& i0, int& j0) {
        const int ct = item % g.nCT; const int tt = item / g.nCT;
        const int hs = tt % g.nHS; n = tt / g.nHS;
        i0 = hs * kExTHB; j0 = ct * PPB;
    };
    auto stage = [&](int n, int i0, int j0) {                            // X tile: rows 2 i0 .., columns 2 j0 ..; zeros outside the image
        const float* xn = p.x + (int64_t)n * g.H * g.W * K;
        float4 stg[NSB];
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            const int pix = pix_s + i * PS;
            const int rr = pix / NCOLS, cc = pix - rr * NCOLS;
            const int gr = min(2 * i0 + rr, g.H - 1), gc = min(2 * j0 + cc, g.W - 1);
            stg[i] = ld4(xn + ((int64_t)gr * g.W + gc) * K + 4 * kq_s);
        }
#pragma unroll
        for (int i = 0; i < NSB; ++i) {
            const int pix = pix_s + i * PS;
            const int rr = pix / NCOLS, cc = pix - rr * NCOLS;
            const bool valid = 2 * i0 + rr < g.H && 2 * j0 + cc < g.W;
            if (stager && pix < B::XPIX) *reinterpret_cast<float4*>(&xs[pix * KP + 4 * kq_s]) = valid ? ex_xf<XF>(stg[i], xsc, xsh, slope, hi) : f4zero();
        }
    };
    float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(C, K);
    float* fold = T;
    float4* red = reinterpret_cast<float4*>(xs);
    float* s3buf = xs + 256 * 8;

    if (wave < 4) {
        // ================================================= producers =================================================
        auto run = [&](auto wtag) {
            constexpr int W = decltype(wtag)::value;
            constexpr int UB = B::ub(W), UE = B::ue(W), T0 = UB / NG;
            float wb[SL][KQ];
            f32x4 accP[SL][KT], accG[KT][KT];
            float s3a[KT];
            // B operand of the data gradient's first term dz (ca o W): ca = gamma * invstd is known BEFORE the BN-backward sums, so that
            // term (the contraction over the C channels) is formed here, from the dz tile this pass holds anyway; what depends on the sums
            // (X Q^T + bias) is thin and follows in exdw_dxfix_kernel.  breg[kt][s][v] = ca[n] W[n][k], n = 16 s + 4 lg + v, k = 16 kt + l16
            float breg[KT][NT][4];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int sI = 0; sI < NT; ++sI)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int n = 16 * sI + 4 * lg + v, k = 16 * kt + l16;
                        breg[kt][sI][v] = k < K ? (float)((double)p.e_gamma[n] * (double)p.e_invstd[n] * (double)p.w[(int64_t)n * K + k]) : 0.f;
                    }
#pragma unroll
            for (int j = 0; j < SL; ++j) {
                const int ch = min(16 * (T0 + j) + l16, C - 1);
#pragma unroll
                for (int sI = 0; sI < KQ; ++sI) wb[j][sI] = p.w[(int64_t)ch * K + 4 * sI + lg];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) accP[j][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int a = 0; a < KT; ++a) {
                s3a[a] = 0.f;
#pragma unroll
                for (int b = 0; b < KT; ++b) accG[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            auto g0 = [](int j) constexpr { const int t = T0 + j; return (UB > t * NG ? UB : t * NG) - t * NG; };
            auto g1 = [](int j) constexpr { const int t = T0 + j; return (UE < (t + 1) * NG ? UE : (t + 1) * NG) - t * NG; };
            // raw expand output of row pair s -> T[s % 3]
            auto make_y = [&](int sp) {
                const float* xp = xs + (2 * sp * NCOLS + l16) * KP + lg;
                float* tp = T + (sp % 3) * B::T_FLOATS + (4 * lg) * CP + l16;
                float a[NG][KQ];
#pragma unroll
                for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                    for (int sI = 0; sI < KQ; ++sI) a[gi][sI] = xp[16 * gi * KP + 4 * sI];
                f32x4 acc[SL][NG];
                static_for<0, SL>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<0, NG>([&](auto gc) {
                        constexpr int gi = decltype(gc)::value;
                        if constexpr (gi >= g0(j) && gi < g1(j)) acc[j][gi] = f32x4{0.f, 0.f, 0.f, 0.f};
                    });
                });
#pragma unroll
                for (int sI = 0; sI < KQ; ++sI)
                    static_for<0, SL>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        static_for<0, NG>([&](auto gc) {
                            constexpr int gi = decltype(gc)::value;
                            if constexpr (gi >= g0(j) && gi < g1(j)) acc[j][gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi][sI], wb[j][sI], acc[j][gi], 0, 0, 0);
                        });
                    });
                static_for<0, SL>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    static_for<0, NG>([&](auto gc) {
                        constexpr int gi = decltype(gc)::value;
                        if constexpr (gi >= g0(j) && gi < g1(j)) {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (16 * gi + 12 + 3 < NPX || 16 * gi + 4 * lg + v < NPX) tp[(16 * gi + v) * CP + 16 * (T0 + j)] = acc[j][gi][v];
                        }
                    });
                });
            };
            // P1 += dz^T X over row pair s (dz in T[s % 3]); wave 0 also Gram += X^T X and the column sums of X
            auto contract = [&](int sp) {
                const float* dzb = T + (sp % 3) * B::T_FLOATS + lg * CP + l16;
                const float* xb = xs + (2 * sp * NCOLS + lg) * KP + l16;
                static_for<0, NG>([&](auto gc) {
                    constexpr int gi = decltype(gc)::value;
                    static_for<0, 4>([&](auto pc) {
                        constexpr int pg = decltype(pc)::value;
                        constexpr int px0 = 16 * gi + 4 * pg;
                        constexpr bool mine = (gi >= g0(0) && gi < g1(0)) || (SL > 1 && gi >= g0(1) && gi < g1(1)) || (SL > 2 && gi >= g0(2) && gi < g1(2));
                        if constexpr (px0 < NPX && mine) {
                            float b[KT];
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) b[kt] = xb[px0 * KP + 16 * kt];
                            static_for<0, SL>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                if constexpr (gi >= g0(j) && gi < g1(j)) {
                                    const float a = dzb[px0 * CP + 16 * (T0 + j)];
#pragma unroll
                                    for (int kt = 0; kt < KT; ++kt) accP[j][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[kt], accP[j][kt], 0, 0, 0);
                                }
                            });
                            if constexpr (W == 0) {
#pragma unroll
                                for (int ka = 0; ka < KT; ++ka) {
                                    s3a[ka] += b[ka];
#pragma unroll
                                    for (int kb = 0; kb < KT; ++kb) accG[ka][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ka], b[kb], accG[ka][kb], 0, 0, 0);
                                }
                            }
                        }
                    });
                });
            };
            // first term of the data gradient for row pair sp: 16-pixel tiles of dz (T[sp % 3]) against breg, reduction over the C channels;
            // the tiles of a pair go to the waves in rotation (tile counter modulo 4)
            int tile_ctr = 0;
            auto dx_tiles = [&](int sp, int n, int i, int j0) {
                const float* dzb = T + (sp % 3) * B::T_FLOATS;
#pragma unroll
                for (int tI = 0; tI < NG; ++tI) {
                    if (((tile_ctr + tI) & 3) != W) continue;
                    f32x4 accD[KT];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) accD[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float* dzp = dzb + (16 * tI + l16) * CP + 4 * lg;    // (a padded last tile reads the next buffer's rows: finite, discarded)
#pragma unroll
                    for (int sI = 0; sI < NT; ++sI) {
                        const float4 a4 = *reinterpret_cast<const float4*>(dzp + 16 * sI);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) accD[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v], breg[kt][sI][v], accD[kt], 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int px = 16 * tI + 4 * lg + v;
                        const int rr = px >= NCOLS ? 1 : 0, cc = px - rr * NCOLS;
                        const int gc = 2 * j0 + cc;
                        if (px < NPX && gc < g.W) {
                            float* dp = p.dx + (((int64_t)n * g.H + 2 * i + rr) * g.W + gc) * K;
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt)
                                if (16 * kt + l16 < K) dp[16 * kt + l16] = accD[kt][v];
                        }
                    }
                }
                tile_ctr += NG;
            };
            for (int item = ex_lb(); item < g.items; item += gridDim.x) {
                int n, i0, j0;
                decode(item, n, i0, j0);
                const int nq = min(i0 + kExTHB, Ho) - i0;
                ex_barrier();
                stage(n, i0, j0);
                ex_barrier();
                make_y(0);
                ex_barrier();
                for (int sp = 0; sp < nq; ++sp) {
                    if (sp + 1 < nq) make_y(sp + 1);
                    if (sp >= 1) { contract(sp - 1); dx_tiles(sp - 1, n, i0 + sp - 1, j0); }
                    ex_barrier();
                }
                contract(nq - 1);
                dx_tiles(nq - 1, n, i0 + nq - 1, j0);
            }
            // tail (same barrier sequence as the consumers'): fold the matrix accumulators in wave order
            ex_barrier();
            for (int i = tid; i < C * K + K * K; i += 256) fold[i] = 0.f;
            if constexpr (W == 0) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) s3buf[lg * (16 * KT) + 16 * kt + l16] = s3a[kt];
            }
            ex_barrier();
            for (int w = 0; w < 4; ++w) {
                if (w == W) {
                    static_for<0, SL>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (g0(j) < g1(j)) {
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const int nn = 16 * (T0 + j) + 4 * lg + v, k = 16 * kt + l16;
                                    if (k < K) fold[nn * K + k] += accP[j][kt][v];
                                }
                        }
                    });
                    if constexpr (W == 0) {
#pragma unroll
                        for (int ka = 0; ka < KT; ++ka)
#pragma unroll
                            for (int kb = 0; kb < KT; ++kb)
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const int r = 16 * ka + 4 * lg + v, k = 16 * kb + l16;
                                    if (r < K && k < K) fold[C * K + r * K + k] = accG[ka][kb][v];
                                }
                    }
                }
                ex_barrier();
            }
            for (int t = 0; t < 9; ++t) { ex_barrier(); ex_barrier(); }
        };
        if (wave == 0) run(std::integral_constant<int, 0>{});
        else if (wave == 1) run(std::integral_constant<int, 1>{});
        else if (wave == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
        return;
    }
    // ===================================================== consumers =====================================================
    const int st = tid - 256;
    const int q = st % NQ, pp = st / NQ;
    const bool worker = pp < PPB;
    const int ppa = worker ? pp : 0;
    const int c = 4 * q;
    F4P wp[9], s1 = f4p0(), s2 = f4p0();
#pragma unroll
    for (int t = 0; t < 9; ++t) wp[t] = f4p0();
    for (int item = ex_lb(); item < g.items; item += gridDim.x) {
        int n, i0, j0;
        decode(item, n, i0, j0);
        const int nq = min(i0 + kExTHB, Ho) - i0;
        ex_barrier();
        stage(n, i0, j0);
        ex_barrier();
        const int j = j0 + ppa;
        const bool colv = worker && j < Wo;
        const float am = colv ? 1.f : 0.f;
        const float jm1 = (j + 1 < Wo) ? am : 0.f;
        const int jc0 = min(j, Wo - 1), jc1 = min(j + 1, Wo - 1);
        int lo = q;
        asm volatile("" : "+v"(lo));
        const float4* my = cst + lo;
        // (G_z, Z) of output row ho at columns j, j + 1: per-item 64-bit bases, per-row 32-bit offsets (an image plane stays below 2^31 elements) — the
        // full 64-bit index arithmetic per row was 13 v_mad_u64_u32 per thread and row pair in a loop bound by instruction issue
        const float* gz0 = p.gz + ((int64_t)n * Ho * Wo + jc0) * C + c;
        const float* zz0 = p.z + ((int64_t)n * Ho * Wo + jc0) * C + c;
        const int dj = (jc1 - jc0) * C;
        auto dz_fetch = [&](int ho, float4 (&r)[4]) {
            const int ro = min(ho, Ho - 1) * Wo * C;
            r[0] = ld4(gz0 + ro); r[1] = ld4(zz0 + ro);
            r[2] = ld4(gz0 + ro + dj); r[3] = ld4(zz0 + ro + dj);
        };
        auto dz2 = [&](v2f gv, v2f zv, v2f s, v2f h, v2f a, v2f b, v2f cterm) {
            const v2f t = __builtin_elementwise_fma(zv, s, h);
            const v2f d = gv * v2f{(t.x > 0.f ? 1.f : 0.f) * (t.x < 6.f ? 1.f : 0.f), (t.y > 0.f ? 1.f : 0.f) * (t.y < 6.f ? 1.f : 0.f)};
            return __builtin_elementwise_fma(a, d, __builtin_elementwise_fma(b, zv, cterm));
        };
        auto dz_finish = [&](int ho, const float4 (&r)[4], F4P& d0, F4P& d1) {
            const float rm = (ho < Ho) ? 1.f : 0.f;
            const F4P zs = f4p(my[9 * NQ]), zh = f4p(my[10 * NQ]), ca = f4p(my[11 * NQ]), cb = f4p(my[12 * NQ]), cc = f4p(my[13 * NQ]);
            const F4P G0 = f4p(r[0]), Z0 = f4p(r[1]), G1 = f4p(r[2]), Z1 = f4p(r[3]);
            const v2f m0 = v2f{rm * am, rm * am}, m1 = v2f{rm * jm1, rm * jm1};
            d0.lo = dz2(G0.lo, Z0.lo, zs.lo, zh.lo, ca.lo, cb.lo, cc.lo) * m0; d0.hi = dz2(G0.hi, Z0.hi, zs.hi, zh.hi, ca.hi, cb.hi, cc.hi) * m0;
            d1.lo = dz2(G1.lo, Z1.lo, zs.lo, zh.lo, ca.lo, cb.lo, cc.lo) * m1; d1.hi = dz2(G1.hi, Z1.hi, zs.hi, zh.hi, ca.hi, cb.hi, cc.hi) * m1;
        };
        float4 rawn[4];
        F4P d00, d01;
        {
            float4 r0[4];
            dz_fetch(i0, r0);
            dz_fetch(i0 + 1, rawn);
            dz_finish(i0, r0, d00, d01);
        }
        ex_barrier();                                                    // row pair 0's Y is in T[0]
        for (int sp = 0; sp < nq; ++sp) {
            const int i = i0 + sp;
            float* tb = T + (sp % 3) * B::T_FLOATS + (2 * ppa) * CP + c;
            F4P d10, d11;
            dz_finish(i + 1, rawn, d10, d11);
            if (sp + 1 < nq) dz_fetch(i + 2, rawn);
            // one pixel of the quad at a time (register budget: 128 with four waves per SIMD): gather G_a with the statically known taps,
            // read the pixel's Y quad, mask, sums, the depthwise weight gradient's terms of this pixel, dz back in place
#define WG(t) f4p(my[(t) * NQ])
            static_for<0, 4>([&](auto pc) {
                constexpr int px = decltype(pc)::value;
                F4P o = f4p0();
                if constexpr (px == 0) { pfma(o, d00, WG(4)); }
                if constexpr (px == 1) { pfma(o, d00, WG(5)); pfma(o, d01, WG(3)); }
                if constexpr (px == 2) { pfma(o, d00, WG(7)); pfma(o, d10, WG(1)); }
                if constexpr (px == 3) { pfma(o, d00, WG(8)); pfma(o, d01, WG(6)); pfma(o, d10, WG(2)); pfma(o, d11, WG(0)); }
                float* yp = tb + ((px >> 1) * NCOLS + (px & 1)) * CP;
                const F4P y = f4p(ld4(yp));
                const F4P esc = f4p(my[14 * NQ]), esh = f4p(my[15 * NQ]);
                const v2f z0 = __builtin_elementwise_fma(y.lo, esc.lo, esh.lo), z1 = __builtin_elementwise_fma(y.hi, esc.hi, esh.hi);
                F4P a;
                a.lo = v2f{__builtin_amdgcn_fmed3f(z0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z0.y, 0.f, 6.f)};
                a.hi = v2f{__builtin_amdgcn_fmed3f(z1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z1.y, 0.f, 6.f)};
                const bool m0 = z0.x > 0.f && z0.x < 6.f, m1 = z0.y > 0.f && z0.y < 6.f, m2 = z1.x > 0.f && z1.x < 6.f, m3 = z1.y > 0.f && z1.y < 6.f;
                o.lo = v2f{m0 ? o.lo.x : 0.f, m1 ? o.lo.y : 0.f};
                o.hi = v2f{m2 ? o.hi.x : 0.f, m3 ? o.hi.y : 0.f};
                s1.lo += o.lo; s1.hi += o.hi;
                const F4P emu = f4p(my[16 * NQ]), eis = f4p(my[17 * NQ]);
                const v2f h0 = (y.lo - emu.lo) * eis.lo, h1 = (y.hi - emu.hi) * eis.hi;
                s2.lo = __builtin_elementwise_fma(o.lo, h0, s2.lo); s2.hi = __builtin_elementwise_fma(o.hi, h1, s2.hi);
                if constexpr (px == 0) { pfma(wp[4], a, d00); }
                if constexpr (px == 1) { pfma(wp[5], a, d00); pfma(wp[3], a, d01); }
                if constexpr (px == 2) { pfma(wp[7], a, d00); pfma(wp[1], a, d10); }
                if constexpr (px == 3) { pfma(wp[8], a, d00); pfma(wp[6], a, d01); pfma(wp[2], a, d10); pfma(wp[0], a, d11); }
                if (worker) *reinterpret_cast<float4*>(yp) = f4u(o);
            });
#undef WG
            ex_barrier();
            d00 = d10; d01 = d11;
        }
    }
    // tail: the producers fold their accumulators (6 barriers), then this half writes the block's partial rows
    ex_barrier();
    red[st * 2 + 0] = worker ? f4u(s1) : f4zero();
    red[st * 2 + 1] = worker ? f4u(s2) : f4zero();
    ex_barrier();
    for (int w = 0; w < 4; ++w) ex_barrier();
    for (int e = st; e < C * K + K * K; e += 256) dst[e] = fold[e];
    if (pp == 0) {
        float4 a = f4zero(), b = f4zero();
        for (int i = 0; i < PPB; ++i) { add4(a, red[(i * NQ + q) * 2]); add4(b, red[(i * NQ + q) * 2 + 1]); }
        st4(dst + C * K + K * K + c, a);
        st4(dst + C * K + K * K + C + c, b);
    }
    if (st < K) {
        float a = 0.f;
        for (int i = 0; i < 4; ++i) a += s3buf[i * (16 * KT) + st];
        dst[C * K + K * K + 2 * C + st] = a;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        ex_barrier();
        red[st] = worker ? f4u(wp[t]) : f4zero();
        ex_barrier();
        if (pp == 0) {
            float4 a = f4zero();
            for (int i = 0; i < PPB; ++i) add4(a, red[i * NQ + q]);
            float* dd = p.dw_parts + (int64_t)blockIdx.x * C * 9;
            dd[(c + 0) * 9 + t] = a.x; dd[(c + 1) * 9 + t] = a.y; dd[(c + 2) * 9 + t] = a.z; dd[(c + 3) * 9 + t] = a.w;
        }
    }
}

template <int K>
static int ex_bwd1v2_launch(const ExBwdArgs& a, bool xf, int grid, hipStream_t st) {
    const size_t lds = ExB1<K>::LDS;
    if (!allow_lds((const void*)exdw_bwd1v2_s2_kernel<K, 0>, lds) || !allow_lds((const void*)exdw_bwd1v2_s2_kernel<K, 1>, lds)) {
        set_error("exdw_bwd: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    if (xf) hipLaunchKernelGGL((exdw_bwd1v2_s2_kernel<K, 1>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((exdw_bwd1v2_s2_kernel<K, 0>), dim3(grid), dim3(512), lds, st, a);
    return check_launch("exdw_bwd1v2_s2_kernel");
}

// ---- the thin remainder of the data gradient: dx += view(X) Q^T + bias (+ addend) -------------------------------------------------
// Q[K][K] = W^T diag(cb) W and bias[K] = cc . W depend on the BN-backward sums (fp64 finalize, pwgemm.hip); X and dx are K channels wide.
// Thread = one pixel x 4 output channels; the pixel's K inputs arrive as 16-B loads shared by its K/4 threads (adjacent lanes).
// RED: x is the raw output of a conv+BN(+act) unit consumed only here, so the finished dx IS that unit's complete output gradient: its
// BN-backward sums (sum dz, sum dz * yhat; mny_bn_bwd_reduce's arithmetic) leave with it as partial rows [gridDim.x][2][K] and the
// unit's separate reduce pass — a re-read of dx and x — disappears (the project conv 32->16 @176^2 in front of the first expand unit).
struct ExFixArgs { float* dx; const float* x; const float* in_scale; const float* in_shift; int in_act; const float* Q; const float* bias;
                   const float* addend; int64_t M; const float* in_mean; const float* in_invstd; float* in_red; };
template <int K, int XF, bool RED>
__global__ __launch_bounds__(256) void exdw_dxfix_kernel(ExFixArgs p) {
    constexpr int KQ = K / 4;
    __shared__ float4 red[RED ? 256 * 2 : 1];
    const int kq = threadIdx.x % KQ;
    float4 s1 = f4zero(), s2 = f4zero(), rsc = f4one(), rsh = f4zero(), rmu = f4zero(), ris = f4zero();
    if (RED) {
        if (p.in_scale) { rsc = ld4(p.in_scale + 4 * kq); rsh = ld4(p.in_shift + 4 * kq); }
        rmu = ld4(p.in_mean + 4 * kq); ris = ld4(p.in_invstd + 4 * kq);
    }
    float4 qrow[K];                                   // Q[4 kq + e][k'] as qrow[k'].{x,y,z,w}
#pragma unroll
    for (int k2 = 0; k2 < K; ++k2)
        qrow[k2] = make_float4(p.Q[(4 * kq + 0) * K + k2], p.Q[(4 * kq + 1) * K + k2], p.Q[(4 * kq + 2) * K + k2], p.Q[(4 * kq + 3) * K + k2]);
    const float4 b4 = ld4(p.bias + 4 * kq);
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    // A lane loads ITS quarter of the pixel's input row (one coalesced 16-byte load) and the K/4 lanes of a pixel trade quarters through
    // LDS (double-buffered, one barrier per pixel group); every lane loading the whole row itself was K/4 loads of which the lanes of a
    // pixel fetched the same 16 bytes — 3.1 TB/s on a stream that is three tensor passes.
    __shared__ float4 xs[2][2][256];
    float4 xsc = f4one(), xsh = f4zero();
    if (XF) { xsc = ld4(p.in_scale + 4 * kq); xsh = ld4(p.in_shift + 4 * kq); }
    const int pix0 = (threadIdx.x / KQ) * KQ;
    constexpr int PPB = 256 / KQ;
    const int64_t stride = (int64_t)gridDim.x * (2 * PPB);
    int buf = 0;
    // two pixel groups per trip: both groups' loads are in the air before the barrier (a lane has 2 x 32 bytes in flight; with one group
    // the 146-register producer-sums form ran 3 waves per SIMD at 3.4 TB/s)
    for (int64_t base = (int64_t)blockIdx.x * (2 * PPB); base < p.M; base += stride, buf ^= 1) {      // workgroup-uniform trip count
        int64_t m[2]; bool live[2]; float4 yraw[2], acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            m[u] = base + u * PPB + threadIdx.x / KQ;
            live[u] = m[u] < p.M;
            yraw[u] = live[u] ? ld4(p.x + m[u] * K + 4 * kq) : f4zero();
            acc[u] = live[u] ? ld4(p.dx + m[u] * K + 4 * kq) : f4zero();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xs[buf][u][threadIdx.x] = XF ? ex_xf<1>(yraw[u], xsc, xsh, slope, hi) : yraw[u];
            acc[u].x += b4.x; acc[u].y += b4.y; acc[u].z += b4.z; acc[u].w += b4.w;
            if (p.addend && live[u]) add4(acc[u], ld4(p.addend + m[u] * K + 4 * kq));
        }
        __syncthreads();                                  // the other buffer is rewritten after the NEXT barrier only
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float4 xv = xs[buf][u][pix0 + j];
                fma4(acc[u], make_float4(xv.x, xv.x, xv.x, xv.x), qrow[4 * j + 0]); fma4(acc[u], make_float4(xv.y, xv.y, xv.y, xv.y), qrow[4 * j + 1]);
                fma4(acc[u], make_float4(xv.z, xv.z, xv.z, xv.z), qrow[4 * j + 2]); fma4(acc[u], make_float4(xv.w, xv.w, xv.w, xv.w), qrow[4 * j + 3]);
            }
            if (!live[u]) continue;
            st4_stream(p.dx + m[u] * K + 4 * kq, acc[u]);
            if (RED) {
                const float4 yr = yraw[u], ac = acc[u];
                float4 dz;
                dz.x = ac.x * act_bwd(fmaf(yr.x, rsc.x, rsh.x), p.in_act); dz.y = ac.y * act_bwd(fmaf(yr.y, rsc.y, rsh.y), p.in_act);
                dz.z = ac.z * act_bwd(fmaf(yr.z, rsc.z, rsh.z), p.in_act); dz.w = ac.w * act_bwd(fmaf(yr.w, rsc.w, rsh.w), p.in_act);
                add4(s1, dz);
                s2.x = fmaf(dz.x, (yr.x - rmu.x) * ris.x, s2.x); s2.y = fmaf(dz.y, (yr.y - rmu.y) * ris.y, s2.y);
                s2.z = fmaf(dz.z, (yr.z - rmu.z) * ris.z, s2.z); s2.w = fmaf(dz.w, (yr.w - rmu.w) * ris.w, s2.w);
            }
        }
    }
    if (RED) {
        red[threadIdx.x * 2] = s1;
        red[threadIdx.x * 2 + 1] = s2;
        __syncthreads();
        if (threadIdx.x < KQ) {
            float4 a = f4zero(), b = f4zero();
            for (int r = 0; r < (int)blockDim.x / KQ; ++r) { add4(a, red[(r * KQ + kq) * 2]); add4(b, red[(r * KQ + kq) * 2 + 1]); }
            st4(p.in_red + (int64_t)blockIdx.x * 2 * K + 4 * kq, a);
            st4(p.in_red + (int64_t)blockIdx.x * 2 * K + K + 4 * kq, b);
        }
    }
}

static int ex_dxfix_grid(int64_t M, int K) {
    const int64_t want = cdiv(M, 2 * (256 / (K / 4)));      // two pixel groups per workgroup trip
    return (int)(want < 1024 ? want : 1024);          // (= partial rows of the RED form: mny_max_parts bounds them)
}
template <int K>
static int ex_dxfix_launch(const ExFixArgs& a, bool xf, hipStream_t st) {
    const int grid = ex_dxfix_grid(a.M, K);
    const dim3 block(256 / (K / 4) * (K / 4));
    if (a.in_red) {
        if (xf) hipLaunchKernelGGL((exdw_dxfix_kernel<K, 1, true>), dim3(grid), block, 0, st, a);
        else hipLaunchKernelGGL((exdw_dxfix_kernel<K, 0, true>), dim3(grid), block, 0, st, a);
    } else if (xf) hipLaunchKernelGGL((exdw_dxfix_kernel<K, 1, false>), dim3(grid), block, 0, st, a);
    else hipLaunchKernelGGL((exdw_dxfix_kernel<K, 0, false>), dim3(grid), block, 0, st, a);
    return check_launch("exdw_dxfix_kernel");
}

struct ExWs { size_t partials, red, B1, Q, bias, total; };
static ExWs ex_ws(const ExGeom& g, int grid) {
    ExWs w;
    const size_t stride = (size_t)bnw_stride(g.C, g.K);
    w.partials = 0;
    w.red = (size_t)grid * stride;
    w.B1 = w.red + stride;
    w.Q = w.B1 + (size_t)g.C * g.K;
    w.bias = w.Q + (size_t)g.K * g.K;
    w.total = w.bias + 64;
    return w;
}

}  // namespace mny

// workgroups (= partial rows) of the backward pass: ONE 512-thread workgroup per CU (its stencil half needs ~210 VGPRs: accumulators of
// the depthwise weight gradient, the BN sums, two dZ rows, the prefetched (G_z, Z) row; at 128 registers they spilled to scratch)
static int ex_bwd1_grid(const ExGeom& g) { return ex_grid(g, 1); }

extern "C" int mny_exdw_bwd_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_bwd1_grid(ex_geom(N, H, W, K, C, kExTHB));
}

extern "C" size_t mny_exdw_bwd_ws_floats(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return 0;
    const ExGeom g = ex_geom(N, H, W, K, C, kExTHB);
    return ex_ws(g, ex_bwd1_grid(g)).total;
}

static int exdw_bwd_impl(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                         const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                         const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd, const float* e_gamma,
                         const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                         float* dw_dw, float* dw_ws, float* ws, int N, int H, int W, int K, int C, int stride, void* stream,
                         const float* in_mean, const float* in_invstd, float* in_red) {
    MNY_REQUIRE(gz && z && z_scale && z_shift && z_coef && x && w_exp && e_scale && e_shift && e_mean && e_invstd && e_gamma && w_dw && dx && dw_exp &&
                    dgamma_e && dbeta_e && dw_ws && ws, "exdw_bwd: null pointer");
    MNY_REQUIRE(ex_shape_ok(N, H, W, K, C, stride), "exdw_bwd: N=%d H=%d W=%d K=%d C=%d stride=%d not supported", N, H, W, K, C, stride);
    MNY_REQUIRE(z_act == MNY_ACT_RELU6, "exdw_bwd: the depthwise unit's activation must be ReLU6 (got %d)", z_act);
    MNY_REQUIRE(!in_scale == !in_shift, "exdw_bwd: scale and shift come together");
    MNY_REQUIRE(in_act <= MNY_ACT_RELU, "exdw_bwd: unsupported input activation %d", in_act);
    // two passes: the first OVERWRITES dx with dz (ca o W)^T, only the remainder kernel reads the addend -> an addend aliased to dx would be lost
    MNY_REQUIRE(addend != dx, "exdw_bwd: addend must not alias dx (the first pass overwrites dx before the addend is read)");
    const ExGeom g = ex_geom(N, H, W, K, C, kExTHB);
    const int grid = ex_bwd1_grid(g);
    const ExWs o = ex_ws(g, grid);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    hipStream_t st = (hipStream_t)stream;
    ExBwdArgs a{gz, z, z_scale, z_shift, z_coef, x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, e_mean, e_invstd, w_dw, e_gamma,
                ws + o.partials, dw_ws, dx, g};
    static const int only = getenv("MNY_EXDW_ONLY") ? atoi(getenv("MNY_EXDW_ONLY")) : 0;       // timing aid: 1 = without the thin remainder kernel
    int rc = K == 16 ? ex_bwd1v2_launch<16>(a, xf, grid, st) : (K == 24 ? ex_bwd1v2_launch<24>(a, xf, grid, st) : ex_bwd1v2_launch<32>(a, xf, grid, st));
    if (rc) return rc;
    rc = pw_bnbwd_finalize_launch(ws + o.partials, grid, ws + o.red, w_exp, e_gamma, e_mean, e_invstd, (int64_t)N * H * W, C, K, dw_exp, dgamma_e, dbeta_e,
                                  ws + o.B1, ws + o.Q, ws + o.bias, st);
    if (rc) return rc;
    if (dw_dw) { rc = launch_reduce_parts(dw_ws, grid, C * 9, dw_dw, st); if (rc) return rc; }
    if (only == 1) return MNY_OK;
    // the pass left dz (ca o W)^T in dx; what depends on the BN-backward sums (Q, bias) is thin and follows
    ExFixArgs f{dx, x, in_scale, in_shift, in_act, ws + o.Q, ws + o.bias, addend, (int64_t)N * H * W, in_mean, in_invstd, in_red};
    return K == 16 ? ex_dxfix_launch<16>(f, xf, st) : (K == 24 ? ex_dxfix_launch<24>(f, xf, st) : ex_dxfix_launch<32>(f, xf, st));
}

extern "C" int mny_exdw_bwd(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                            const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                            const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd, const float* e_gamma,
                            const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                            float* dw_dw, float* dw_ws, float* ws, int N, int H, int W, int K, int C, int stride, void* stream) {
    return exdw_bwd_impl(gz, z, z_scale, z_shift, z_act, z_coef, x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, e_mean, e_invstd, e_gamma, w_dw,
                         addend, dx, dw_exp, dgamma_e, dbeta_e, dw_dw, dw_ws, ws, N, H, W, K, C, stride, stream, nullptr, nullptr, nullptr);
}

extern "C" int mny_exdw_bwd_red_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_dxfix_grid((int64_t)N * H * W, K);
}

extern "C" int mny_exdw_bwd_red(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                                const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                                const float* in_invstd, const float* w_exp, const float* e_scale, const float* e_shift, const float* e_mean,
                                const float* e_invstd, const float* e_gamma, const float* w_dw, const float* addend, float* dx, float* dw_exp,
                                float* dgamma_e, float* dbeta_e, float* dw_dw, float* dw_ws, float* ws, float* in_red, int N, int H, int W, int K, int C,
                                int stride, void* stream) {
    MNY_REQUIRE(in_mean && in_invstd && in_red, "exdw_bwd_red: null pointer");
    return exdw_bwd_impl(gz, z, z_scale, z_shift, z_act, z_coef, x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, e_mean, e_invstd, e_gamma, w_dw,
                         addend, dx, dw_exp, dgamma_e, dbeta_e, dw_dw, dw_ws, ws, N, H, W, K, C, stride, stream, in_mean, in_invstd, in_red);
}
