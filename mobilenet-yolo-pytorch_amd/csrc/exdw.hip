// The front half of an inverted-residual block as ONE unit: expand 1x1 conv (K = 16 / 24 / 32 input channels -> C = 6K) + BN + ReLU6
// + depthwise 3x3 STRIDE-2 conv + BN, with the 6x-wide expand output NEVER written to HBM (mobilenetv2.py:73-85, the three
// down-sampling blocks 16->96 @176^2, 24->144 @88^2, 32->192 @44^2 of the 352x352 network).
//
// Why: the expand output is the largest tensor of the network (3.05 GB at bs 256 for 16->96 @176^2).  Materialised, it is written
// once and read three times (depthwise forward, depthwise backward, the expand unit's own BN backward), and its gradient is written
// once and read twice: 21 GB of HBM traffic for ONE unit, 5.0 ms of a 40.6 ms step (round-3 profile).  But the reduction that
// produces it is 16 channels deep — recomputing an element is 8 packed FMAs — and behind a stride-2 depthwise conv everything that
// really has to cross HBM is 4x smaller (the depthwise output Z and its gradient) or 6x thinner (the block input X and its gradient).
// So every pass recomputes  a = relu6(sc * (X W^T) + sh)  where it needs it:
//
//   forward   exdw_stats   : column sums / sums of squares of Y = X W^T (BN batch statistics), no store      (reads X)
//             exdw_fwd     : a on the fly -> Z = dw3x3_s2(a), Z statistics                                   (reads X, writes Z)
//   backward  exdw_bwd<1>  : dZ rebuilt from (G_z, Z) -> dW_dw, G_a = dw^T(dZ), dz = G_a * relu6'(z) -> BN sums of the expand unit,
//                            P1 = dz^T X, Gram = X^T X, colsum(X) on the matrix cores                        (reads X, G_z, Z)
//             finalize     : the expand unit's dgamma / dbeta / dW and the three operands of its data gradient (pwgemm.hip, fp64)
//             exdw_bwd<2>  : dz again -> dX = dz B1^T + X Q^T + bias (+ addend) on the matrix cores          (reads X, G_z, Z; writes dX)
//
// The recomputed Y is the SAME fmaf chain, in the same order, as pw_thin_kernel's (pwthin.hip) — bit-identical values — and the stencil
// keeps dw3_fwd_kernel's tap order, so the unit computes what the materialised path computes (tests/test_gpu_exdw.py).
//
// Mapping ("thread = 4 channels x one output column"): a workgroup owns `ppb` = 256 / (C/4) adjacent output columns x 8 output rows
// of one image; the X tile it needs (17 x (2 ppb + 1) pixels x K floats, ~28 KB) is staged in LDS once, transformed (the producer's
// BN), and read back as wave broadcasts; the thread keeps its 4 x K expand weights and 9 x 4 depthwise taps in registers and slides
// down the rows: two new input rows (x 3 columns) per output row, the third carried over.  Lanes = channel quads first, so every
// global store is a run of 16-B vectors.
#include "common.h"

namespace mny {

constexpr int kExTH = 8;            // output rows (= input row pairs) per work item
constexpr int kExNS = 7;            // staging chunks per thread and item, upper bound (17 x 21 pixels x K/4 chunks over >= 252 threads)

struct ExGeom {
    int N, H, W, K, C, Ho, Wo;
    int nq, ppb;                    // channel quads (C/4), output columns per item (256 / nq)
    int nHS, nCT;                   // row strips / column tiles per image
    int items;
};

static bool ex_shape_ok(int N, int H, int W, int K, int C, int stride) {
    return N > 0 && stride == 2 && (K == 16 || K == 24 || K == 32) && C == 6 * K && H >= 4 && W >= 4 && (H & 1) == 0 && (W & 1) == 0;
}

static ExGeom ex_geom(int N, int H, int W, int K, int C) {
    ExGeom g;
    g.N = N; g.H = H; g.W = W; g.K = K; g.C = C; g.Ho = H / 2; g.Wo = W / 2;
    g.nq = C / 4; g.ppb = 256 / g.nq;
    g.nHS = (int)cdiv(g.Ho, kExTH); g.nCT = (int)cdiv(g.Wo, g.ppb);
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

// Y quad of one pixel: the fmaf chain of pw_thin_kernel::row_step (k ascending, accumulators start at zero)
template <int K>
__device__ __forceinline__ void ex_y(const float* __restrict__ xp, const v2f (&w01)[K], const v2f (&w23)[K], v2f& c01, v2f& c23) {
    c01 = v2f{0.f, 0.f}; c23 = v2f{0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < K / 4; ++kq) {
        const float4 a = *reinterpret_cast<const float4*>(xp + 4 * kq);
        c01 = __builtin_elementwise_fma(v2f{a.x, a.x}, w01[4 * kq + 0], c01); c23 = __builtin_elementwise_fma(v2f{a.x, a.x}, w23[4 * kq + 0], c23);
        c01 = __builtin_elementwise_fma(v2f{a.y, a.y}, w01[4 * kq + 1], c01); c23 = __builtin_elementwise_fma(v2f{a.y, a.y}, w23[4 * kq + 1], c23);
        c01 = __builtin_elementwise_fma(v2f{a.z, a.z}, w01[4 * kq + 2], c01); c23 = __builtin_elementwise_fma(v2f{a.z, a.z}, w23[4 * kq + 2], c23);
        c01 = __builtin_elementwise_fma(v2f{a.w, a.w}, w01[4 * kq + 3], c01); c23 = __builtin_elementwise_fma(v2f{a.w, a.w}, w23[4 * kq + 3], c23);
    }
}

template <int K>
__device__ __forceinline__ void ex_load_w(const float* __restrict__ W, int n0, v2f (&w01)[K], v2f (&w23)[K]) {
#pragma unroll
    for (int kq = 0; kq < K / 4; ++kq) {
        const float4 a = ld4(W + (int64_t)(n0 + 0) * K + 4 * kq), b = ld4(W + (int64_t)(n0 + 1) * K + 4 * kq);
        const float4 c = ld4(W + (int64_t)(n0 + 2) * K + 4 * kq), d = ld4(W + (int64_t)(n0 + 3) * K + 4 * kq);
        w01[4 * kq + 0] = v2f{a.x, b.x}; w01[4 * kq + 1] = v2f{a.y, b.y}; w01[4 * kq + 2] = v2f{a.z, b.z}; w01[4 * kq + 3] = v2f{a.w, b.w};
        w23[4 * kq + 0] = v2f{c.x, d.x}; w23[4 * kq + 1] = v2f{c.y, d.y}; w23[4 * kq + 2] = v2f{c.z, d.z}; w23[4 * kq + 3] = v2f{c.w, d.w};
    }
}

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

template <int K, int XF>
__global__ __launch_bounds__(256) void exdw_stats_kernel(ExStatArgs p) {
    constexpr int KQ = K / 4, KP = K + 4, ST = 256 / KQ * KQ, PS = ST / KQ;
    constexpr int NS = (kExStatRows * KQ + ST - 1) / ST;
    __shared__ __attribute__((aligned(16))) float xs[2][kExStatRows * KP];
    __shared__ float4 red[256 * 2];
    const int tid = threadIdx.x;
    const int q = tid % p.nq, pp = tid / p.nq;
    const bool worker = pp < p.ppb;
    v2f w01[K], w23[K];
    ex_load_w<K>(p.w, worker ? 4 * q : 0, w01, w23);
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
    auto park = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int tr = row_s + i * PS;
            if (stager && tr < kExStatRows) *reinterpret_cast<float4*>(&xs[buf][tr * KP + 4 * kq_s]) = ex_xf<XF>(stg[i], xsc, xsh, slope, hi);
        }
    };
    F4P s1 = f4p0(), s2 = f4p0();
    int64_t tile = blockIdx.x;
    int buf = 0;
    if (tile < p.ntiles) { fetch(tile); park(0); }
    __syncthreads();
    for (; tile < p.ntiles; tile += gridDim.x) {
        const bool has_next = tile + gridDim.x < p.ntiles;
        if (has_next) fetch(tile + gridDim.x);
        const int64_t left = p.M - tile * kExStatRows;
        const int rows = (int)(left < kExStatRows ? left : kExStatRows);
        if (worker) {
            for (int r = pp; r < rows; r += p.ppb) {
                v2f c01, c23;
                ex_y<K>(&xs[buf][r * KP], w01, w23, c01, c23);
                s1.lo += c01; s1.hi += c23;
                s2.lo = __builtin_elementwise_fma(c01, c01, s2.lo); s2.hi = __builtin_elementwise_fma(c23, c23, s2.hi);
            }
        }
        if (has_next) park(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    red[tid * 2 + 0] = worker ? f4u(s1) : f4zero();
    red[tid * 2 + 1] = worker ? f4u(s2) : f4zero();
    __syncthreads();
    if (pp == 0) {
        float4 a = f4zero(), b = f4zero();
        for (int i = 0; i < p.ppb; ++i) { add4(a, red[(i * p.nq + q) * 2]); add4(b, red[(i * p.nq + q) * 2 + 1]); }
        float* dst = p.parts + (int64_t)blockIdx.x * 2 * p.C;
        st4(dst + 4 * q, a);
        st4(dst + p.C + 4 * q, b);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: Z = dw3x3_s2(relu6(e_scale * (view(X) W^T) + e_shift)), Z statistics
// ---------------------------------------------------------------------------------------------------------------------
struct ExFwdArgs {
    const float* x; const float* in_scale; const float* in_shift; int in_act;
    const float* w; const float* e_scale; const float* e_shift; const float* w_dw;
    float* z; float* parts; ExGeom g;
};

template <int K, int XF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void exdw_fwd_s2_kernel(ExFwdArgs p) {
    constexpr int KQ = K / 4, KP = K + 4, ST = 256 / KQ * KQ, PS = ST / KQ, NS = kExNS;
    constexpr int ROWS = 2 * kExTH + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ExGeom& g = p.g;
    const int ncols = 2 * g.ppb + 1;
    const int npix = ROWS * ncols;
    float* xs = lds;                                         // [ROWS][ncols][KP]
    float4* red = reinterpret_cast<float4*>(lds);            // reused after the last item
    const int tid = threadIdx.x;
    const int q = tid % g.nq, pp = tid / g.nq;
    const bool worker = pp < g.ppb;
    const int c = 4 * q;

    v2f w01[K], w23[K];
    ex_load_w<K>(p.w, worker ? c : 0, w01, w23);
    F4P wt[9], esc, esh;
    {
        const int cc = worker ? c : 0;
        float raw[36];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const float4 v = ld4(p.w_dw + (int64_t)cc * 9 + 4 * i);
            raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) { wt[t].lo = v2f{raw[t], raw[9 + t]}; wt[t].hi = v2f{raw[18 + t], raw[27 + t]}; }
        esc = f4p(ld4(p.e_scale + cc)); esh = f4p(ld4(p.e_shift + cc));
    }
    const int kq_s = tid % KQ, pix_s = tid / KQ;
    const bool stager = tid < ST;
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);

    float4 stg[NS];
    auto decode = [&](int item, int& n, int& i0, int& j0) {
        const int ct = item % g.nCT; const int t = item / g.nCT;
        const int hs = t % g.nHS; n = t / g.nHS;
        i0 = hs * kExTH; j0 = ct * g.ppb;
    };
    auto fetch = [&](int item) {                             // global -> registers; out-of-image pixels re-read a clamped address
        int n, i0, j0;
        decode(item, n, i0, j0);
        const int r0 = 2 * i0 - 1, c0 = 2 * j0 - 1;
        const float* xn = p.x + (int64_t)n * g.H * g.W * K;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pix = pix_s + i * PS;
            if (pix < npix) {                                // uniform per i except in the last chunk
                const int rr = pix / ncols, cc = pix - rr * ncols;
                const int gr = min(max(r0 + rr, 0), g.H - 1), gc = min(max(c0 + cc, 0), g.W - 1);
                stg[i] = ld4(xn + ((int64_t)gr * g.W + gc) * K + 4 * kq_s);
            }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pix = pix_s + i * PS;
            if (stager && pix < npix) *reinterpret_cast<float4*>(&xs[pix * KP + 4 * kq_s]) = ex_xf<XF>(stg[i], xsc, xsh, slope, hi);
        }
    };

    F4P acc1 = f4p0(), acc2 = f4p0();
    const int gx = gridDim.x;
    int item = ex_lb();
    if (item < g.items) fetch(item);
    for (; item < g.items; item += gx) {
        __syncthreads();                                     // every wave is done with the previous tile
        park();
        __syncthreads();
        if (item + gx < g.items) fetch(item + gx);           // the next tile's loads fly under this tile's arithmetic
        int n, i0, j0;
        decode(item, n, i0, j0);
        const int j = j0 + pp;
        if (worker && j < g.Wo) {
            const int i1 = min(i0 + kExTH, g.Ho);
            const int r0 = 2 * i0 - 1;
            const float cm[3] = {(2 * j - 1 >= 0) ? 1.f : 0.f, 1.f, (2 * j + 1 < g.W) ? 1.f : 0.f};
            const float* xcol = xs + (2 * pp) * KP;          // LDS column 2 pp = input column 2 j - 1
            auto arow = [&](int lr, F4P (&r)[3]) {           // activated expand output at LDS row lr, the thread's three columns
                const int gr = r0 + lr;
                const float rm = (gr >= 0 && gr < g.H) ? 1.f : 0.f;
                const float* xr = xcol + lr * ncols * KP;
#pragma unroll
                for (int qc = 0; qc < 3; ++qc) {
                    v2f c01, c23;
                    ex_y<K>(xr + qc * KP, w01, w23, c01, c23);
                    const v2f z0 = __builtin_elementwise_fma(c01, esc.lo, esh.lo), z1 = __builtin_elementwise_fma(c23, esc.hi, esh.hi);
                    const float m = rm * cm[qc];
                    r[qc].lo = v2f{__builtin_amdgcn_fmed3f(z0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z0.y, 0.f, 6.f)} * v2f{m, m};
                    r[qc].hi = v2f{__builtin_amdgcn_fmed3f(z1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z1.y, 0.f, 6.f)} * v2f{m, m};
                }
            };
            float* zo = p.z + (((int64_t)n * g.Ho + i0) * g.Wo + j) * g.C + c;
            const int64_t opitch = (int64_t)g.Wo * g.C;
            auto emit = [&](const F4P (&top)[3], const F4P (&mid)[3], const F4P (&bot)[3]) {
                F4P o = f4p0();
#pragma unroll
                for (int qc = 0; qc < 3; ++qc) { pfma(o, top[qc], wt[qc]); pfma(o, mid[qc], wt[3 + qc]); pfma(o, bot[qc], wt[6 + qc]); }
                st4_stream(zo, f4u(o));
                zo += opitch;
                acc1.lo += o.lo; acc1.hi += o.hi;
                pfma(acc2, o, o);
            };
            F4P ra[3], rb[3], rc[3];
            arow(0, ra);
            int lr = 1;
            int left = i1 - i0;
#pragma clang loop unroll(disable)
            for (; left >= 3; left -= 3, lr += 6) {
                arow(lr, rb); arow(lr + 1, rc); emit(ra, rb, rc);
                arow(lr + 2, ra); arow(lr + 3, rb); emit(rc, ra, rb);
                arow(lr + 4, rc); arow(lr + 5, ra); emit(rb, rc, ra);
            }
            if (left >= 1) { arow(lr, rb); arow(lr + 1, rc); emit(ra, rb, rc); }
            if (left >= 2) { arow(lr + 2, ra); arow(lr + 3, rb); emit(rc, ra, rb); }
        }
    }
    if (p.parts == nullptr) return;
    __syncthreads();
    red[tid * 2 + 0] = worker ? f4u(acc1) : f4zero();
    red[tid * 2 + 1] = worker ? f4u(acc2) : f4zero();
    __syncthreads();
    if (pp == 0) {
        float4 a = f4zero(), b = f4zero();
        for (int i = 0; i < g.ppb; ++i) { add4(a, red[(i * g.nq + q) * 2]); add4(b, red[(i * g.nq + q) * 2 + 1]); }
        float* dst = p.parts + (int64_t)blockIdx.x * 2 * g.C;
        st4(dst + c, a);
        st4(dst + g.C + c, b);
    }
}

static size_t ex_fwd_lds(const ExGeom& g) {
    const size_t xs = (size_t)(2 * kExTH + 1) * (2 * g.ppb + 1) * (g.K + 4) * sizeof(float);
    return xs < 256 * 2 * sizeof(float4) ? 256 * 2 * sizeof(float4) : xs;
}

}  // namespace mny

using namespace mny;

extern "C" int mny_exdw_supported(int N, int H, int W, int K, int C, int stride) {
    static const bool off = getenv("MNY_NO_EXDW") != nullptr;      // A/B switch: the materialised path
    return (!off && ex_shape_ok(N, H, W, K, C, stride)) ? 1 : 0;
}

extern "C" int mny_exdw_stat_parts(int64_t M, int K, int C) {
    (void)K; (void)C;
    const int64_t tiles = cdiv(M, kExStatRows);
    return (int)(tiles < 768 ? tiles : 768);
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
#define MNY_EXS(K_) do { if (xf) hipLaunchKernelGGL((exdw_stats_kernel<K_, 1>), dim3(grid), dim3(256), 0, st, a); \
                         else hipLaunchKernelGGL((exdw_stats_kernel<K_, 0>), dim3(grid), dim3(256), 0, st, a); } while (0)
    if (K == 16) MNY_EXS(16); else if (K == 24) MNY_EXS(24); else MNY_EXS(32);
#undef MNY_EXS
    return check_launch("exdw_stats_kernel");
}

extern "C" int mny_exdw_fwd_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_grid(ex_geom(N, H, W, K, C), 2);
}

extern "C" int mny_exdw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                            const float* e_scale, const float* e_shift, const float* w_dw, float* z, float* z_stats,
                            int N, int H, int W, int K, int C, int stride, void* stream) {
    MNY_REQUIRE(x && w_exp && e_scale && e_shift && w_dw && z, "exdw_fwd: null pointer");
    MNY_REQUIRE(ex_shape_ok(N, H, W, K, C, stride), "exdw_fwd: N=%d H=%d W=%d K=%d C=%d stride=%d not supported", N, H, W, K, C, stride);
    MNY_REQUIRE(!in_scale == !in_shift, "exdw_fwd: scale and shift come together");
    MNY_REQUIRE(in_act <= MNY_ACT_RELU, "exdw_fwd: unsupported input activation %d", in_act);
    const ExGeom g = ex_geom(N, H, W, K, C);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    ExFwdArgs a{x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, w_dw, z, z_stats, g};
    const int grid = ex_grid(g, 2);
    const size_t lds = ex_fwd_lds(g);
    hipStream_t st = (hipStream_t)stream;
#define MNY_EXF(K_) do { if (xf) hipLaunchKernelGGL((exdw_fwd_s2_kernel<K_, 1>), dim3(grid), dim3(256), lds, st, a); \
                         else hipLaunchKernelGGL((exdw_fwd_s2_kernel<K_, 0>), dim3(grid), dim3(256), lds, st, a); } while (0)
    if (K == 16) MNY_EXF(16); else if (K == 24) MNY_EXF(24); else MNY_EXF(32);
#undef MNY_EXF
    return check_launch("exdw_fwd_s2_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------
// backward.  Thread = 4 channels x one INPUT-QUAD column j (input columns 2j, 2j+1 = output column j), walking down quad rows i
// (dw_bnbwd_s2k3_kernel's mapping, dwbwd.hip): dZ[i..i+1][j..j+1] rebuilt from (G_z, Z), the quad's four expand outputs
// recomputed from the staged X tile (MODE 1) or their ReLU6 masks read back as 16 bits (MODE 2), G_a by the transposed stencil with
// statically known taps, dz = G_a * relu6'(z) parked in LDS [pixel][channel] for the matrix cores:
//   MODE 1: P1 += dz^T X, Gram += X^T X (v_mfma_f32_16x16x4_f32, reduction over pixels, 4 per instruction), BN sums, dW_dw, masks out
//   MODE 2: dX = dz B1^T + X Q^T + bias (16 pixels x 16 input channels per accumulator tile, reduction over the C channels)
// One barrier per quad row (dz tiles double-buffered).
// ---------------------------------------------------------------------------------------------------------------------
namespace mny {

struct ExBwdArgs {
    const float* gz; const float* z; const float* z_scale; const float* z_shift; const float* z_coef;
    const float* x; const float* in_scale; const float* in_shift; int in_act;
    const float* w; const float* e_scale; const float* e_shift; const float* e_mean; const float* e_invstd; const float* w_dw;
    float* partial; float* dw_parts; unsigned short* mask;
    const float* B1; const float* Q; const float* bias; const float* addend; float* dx;
    ExGeom g;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K> struct ExB {
    static constexpr int KQ = K / 4, KP = K + 4, C = 6 * K, CP = C + 4, NQ = C / 4, PPB = 256 / NQ, NCOLS = 2 * PPB, NPX = 2 * NCOLS;
    static constexpr int NT = C / 16, KT = (K + 15) / 16;
    static constexpr int XROWS = 2 * kExTH, XPIX = XROWS * NCOLS;
    static constexpr int DZPX = (NPX + 15) / 16 * 16;                     // rows of a dz buffer: MODE 2 reads whole 16-pixel tiles
    static constexpr int XS_FLOATS = (XPIX + DZPX - NPX + 2) * KP;        // + the pixels the last padded tile reads past the X tile
    static constexpr int DZ_FLOATS = DZPX * CP;
    static constexpr int CST_F4 = 14 * NQ;
    static constexpr size_t LDS = (size_t)(XS_FLOATS + 2 * DZ_FLOATS) * 4 + (size_t)CST_F4 * 16;
    static_assert(2 * DZ_FLOATS >= C * K + K * K, "the end-of-kernel fold of P1 / Gram lives in the dz buffers");
    static_assert(XS_FLOATS >= 256 * 8, "the end-of-kernel folds of the per-thread sums live in the X tile");
};

template <int K, int XF, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void exdw_bwd_s2_kernel(ExBwdArgs p) {
    using B = ExB<K>;
    constexpr int KQ = B::KQ, KP = B::KP, C = B::C, CP = B::CP, NQ = B::NQ, PPB = B::PPB, NCOLS = B::NCOLS, NPX = B::NPX, NT = B::NT, KT = B::KT;
    constexpr int ST = 256 / KQ * KQ, PS = ST / KQ, NSB = (B::XPIX * KQ + ST - 1) / ST;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                                     // [XROWS][NCOLS][KP] (+ slack)
    float* dzs = lds + B::XS_FLOATS;                                     // [2][DZPX][CP]
    float4* cst = reinterpret_cast<float4*>(dzs + 2 * B::DZ_FLOATS);     // [14][NQ]: 9 taps, z scale / shift, ca, cb, cc
    const ExGeom& g = p.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lg = lane >> 4;
    const int q = tid % NQ, pp = tid / NQ;
    const bool worker = pp < PPB;
    const int ppa = worker ? pp : 0;
    const int c = 4 * q;

    for (int i = tid; i < B::XS_FLOATS + 2 * B::DZ_FLOATS; i += 256) lds[i] = 0.f;     // pads, slack and the padded dz rows stay zero
    if (pp == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) cst[t * NQ + q] = make_float4(p.w_dw[(c + 0) * 9 + t], p.w_dw[(c + 1) * 9 + t], p.w_dw[(c + 2) * 9 + t], p.w_dw[(c + 3) * 9 + t]);
        cst[9 * NQ + q] = ld4(p.z_scale + c);
        cst[10 * NQ + q] = ld4(p.z_shift + c);
        cst[11 * NQ + q] = ld4(p.z_coef + c);
        cst[12 * NQ + q] = ld4(p.z_coef + C + c);
        cst[13 * NQ + q] = ld4(p.z_coef + 2 * C + c);
    }
    // MODE 1 state
    v2f w01[MODE == 1 ? K : 1], w23[MODE == 1 ? K : 1];
    F4P esc = f4p0(), esh = f4p0(), emu = f4p0(), eis = f4p0();
    F4P wp[9], s1 = f4p0(), s2 = f4p0();
    f32x4 accP[NT][KT], accG[KT][KT];
    float s3a[KT];
    // MODE 2 state
    float Breg[MODE == 2 ? KT : 1][MODE == 2 ? NT : 1][4], Qreg[MODE == 2 ? KT : 1][MODE == 2 ? KT : 1][4], biasr[KT];
    if constexpr (MODE == 1) {
        ex_load_w<K>(p.w, c, w01, w23);
        esc = f4p(ld4(p.e_scale + c)); esh = f4p(ld4(p.e_shift + c)); emu = f4p(ld4(p.e_mean + c)); eis = f4p(ld4(p.e_invstd + c));
#pragma unroll
        for (int t = 0; t < 9; ++t) wp[t] = f4p0();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) accP[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            s3a[a] = 0.f;
#pragma unroll
            for (int b = 0; b < KT; ++b) accG[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int k = 16 * kt + l16;
            const bool kv = k < K;
            biasr[kt] = kv ? p.bias[k] : 0.f;
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int v = 0; v < 4; ++v) Breg[kt][s][v] = kv ? p.B1[(int64_t)k * C + 16 * s + 4 * lg + v] : 0.f;
#pragma unroll
            for (int s = 0; s < KT; ++s)
#pragma unroll
                for (int v = 0; v < 4; ++v) { const int k2 = 16 * s + 4 * lg + v; Qreg[kt][s][v] = (kv && k2 < K) ? p.Q[k * K + k2] : 0.f; }
        }
    }
    const int kq_s = tid % KQ, pix_s = tid / KQ;
    const bool stager = tid < ST;
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    const int Ho = g.Ho, Wo = g.Wo;

    for (int item = ex_lb(); item < g.items; item += gridDim.x) {
        const int ct = item % g.nCT; const int tt = item / g.nCT;
        const int hs = tt % g.nHS; const int n = tt / g.nHS;
        const int i0 = hs * kExTH, j0 = ct * PPB;
        const int i1 = min(i0 + kExTH, Ho);
        __syncthreads();                                                 // the previous item's matrix phase is done with xs / dzs
        {
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
                const bool valid = 2 * i0 + rr < g.H && 2 * j0 + cc < g.W;     // zeros outside the image: they enter Gram / colsum / P1
                if (stager && pix < B::XPIX) *reinterpret_cast<float4*>(&xs[pix * KP + 4 * kq_s]) = valid ? ex_xf<XF>(stg[i], xsc, xsh, slope, hi) : f4zero();
            }
        }
        __syncthreads();
        const int j = j0 + ppa;
        const bool colv = worker && j < Wo;
        const float am = colv ? 1.f : 0.f;
        const float jm1 = (j + 1 < Wo) ? am : 0.f;
        const int jc0 = min(j, Wo - 1), jc1 = min(j + 1, Wo - 1);
        int lo = q;
        asm volatile("" : "+v"(lo));                                     // keeps the LDS constant reads where they are used (see dwbwd.hip)
        const float4* my = cst + lo;
        auto dz_fetch = [&](int ho, float4 (&r)[4]) {
            const int64_t ro = ((int64_t)n * Ho + min(ho, Ho - 1)) * Wo;
            r[0] = ld4(p.gz + (ro + jc0) * C + c); r[1] = ld4(p.z + (ro + jc0) * C + c);
            r[2] = ld4(p.gz + (ro + jc1) * C + c); r[3] = ld4(p.z + (ro + jc1) * C + c);
        };
        auto dz2 = [&](v2f gv, v2f zv, v2f s, v2f h, v2f a, v2f b, v2f cterm) {       // dZ = ca * (G * relu6'(s z + h)) + cb * z + cc
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
        unsigned short* mrow = p.mask + ((int64_t)item * kExTH) * 256 + tid;
        unsigned bits_next = 0;
        if constexpr (MODE == 2) bits_next = mrow[0];
        for (int i = i0; i < i1; ++i) {
            const int li = i - i0;
            float* dzb = dzs + (li & 1) * B::DZ_FLOATS;
            F4P d10, d11;
            dz_finish(i + 1, rawn, d10, d11);
            if (i + 1 < i1) dz_fetch(i + 2, rawn);
            unsigned bits = bits_next;
            if constexpr (MODE == 2) { if (i + 1 < i1) bits_next = mrow[(li + 1) * 256]; }
            // data gradient of the depthwise conv at the quad (2i..2i+1, 2j..2j+1): taps as in dw_bwd_data_s2k3_kernel
            F4P o[4] = {f4p0(), f4p0(), f4p0(), f4p0()};
#define WG(t) f4p(my[(t) * NQ])
            pfma(o[0], d00, WG(4));
            pfma(o[1], d00, WG(5)); pfma(o[1], d01, WG(3));
            pfma(o[2], d00, WG(7)); pfma(o[2], d10, WG(1));
            pfma(o[3], d00, WG(8)); pfma(o[3], d01, WG(6)); pfma(o[3], d10, WG(2)); pfma(o[3], d11, WG(0));
#undef WG
            const float* xq = xs + ((2 * li) * NCOLS + 2 * ppa) * KP;
            if constexpr (MODE == 1) {
                F4P a[4];
                bits = 0;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    v2f y01, y23;
                    ex_y<K>(xq + ((px >> 1) * NCOLS + (px & 1)) * KP, w01, w23, y01, y23);
                    const v2f z0 = __builtin_elementwise_fma(y01, esc.lo, esh.lo), z1 = __builtin_elementwise_fma(y23, esc.hi, esh.hi);
                    a[px].lo = v2f{__builtin_amdgcn_fmed3f(z0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z0.y, 0.f, 6.f)};
                    a[px].hi = v2f{__builtin_amdgcn_fmed3f(z1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z1.y, 0.f, 6.f)};
                    const bool m0 = z0.x > 0.f && z0.x < 6.f, m1 = z0.y > 0.f && z0.y < 6.f, m2 = z1.x > 0.f && z1.x < 6.f, m3 = z1.y > 0.f && z1.y < 6.f;
                    bits |= ((m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u)) << (4 * px);
                    o[px].lo = v2f{m0 ? o[px].lo.x : 0.f, m1 ? o[px].lo.y : 0.f};
                    o[px].hi = v2f{m2 ? o[px].hi.x : 0.f, m3 ? o[px].hi.y : 0.f};
                    // BN-backward sums of the expand unit: sum dz, sum dz * yhat
                    s1.lo += o[px].lo; s1.hi += o[px].hi;
                    const v2f h0 = (y01 - emu.lo) * eis.lo, h1 = (y23 - emu.hi) * eis.hi;
                    s2.lo = __builtin_elementwise_fma(o[px].lo, h0, s2.lo); s2.hi = __builtin_elementwise_fma(o[px].hi, h1, s2.hi);
                }
                // weight gradient of the depthwise conv over the input pixels this thread owns (dwbwd.hip, stride-2 kernel)
                pfma(wp[4], a[0], d00);
                pfma(wp[5], a[1], d00); pfma(wp[3], a[1], d01);
                pfma(wp[7], a[2], d00); pfma(wp[1], a[2], d10);
                pfma(wp[8], a[3], d00); pfma(wp[6], a[3], d01); pfma(wp[2], a[3], d10); pfma(wp[0], a[3], d11);
                mrow[li * 256] = (unsigned short)bits;
            } else {
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const unsigned b = bits >> (4 * px);
                    o[px].lo = v2f{(b & 1u) ? o[px].lo.x : 0.f, (b & 2u) ? o[px].lo.y : 0.f};
                    o[px].hi = v2f{(b & 4u) ? o[px].hi.x : 0.f, (b & 8u) ? o[px].hi.y : 0.f};
                }
            }
            if (worker) {
#pragma unroll
                for (int px = 0; px < 4; ++px) *reinterpret_cast<float4*>(&dzb[((px >> 1) * NCOLS + 2 * pp + (px & 1)) * CP + c]) = f4u(o[px]);
            }
            __syncthreads();
            if constexpr (MODE == 1) {
                for (int gq = wave; gq < NPX / 4; gq += 4) {
                    const int px = 4 * gq + lg;
                    const float* dzp = dzb + px * CP + l16;
                    const float* xp = xs + ((2 * li) * NCOLS + px) * KP + l16;
                    float b[KT];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) b[kt] = xp[16 * kt];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float a = dzp[16 * t];
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt) accP[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[kt], accP[t][kt], 0, 0, 0);
                    }
#pragma unroll
                    for (int ka = 0; ka < KT; ++ka) {
                        s3a[ka] += b[ka];
#pragma unroll
                        for (int kb = 0; kb < KT; ++kb) accG[ka][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ka], b[kb], accG[ka][kb], 0, 0, 0);
                    }
                }
            } else {
                for (int T = wave; T < B::DZPX / 16; T += 4) {
                    f32x4 accD[KT];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) accD[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float* dzp = dzb + (16 * T + l16) * CP + 4 * lg;
#pragma unroll
                    for (int s = 0; s < NT; ++s) {
                        const float4 a4 = *reinterpret_cast<const float4*>(dzp + 16 * s);
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) accD[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[v], Breg[kt][s][v], accD[kt], 0, 0, 0);
                    }
                    const float* xp = xs + ((2 * li) * NCOLS + 16 * T + l16) * KP + 4 * lg;
#pragma unroll
                    for (int s = 0; s < KT; ++s) {
                        const float4 x4 = *reinterpret_cast<const float4*>(xp + 16 * s);
                        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) accD[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[v], Qreg[kt][s][v], accD[kt], 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int px = 16 * T + 4 * lg + v;
                        const int rr = px / NCOLS, cc = px - rr * NCOLS;
                        const int gc = 2 * j0 + cc;
                        if (px < NPX && gc < g.W) {
                            const int64_t base = (((int64_t)n * g.H + 2 * i + rr) * g.W + gc) * K;
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
                                const int k = 16 * kt + l16;
                                if (k < K) {
                                    float r = accD[kt][v] + biasr[kt];
                                    if (p.addend) r += p.addend[base + k];
                                    p.dx[base + k] = r;
                                }
                            }
                        }
                    }
                }
            }
            d00 = d10; d01 = d11;
        }
    }
    if constexpr (MODE == 1) {
        // per-block partial row  P1[C*K] | Gram[K*K] | s1[C] | s2[C] | s3[K]  (bnw_stride) + the depthwise weight-gradient row [C*9]
        float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(C, K);
        float* fold = dzs;
        for (int w = 0; w < 4; ++w) {                                    // the four waves' matrix accumulators, added in wave order
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int nn = 16 * t + 4 * lg + v, k = 16 * kt + l16;
                            if (k < K) fold[nn * K + k] = (w == 0 ? 0.f : fold[nn * K + k]) + accP[t][kt][v];
                        }
#pragma unroll
                for (int ka = 0; ka < KT; ++ka)
#pragma unroll
                    for (int kb = 0; kb < KT; ++kb)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int r = 16 * ka + 4 * lg + v, k = 16 * kb + l16;
                            if (r < K && k < K) fold[C * K + r * K + k] = (w == 0 ? 0.f : fold[C * K + r * K + k]) + accG[ka][kb][v];
                        }
            }
        }
        float4* red = reinterpret_cast<float4*>(xs);
        float* s3buf = xs + 256 * 8;
        red[tid * 2 + 0] = worker ? f4u(s1) : f4zero();
        red[tid * 2 + 1] = worker ? f4u(s2) : f4zero();
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) s3buf[(wave * 4 + lg) * (16 * KT) + 16 * kt + l16] = s3a[kt];
        __syncthreads();
        for (int e = tid; e < C * K + K * K; e += 256) dst[e] = fold[e];
        if (pp == 0) {
            float4 a = f4zero(), b = f4zero();
            for (int i = 0; i < PPB; ++i) { add4(a, red[(i * NQ + q) * 2]); add4(b, red[(i * NQ + q) * 2 + 1]); }
            st4(dst + C * K + K * K + c, a);
            st4(dst + C * K + K * K + C + c, b);
        }
        if (tid < K) {
            float a = 0.f;
            for (int i = 0; i < 16; ++i) a += s3buf[i * (16 * KT) + tid];
            dst[C * K + K * K + 2 * C + tid] = a;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            red[tid] = worker ? f4u(wp[t]) : f4zero();
            __syncthreads();
            if (pp == 0) {
                float4 a = f4zero();
                for (int i = 0; i < PPB; ++i) add4(a, red[i * NQ + q]);
                float* dd = p.dw_parts + (int64_t)blockIdx.x * C * 9;
                dd[(c + 0) * 9 + t] = a.x; dd[(c + 1) * 9 + t] = a.y; dd[(c + 2) * 9 + t] = a.z; dd[(c + 3) * 9 + t] = a.w;
            }
        }
    }
}

struct ExWs { size_t partials, red, B1, Q, bias, mask, total; };
static ExWs ex_ws(const ExGeom& g, int grid) {
    ExWs w;
    const size_t stride = (size_t)bnw_stride(g.C, g.K);
    w.partials = 0;
    w.red = (size_t)grid * stride;
    w.B1 = w.red + stride;
    w.Q = w.B1 + (size_t)g.C * g.K;
    w.bias = w.Q + (size_t)g.K * g.K;
    w.mask = (w.bias + 64 + 3) / 4 * 4;
    w.total = w.mask + (size_t)g.items * kExTH * 128;          // 256 x 16-bit mask words per quad row = 128 floats
    return w;
}

template <int K>
static int ex_bwd_launch(const ExBwdArgs& a, bool xf, int grid, hipStream_t st, int mode) {
    const size_t lds = ExB<K>::LDS;
    static bool attr = false;
    if (!attr) {
        const void* ks[4] = {(const void*)exdw_bwd_s2_kernel<K, 0, 1>, (const void*)exdw_bwd_s2_kernel<K, 1, 1>, (const void*)exdw_bwd_s2_kernel<K, 0, 2>,
                             (const void*)exdw_bwd_s2_kernel<K, 1, 2>};
        for (const void* k : ks)
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { set_error("exdw_bwd: hipFuncSetAttribute failed"); return MNY_EHIP; }
        attr = true;
    }
    if (mode == 1) {
        if (xf) hipLaunchKernelGGL((exdw_bwd_s2_kernel<K, 1, 1>), dim3(grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((exdw_bwd_s2_kernel<K, 0, 1>), dim3(grid), dim3(256), lds, st, a);
    } else {
        if (xf) hipLaunchKernelGGL((exdw_bwd_s2_kernel<K, 1, 2>), dim3(grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL((exdw_bwd_s2_kernel<K, 0, 2>), dim3(grid), dim3(256), lds, st, a);
    }
    return check_launch(mode == 1 ? "exdw_bwd_s2_kernel<1>" : "exdw_bwd_s2_kernel<2>");
}

}  // namespace mny

extern "C" int mny_exdw_bwd_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_grid(ex_geom(N, H, W, K, C), 2);
}

extern "C" size_t mny_exdw_bwd_ws_floats(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return 0;
    const ExGeom g = ex_geom(N, H, W, K, C);
    return ex_ws(g, ex_grid(g, 2)).total;
}

extern "C" int mny_exdw_bwd(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                            const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                            const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd, const float* e_gamma,
                            const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                            float* dw_dw, float* dw_ws, float* ws, int N, int H, int W, int K, int C, int stride, void* stream) {
    MNY_REQUIRE(gz && z && z_scale && z_shift && z_coef && x && w_exp && e_scale && e_shift && e_mean && e_invstd && e_gamma && w_dw && dx && dw_exp &&
                    dgamma_e && dbeta_e && dw_ws && ws, "exdw_bwd: null pointer");
    MNY_REQUIRE(ex_shape_ok(N, H, W, K, C, stride), "exdw_bwd: N=%d H=%d W=%d K=%d C=%d stride=%d not supported", N, H, W, K, C, stride);
    MNY_REQUIRE(z_act == MNY_ACT_RELU6, "exdw_bwd: the depthwise unit's activation must be ReLU6 (got %d)", z_act);
    MNY_REQUIRE(!in_scale == !in_shift, "exdw_bwd: scale and shift come together");
    MNY_REQUIRE(in_act <= MNY_ACT_RELU, "exdw_bwd: unsupported input activation %d", in_act);
    const ExGeom g = ex_geom(N, H, W, K, C);
    const int grid = ex_grid(g, 2);
    const ExWs o = ex_ws(g, grid);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    hipStream_t st = (hipStream_t)stream;
    ExBwdArgs a{gz, z, z_scale, z_shift, z_coef, x, in_scale, in_shift, in_act, w_exp, e_scale, e_shift, e_mean, e_invstd, w_dw,
                ws + o.partials, dw_ws, reinterpret_cast<unsigned short*>(ws + o.mask), ws + o.B1, ws + o.Q, ws + o.bias, addend, dx, g};
    int rc = K == 16 ? ex_bwd_launch<16>(a, xf, grid, st, 1) : (K == 24 ? ex_bwd_launch<24>(a, xf, grid, st, 1) : ex_bwd_launch<32>(a, xf, grid, st, 1));
    if (rc) return rc;
    rc = pw_bnbwd_finalize_launch(ws + o.partials, grid, ws + o.red, w_exp, e_gamma, e_mean, e_invstd, (int64_t)N * H * W, C, K, dw_exp, dgamma_e, dbeta_e,
                                  ws + o.B1, ws + o.Q, ws + o.bias, st);
    if (rc) return rc;
    if (dw_dw) { rc = launch_reduce_parts(dw_ws, grid, C * 9, dw_dw, st); if (rc) return rc; }
    return K == 16 ? ex_bwd_launch<16>(a, xf, grid, st, 2) : (K == 24 ? ex_bwd_launch<24>(a, xf, grid, st, 2) : ex_bwd_launch<32>(a, xf, grid, st, 2));
}
