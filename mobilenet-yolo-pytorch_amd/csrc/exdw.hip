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

// ---- backward (below) -------------------------------------------------------------------------------------------------
extern "C" int mny_exdw_bwd_parts(int N, int H, int W, int K, int C, int stride) {
    if (!ex_shape_ok(N, H, W, K, C, stride)) return MNY_EINVAL;
    return ex_grid(ex_geom(N, H, W, K, C), 2);
}
extern "C" size_t mny_exdw_bwd_ws_floats(int N, int H, int W, int K, int C, int stride) { return 0; }
extern "C" int mny_exdw_bwd(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                            const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                            const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd, const float* e_gamma,
                            const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                            float* dw_dw, float* dw_ws, float* ws, int N, int H, int W, int K, int C, int stride, void* stream) {
    set_error("exdw_bwd: not built yet");
    return MNY_EUNSUPPORTED;
}
