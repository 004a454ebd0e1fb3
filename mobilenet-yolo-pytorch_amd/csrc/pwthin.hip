// Pointwise (1x1) convolution with a SHORT reduction (K = 8 / 16 / 24 / 32 input channels) as a vector-ALU kernel.
//
// These are the "expand" convs of the early inverted-residual blocks (16->96, 24->144, 32->192: mobilenetv2.py:61) and the data
// gradients of the matching "project" convs (contraction over 16 / 24 / 32 output channels: autograd of mobilenetv2.py:69,83), all
// at the largest feature maps.  They carry 8-16 flops per output byte: pure HBM streams.  On the matrix-core tile kernel
// (pwgemm.hip) they reached 2.8-4.7 of the 5.5-6 TB/s a flat copy gets on this part (profiles/r02_kernels_time_and_hbm.md):
// a 16/24/32-deep reduction is one or two MFMA k-steps, so the tile machinery (LDS ring, barriers per k-step, DPP transposes in
// the epilogue, 32-column padding of 144 = 4.5 tiles) is all overhead.  Here instead:
//   * a thread owns 4 output columns and keeps their 4 x K weights in registers for the whole launch (persistent blocks);
//   * a block = `rpb` rows x N/4 column quads (<= 256 threads) and walks row tiles of R*rpb rows: the tile's A rows are loaded
//     once (16-B chunks, BN-apply + activation of the producing unit applied once per element), parked in LDS (double-buffered,
//     one barrier per tile) and read back as broadcasts by the N/4 threads of a row;
//   * 2K packed FMAs (v_pk_fma_f32) per thread-row, one 16-B streaming store: ~20 instructions per output quad, below what the
//     store stream needs to stay at HBM rate;
//   * the column statistics (forward: sum / sum of squares of the stored outputs; RED: the BN-backward sums of the unit the
//     gradient belongs to, pwgemm.hip RED = 1 / 2 semantics) are per-thread running sums, folded once at the end.
#include "common.h"

#include <type_traits>

namespace mny {

struct ThinArgs {
    const void* A; const float* in_scale; const float* in_shift; int in_act;
    const void* W; const float* bias; const void* addend; void* C; float* stats;
    int64_t M; int N; int nq; int rpb; int R; int64_t ntiles;
    const void* rY; const float* r_scale; const float* r_shift; const float* r_mean; const float* r_invstd; int r_act;
};

// four stored elements as they sit in memory: widening a bf16 load where it is issued would put the wait for it there too
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float4 type; };
template <> struct Raw4<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 widen(float4 v) { return v; }
__device__ __forceinline__ float4 widen(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

constexpr int kThinTileRows = 128;          // upper bound on R*rpb (LDS: 2 x 128 x (K+4) floats = 36 KB at K = 32)

// XF: 0 = A as is, 1 = scale/shift + min(max(z, slope z), hi), 2 = scale/shift + hswish.
// RED: 0 = plain (optional column statistics), 1 = BN-backward sums of (C, rY), 2 = same with C = product + addend.
template <typename T, int K, int XF, int RED, int R, bool STATS, bool HAS_ADD>
__global__ __launch_bounds__(256) void pw_thin_kernel(ThinArgs p) {
    constexpr int KQ = K / 4;
    constexpr int LDA = K + 4;                       // row stride in LDS: consecutive rows start 4 banks apart
    constexpr int NCH = (kThinTileRows + 192 / KQ - 1) / (192 / KQ);     // staging passes per tile at the smallest workgroup (193 threads)
    __shared__ __attribute__((aligned(16))) float sA[2][kThinTileRows * LDA];

    const T* pA = (const T*)p.A;
    const T* pW = (const T*)p.W;
    const T* pAdd = (const T*)p.addend;
    const T* pY = (const T*)p.rY;
    T* pC = (T*)p.C;
    const int tid = threadIdx.x;
    const int nq = p.nq, rpb = p.rpb, N = p.N;
    const int tile_rows = R * rpb;
    const int r = tid / nq;                          // blockDim.x = rpb*nq exactly (a partial last wave is masked by the hardware, not by branches)
    const int rr = r;
    const int q = tid - r * nq;
    const int n0 = 4 * q;

    // the thread's weights: w01[k] = {W[n0][k], W[n0+1][k]}, w23[k] = {W[n0+2][k], W[n0+3][k]}
    v2f w01[K], w23[K];
    {
        const int nn = n0;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const float4 a = ld4(pW + (int64_t)(nn + 0) * K + 4 * kq), b = ld4(pW + (int64_t)(nn + 1) * K + 4 * kq);
            const float4 c = ld4(pW + (int64_t)(nn + 2) * K + 4 * kq), d = ld4(pW + (int64_t)(nn + 3) * K + 4 * kq);
            w01[4 * kq + 0] = v2f{a.x, b.x}; w01[4 * kq + 1] = v2f{a.y, b.y}; w01[4 * kq + 2] = v2f{a.z, b.z}; w01[4 * kq + 3] = v2f{a.w, b.w};
            w23[4 * kq + 0] = v2f{c.x, d.x}; w23[4 * kq + 1] = v2f{c.y, d.y}; w23[4 * kq + 2] = v2f{c.z, d.z}; w23[4 * kq + 3] = v2f{c.w, d.w};
        }
    }
    float4 bias4 = f4zero();
    if (p.bias) bias4 = ld4(p.bias + n0);

    // staging role: 16-B chunk kq_s of rows row_s + i*RP, RP = blockDim.x / KQ rows per pass (threads past RP*KQ do not stage)
    const int RP = (int)blockDim.x / KQ;
    const int row_s = tid / KQ, kq_s = tid - row_s * KQ;
    const bool stager = row_s < RP;
    const int nch = (tile_rows + RP - 1) / RP;       // <= NCH
    float4 xsc = f4one(), xsh = f4zero();
    if (XF && p.in_scale) { xsc = ld4(p.in_scale + 4 * kq_s); xsh = ld4(p.in_shift + 4 * kq_s); }
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);

    typedef typename Raw4<T>::type raw_t;
    raw_t stg[NCH];
    // addresses are (uniform per-tile base in SGPRs) + (32-bit offset inside the tile): no 64-bit vector arithmetic per row
    auto fetch = [&](int64_t tile) {                 // global -> registers (raw); no predicated load: out-of-range slots re-read a valid row
        const int64_t base = tile * tile_rows;
        const T* ta = pA + base * K;
        const int64_t left = p.M - 1 - base;
        const int lim = (int)(left < tile_rows - 1 ? left : tile_rows - 1);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i < nch) {                           // uniform
                int tr = row_s + i * RP;
                tr = tr < lim ? tr : lim;
                stg[i] = ldraw(at_bytes(ta, (unsigned)(tr * K + 4 * kq_s) * (unsigned)sizeof(T)));
            }
        }
    };
    auto park = [&](int buf) {                       // registers -> LDS (transformed)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i < nch) {
                const int tr = row_s + i * RP;
                float4 v = widen(stg[i]);
                if (XF == 1) {
                    v.x = fmaf(v.x, xsc.x, xsh.x); v.y = fmaf(v.y, xsc.y, xsh.y); v.z = fmaf(v.z, xsc.z, xsh.z); v.w = fmaf(v.w, xsc.w, xsh.w);
                    v.x = fminf(fmaxf(v.x, slope * v.x), hi); v.y = fminf(fmaxf(v.y, slope * v.y), hi);
                    v.z = fminf(fmaxf(v.z, slope * v.z), hi); v.w = fminf(fmaxf(v.w, slope * v.w), hi);
                } else if (XF == 2) {
                    v = xform4(v, xsc, xsh, MNY_ACT_HSWISH);
                }
                // bf16 plans: the operand is the value a materialised bf16 activation would hold — what the
                // matrix-core path and the weight-gradient kernels multiply with; keep the three consistent
                if (XF && sizeof(T) == 2) v = stored4<T>(v);
                if (stager && tr < tile_rows) *reinterpret_cast<float4*>(&sA[buf][tr * LDA + 4 * kq_s]) = v;
            }
        }
    };

    // RED constants of the thread's four columns
    float4 rsc = f4one(), rsh = f4zero(), rmu = f4zero(), ris = f4zero();
    if (RED) { rsc = ld4(p.r_scale + n0); rsh = ld4(p.r_shift + n0); rmu = ld4(p.r_mean + n0); ris = ld4(p.r_invstd + n0); }
    const float rslope = act_slope(p.r_act), rhi = act_hi(p.r_act);
    float4 s1 = f4zero(), s2 = f4zero();

    // One row of the thread's tile: 2K packed FMAs off the parked A row, epilogue, streaming store.  FULL: the whole tile is
    // inside M, nothing is predicated — with no branch around a store or a load the compiler can count the memory operations
    // in flight, and the wait for the NEXT tile's A rows (issued before this tile's stores) leaves the stores outstanding;
    // vmcnt is one in-order counter on gfx9, a `vmcnt(0)` per tile would drain the write stream.
    auto row_step = [&](auto full_tag, auto hsw_tag, const float* ar, T* crow, bool ok, raw_t yraw, raw_t araw) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool HSW = decltype(hsw_tag)::value;
        v2f c01 = v2f{0.f, 0.f}, c23 = v2f{0.f, 0.f};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const float4 a = *reinterpret_cast<const float4*>(ar + 4 * kq);
            c01 = __builtin_elementwise_fma(v2f{a.x, a.x}, w01[4 * kq + 0], c01); c23 = __builtin_elementwise_fma(v2f{a.x, a.x}, w23[4 * kq + 0], c23);
            c01 = __builtin_elementwise_fma(v2f{a.y, a.y}, w01[4 * kq + 1], c01); c23 = __builtin_elementwise_fma(v2f{a.y, a.y}, w23[4 * kq + 1], c23);
            c01 = __builtin_elementwise_fma(v2f{a.z, a.z}, w01[4 * kq + 2], c01); c23 = __builtin_elementwise_fma(v2f{a.z, a.z}, w23[4 * kq + 2], c23);
            c01 = __builtin_elementwise_fma(v2f{a.w, a.w}, w01[4 * kq + 3], c01); c23 = __builtin_elementwise_fma(v2f{a.w, a.w}, w23[4 * kq + 3], c23);
        }
        float4 o = make_float4(c01.x + bias4.x, c01.y + bias4.y, c23.x + bias4.z, c23.y + bias4.w);
        if (HAS_ADD) add4(o, widen(araw));
        if (FULL || ok) {
            st4_stream(crow, o);
            if (RED) {
                const float4 so = stored4<T>(o);
                const float4 ycur = widen(yraw);
                float4 z = make_float4(fmaf(ycur.x, rsc.x, rsh.x), fmaf(ycur.y, rsc.y, rsh.y), fmaf(ycur.z, rsc.z, rsh.z), fmaf(ycur.w, rsc.w, rsh.w));
                float4 d;
                if (HSW) {                           // h-swish': 0 below -3, 1 above 3, (2z + 3) / 6 between
                    const float4 t = make_float4(fmaf(z.x, 1.f / 3.f, 0.5f), fmaf(z.y, 1.f / 3.f, 0.5f), fmaf(z.z, 1.f / 3.f, 0.5f), fmaf(z.w, 1.f / 3.f, 0.5f));
                    d.x = z.x <= -3.f ? 0.f : (z.x >= 3.f ? 1.f : t.x); d.y = z.y <= -3.f ? 0.f : (z.y >= 3.f ? 1.f : t.y);
                    d.z = z.z <= -3.f ? 0.f : (z.z >= 3.f ? 1.f : t.z); d.w = z.w <= -3.f ? 0.f : (z.w >= 3.f ? 1.f : t.w);
                } else {
                    d.x = (z.x > 0.f ? 1.f : rslope) * (z.x < rhi ? 1.f : 0.f); d.y = (z.y > 0.f ? 1.f : rslope) * (z.y < rhi ? 1.f : 0.f);
                    d.z = (z.z > 0.f ? 1.f : rslope) * (z.z < rhi ? 1.f : 0.f); d.w = (z.w > 0.f ? 1.f : rslope) * (z.w < rhi ? 1.f : 0.f);
                }
                const float4 dz = make_float4(so.x * d.x, so.y * d.y, so.z * d.z, so.w * d.w);
                add4(s1, dz);
                s2.x = fmaf(dz.x, (ycur.x - rmu.x) * ris.x, s2.x); s2.y = fmaf(dz.y, (ycur.y - rmu.y) * ris.y, s2.y);
                s2.z = fmaf(dz.z, (ycur.z - rmu.z) * ris.z, s2.z); s2.w = fmaf(dz.w, (ycur.w - rmu.w) * ris.w, s2.w);
            } else if (STATS) {
                const float4 so = stored4<T>(o);
                add4(s1, so);
                fma4(s2, so, so);
            }
        }
    };
    const unsigned off0 = (unsigned)(rr * N + n0) * (unsigned)sizeof(T), ostep = (unsigned)(rpb * N) * (unsigned)sizeof(T);   // byte offsets inside a tile (< 128 KB)
    auto tile_step = [&](auto full_tag, auto hsw_tag, int64_t base, const float* ab) {
        constexpr bool FULL = decltype(full_tag)::value;
        T* tc = pC + base * N;                       // uniform
        const T* ty = RED ? pY + base * N : nullptr;
        const T* tadd = HAS_ADD ? pAdd + base * N : nullptr;
        const int rows_here = FULL ? tile_rows : (int)(p.M - base);
        // the fed unit's raw output (and the addend) of row j + PD are requested while row j is computed: one row of arithmetic
        // (~100 instructions) does not cover a memory round trip, three of them nearly do (bf16 24 -> 144: 0.51 -> 0.40 ms)
        constexpr int PD = !(RED || HAS_ADD) ? 0 : (sizeof(T) == 2 && R >= 3 ? 3 : 1);      // fp32: the rows are twice as long, and K = 32 has no registers to spare
        raw_t yq[PD > 0 ? PD : 1], aq[PD > 0 ? PD : 1];
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            yq[j] = raw_t(); aq[j] = raw_t();
            const bool ok = FULL || rr + j * rpb < rows_here;
            unsigned offc = off0 + j * ostep;
            asm volatile("" : "+v"(offc));
            if (RED && ok) yq[j] = ldraw(at_bytes(ty, offc));
            if (HAS_ADD && ok) aq[j] = ldraw(at_bytes(tadd, offc));
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            unsigned off = off0 + j * ostep;
            asm volatile("" : "+v"(off));            // keep (uniform base, 32-bit offset): re-associated 64-bit row pointers cost 6 VGPRs per row
            const raw_t ycur = yq[0], acur = aq[0];
#pragma unroll
            for (int d = 0; d + 1 < PD; ++d) { yq[d] = yq[d + 1]; aq[d] = aq[d + 1]; }
            if constexpr (PD > 0) {               // `if constexpr`: a plain `if` still instantiates yq[-1] for PD = 0 (1 536 -Warray-bounds lines per build)
                yq[PD - 1] = raw_t(); aq[PD - 1] = raw_t();
                if (j + PD < R) {
                    const bool okn = FULL || rr + (j + PD) * rpb < rows_here;
                    unsigned offn = off + PD * ostep;
                    asm volatile("" : "+v"(offn));
                    if (RED && okn) yq[PD - 1] = ldraw(at_bytes(ty, offn));
                    if (HAS_ADD && okn) aq[PD - 1] = ldraw(at_bytes(tadd, offn));
                }
            }
            row_step(full_tag, hsw_tag, ab + j * rpb * LDA, at_bytes(tc, off), FULL || rr + j * rpb < rows_here, ycur, acur);
            if (FULL) __builtin_amdgcn_sched_barrier(0);          // rows in order: hoisting every prefetch of the tile costs ~100 VGPRs
        }
    };

    // Full tiles in the loop (branch-free), the ragged last tile — if this workgroup owns it — after it.  Per iteration: fetch the
    // next tile's A rows, compute + store this tile, then park the fetched rows: the wait in front of the park is vmcnt(R ...),
    // it leaves this tile's stores in flight.
    const int64_t stride = gridDim.x;
    const int64_t nfull = p.M / tile_rows;
    int64_t tile = blockIdx.x;
    int buf = 0;
    if (tile < p.ntiles) { fetch(tile); park(0); }
    __syncthreads();
    const bool hsw = RED && p.r_act == MNY_ACT_HSWISH;           // the derivative family is chosen per tile, not per element
    for (; tile < nfull; tile += stride) {
        const bool has_next = tile + stride < p.ntiles;
        if (has_next) fetch(tile + stride);
        if (RED && hsw) tile_step(std::true_type{}, std::true_type{}, tile * tile_rows, &sA[buf][rr * LDA]);
        else tile_step(std::true_type{}, std::false_type{}, tile * tile_rows, &sA[buf][rr * LDA]);
        if (has_next) park(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (tile < p.ntiles) {
        if (RED && hsw) tile_step(std::false_type{}, std::true_type{}, tile * tile_rows, &sA[buf][rr * LDA]);
        else tile_step(std::false_type{}, std::false_type{}, tile * tile_rows, &sA[buf][rr * LDA]);
    }
    __syncthreads();

    if (RED || STATS) {                              // fold the rpb row slots of every column; [gridDim.x][2][N]
        float* red = &sA[0][0];                      // [rpb][N][2] <= 256*4*2 floats
        {
            float* d = red + (r * N + n0) * 2;
            d[0] = s1.x; d[1] = s2.x; d[2] = s1.y; d[3] = s2.y; d[4] = s1.z; d[5] = s2.z; d[6] = s1.w; d[7] = s2.w;
        }
        __syncthreads();
        if (tid < N) {
            float a = 0.f, b = 0.f;
            for (int i = 0; i < rpb; ++i) { a += red[(i * N + tid) * 2]; b += red[(i * N + tid) * 2 + 1]; }
            p.stats[(int64_t)blockIdx.x * 2 * N + tid] = a;
            p.stats[(int64_t)blockIdx.x * 2 * N + N + tid] = b;
        }
    }
}

struct ThinPlan { int nq, rpb, R, grid; int64_t ntiles; };

// Which problems take this kernel (measured per shape on MI355X against the matrix-core tile kernel, tools/bench_thin.py):
//   fp32: every K in {8,16,24,32} — 4.9-5.7 TB/s here against 3.0-4.9 there;
//   bf16: half the bytes for the same instruction count, so the kernel is issue-bound near 3 TB/s: it wins where the bf16 MFMA
//         k-step (32) is badly filled — K = 8 / 24, and the K = 16 data gradients — and loses at K = 32 and the K = 16 forward.
bool pw_thin_ok(int bf, int red, int64_t M, int K, int N) {
    static const bool off = getenv("MNY_NO_THIN") != nullptr;      // A/B switch
    if (off) return false;
    if (!(M > 0 && (K == 8 || K == 16 || K == 24 || K == 32) && (N & 3) == 0 && N >= 16 && N <= 256)) return false;
    if (bf) return K == 8 || K == 24 || (K == 16 && red);
    return true;
}

// red != 0: the BN-backward epilogue's constants and prefetched rows push the K >= 16 variants past 168 VGPRs (two workgroups per CU)
static ThinPlan thin_plan(int64_t M, int K, int N, int red) {
    static const int res_env = getenv("MNY_THIN_RES") ? atoi(getenv("MNY_THIN_RES")) : 0;
    ThinPlan t;
    t.nq = N / 4;
    t.rpb = 256 / t.nq;
    t.R = kThinTileRows / t.rpb >= 8 ? 8 : 2;                      // the two instantiated row counts (N >= 64 : N < 64)
    t.ntiles = cdiv(M, (int64_t)t.R * t.rpb);
    const int resident = res_env > 0 ? res_env : ((K <= 8 || (K <= 16 && !red)) ? 768 : 512);
    t.grid = (int)(t.ntiles < resident ? t.ntiles : resident);
    return t;
}

int pw_thin_parts(int64_t M, int K, int N, int red) { return thin_plan(M, K, N, red).grid; }

using ThinKernel = void (*)(ThinArgs);
template <typename T, int K, int R>
static ThinKernel thin_pick(int xf, int red, bool stats, bool add) {
    if (red == 1) return pw_thin_kernel<T, K, 0, 1, R, false, false>;
    if (red == 2) return pw_thin_kernel<T, K, 0, 2, R, false, true>;
#define MNY_THIN_MODE(XF)                                                                                              \
    return stats ? (add ? (ThinKernel)pw_thin_kernel<T, K, XF, 0, R, true, true> : (ThinKernel)pw_thin_kernel<T, K, XF, 0, R, true, false>) \
                 : (add ? (ThinKernel)pw_thin_kernel<T, K, XF, 0, R, false, true> : (ThinKernel)pw_thin_kernel<T, K, XF, 0, R, false, false>)
    if (xf == 0) MNY_THIN_MODE(0);
    if (xf == 1) MNY_THIN_MODE(1);
    MNY_THIN_MODE(2);
#undef MNY_THIN_MODE
}
template <typename T>
static ThinKernel thin_pick_kr(int K, int R, int xf, int red, bool stats, bool add) {
    switch (K * 16 + R) {
        case 8 * 16 + 2: return thin_pick<T, 8, 2>(xf, red, stats, add);   case 8 * 16 + 8: return thin_pick<T, 8, 8>(xf, red, stats, add);
        case 16 * 16 + 2: return thin_pick<T, 16, 2>(xf, red, stats, add); case 16 * 16 + 8: return thin_pick<T, 16, 8>(xf, red, stats, add);
        case 24 * 16 + 2: return thin_pick<T, 24, 2>(xf, red, stats, add); case 24 * 16 + 8: return thin_pick<T, 24, 8>(xf, red, stats, add);
        case 32 * 16 + 2: return thin_pick<T, 32, 2>(xf, red, stats, add); case 32 * 16 + 8: return thin_pick<T, 32, 8>(xf, red, stats, add);
    }
    return nullptr;
}

// bf = 0: fp32 operands, 1: bf16 (A, W, addend, C, rY).  red = 0: forward (in_* view, bias, addend, stats), 1 / 2: BN-backward sums.
int pw_thin_launch(int bf, const void* A, const float* in_scale, const float* in_shift, int in_act, const void* W, const float* bias,
                   const void* addend, void* C, float* stats, int64_t M, int K, int N, int red, const void* rY, const float* r_scale,
                   const float* r_shift, const float* r_mean, const float* r_invstd, int r_act, hipStream_t st) {
    const ThinPlan t = thin_plan(M, K, N, red);
    const bool has_xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    MNY_REQUIRE(!in_scale == !in_shift, "pw_thin: scale and shift come together");
    const int xf = !has_xf ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1);
    MNY_REQUIRE(xf != 1 || in_act <= MNY_ACT_RELU, "pw_thin: unsupported input activation %d", in_act);
    MNY_REQUIRE((red == 2) == (red != 0 && addend != nullptr), "pw_thin: red = 2 is the addend form");
    MNY_REQUIRE(!red || stats, "pw_thin: the BN-backward sums need their output");
    ThinArgs a{A, in_scale, in_shift, in_act, W, bias, addend, C, stats, M, N, t.nq, t.rpb, t.R, t.ntiles,
               rY, r_scale, r_shift, r_mean, r_invstd, r_act};
    const ThinKernel k = bf ? thin_pick_kr<bf16_t>(K, t.R, xf, red, stats != nullptr, addend != nullptr)
                            : thin_pick_kr<float>(K, t.R, xf, red, stats != nullptr, addend != nullptr);
    MNY_REQUIRE(k != nullptr, "pw_thin: K=%d", K);
    hipLaunchKernelGGL(k, dim3(t.grid), dim3(t.rpb * t.nq), 0, st, a);
    return check_launch("pw_thin_kernel");
}

}  // namespace mny
