// Depthwise KxK convolution (K = 3 | 5, stride 1 | 2), NHWC fp32 — forward, backward-data,
// backward-weight.  HBM-bandwidth-bound (AI 0.9-2.2 FLOP/B): the design goal is
// one coalesced pass over the input and one over the output.
//
// Thread = 4 consecutive channels (one float4) x one output column; it slides a KxK register
// window down a strip of TH output rows, so every input row is loaded once per strip (K-S halo
// rows re-read per strip come from L2).  Consecutive lanes cover consecutive channel groups and
// then consecutive output columns: every wave-level load/store is a run of full 16-B vectors over
// contiguous NHWC memory.  BatchNorm-apply + activation of the producer is fused into the load
// (`view` arguments), the BN statistics of the output are fused into the epilogue as deterministic
// per-block partial sums.
//
// replaces nn.Conv2d(groups=C) at models/mobilenetv2.py:65,79 and models/mbv2_yolo.py:22.
#include <stdlib.h>

#include "common.h"

namespace mny {

struct DwGeom {
    int N, H, W, C, Ho, Wo;
    int TH, nHS;        // strip height, strips per column
    int64_t nstrips;    // N * Wo * nHS
    int cg_total, cgb;  // channel groups total / per block
    int xcd;            // second-generation forward: XCD-contiguous strip order
    int nt;             // second-generation forward: non-temporal output stores
};

template <int KS>
__device__ __forceinline__ void load_weights(const float* __restrict__ w, int c, bool flip, float4 (&wg)[KS * KS]) {
    constexpr int KK = KS * KS;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        int s = flip ? (KK - 1 - t) : t;
        wg[t] = make_float4(w[(c + 0) * KK + s], w[(c + 1) * KK + s], w[(c + 2) * KK + s], w[(c + 3) * KK + s]);
    }
}

// MODE 0: forward (y = conv(view(x)), optional stats)    [also backward-data for stride 1 with flip]
// MODE 1: backward-weight (accumulate dy * view(x) per tap)
// XF: 0 = input used as is, 1 = scale/shift + min(max(z, slope*z), hi), 2 = scale/shift + hswish (compile-time: the
// inner loop must stay free of control flow)
template <typename T, int KS, int S, int MODE, int XF>
__global__ __launch_bounds__(256) void dw_slide_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act,
    const float* __restrict__ w, int flip, const T* __restrict__ addend,
    T* __restrict__ y,            // MODE 0: output; MODE 1: unused
    const T* __restrict__ dy,     // MODE 1: output gradient
    float* __restrict__ parts,        // MODE 0: stats [grid.x][2][C] (may be null); MODE 1: [grid.x][C*KK]
    DwGeom g) {
    constexpr int P = KS / 2;
    constexpr int KK = KS * KS;
    __shared__ float4 red[256 * 2];

    const int tid = threadIdx.x;
    const int cgl = tid % g.cgb;
    const int pix = tid / g.cgb;
    const int ppb = blockDim.x / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    const bool cvalid = cg < g.cg_total;
    const int c = cg * 4;

    float4 wg[KK];
    float4 sc = f4one(), sh = f4zero();
    const bool has_xf = in_scale != nullptr;
    if (cvalid) {
        if (MODE == 0) load_weights<KS>(w, c, flip != 0, wg);
        if (has_xf) { sc = ld4(in_scale + c); sh = ld4(in_shift + c); }
    }
    float4 acc_s1 = f4zero(), acc_s2 = f4zero();   // MODE 0 stats
    float4 wacc[KK];                               // MODE 1 accumulators
    if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < KK; ++t) wacc[t] = f4zero();
    }

    if (cvalid) {
        for (int64_t strip = (int64_t)blockIdx.x * ppb + pix; strip < g.nstrips; strip += (int64_t)gridDim.x * ppb) {
            const int wo = (int)(strip % g.Wo);
            const int hs = (int)((strip / g.Wo) % g.nHS);
            const int n = (int)(strip / ((int64_t)g.Wo * g.nHS));
            const int ho0 = hs * g.TH;
            const int ho1 = min(ho0 + g.TH, g.Ho);
            const T* xn = x + (int64_t)n * g.H * g.W * g.C + c;
            float4 win[KS][KS];
            const float slope = act_slope(in_act), hi_clip = act_hi(in_act);
            auto tap_ok = [&](int hi, int q) { const int wi = wo * S - P + q; return hi >= 0 && hi < g.H && wi >= 0 && wi < g.W; };
            // loads are UNCONDITIONAL (address clamped into the image) so the inner loop has no control flow and all
            // loads of a row are in flight together; out-of-image taps are zeroed by a select afterwards
            auto tap_ld = [&](int hi, int q) {
                const int wi = min(max(wo * S - P + q, 0), g.W - 1);
                const int hc = min(max(hi, 0), g.H - 1);
                return ld4(xn + ((int64_t)hc * g.W + wi) * g.C);
            };
            auto tap_fin = [&](float4 v, int hi, int q) {       // padding taps are 0 in the ACTIVATED domain
                if (XF != 0) {
                    float z[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        z[e] = XF == 1 ? fminf(fmaxf(z[e], slope * z[e]), hi_clip) : z[e] * fminf(fmaxf(z[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
                    v = make_float4(z[0], z[1], z[2], z[3]);
                }
                return tap_ok(hi, q) ? v : f4zero();
            };
#pragma unroll
            for (int r = 0; r < KS - S; ++r)                     // pre-shifted; the first iteration shifts them down
#pragma unroll
                for (int q = 0; q < KS; ++q) win[r + S][q] = tap_fin(tap_ld(ho0 * S - P + r, q), ho0 * S - P + r, q);
            for (int ho = ho0; ho < ho1; ++ho) {
                float4 cur[S][KS];
#pragma unroll
                for (int r = 0; r < S; ++r)
#pragma unroll
                    for (int q = 0; q < KS; ++q) cur[r][q] = tap_ld(ho * S - P + (KS - S) + r, q);
                float4 d = f4zero();
                if (MODE == 1) d = ld4(dy + (((int64_t)n * g.Ho + ho) * g.Wo + wo) * g.C + c);
#pragma unroll
                for (int r = 0; r < KS - S; ++r)
#pragma unroll
                    for (int q = 0; q < KS; ++q) win[r][q] = win[r + S][q];
#pragma unroll
                for (int r = 0; r < S; ++r)
#pragma unroll
                    for (int q = 0; q < KS; ++q) win[KS - S + r][q] = tap_fin(cur[r][q], ho * S - P + (KS - S) + r, q);
                const int64_t o = (((int64_t)n * g.Ho + ho) * g.Wo + wo) * g.C + c;
                if (MODE == 0) {
                    float4 out = addend ? ld4(addend + o) : f4zero();
#pragma unroll
                    for (int r = 0; r < KS; ++r)
#pragma unroll
                        for (int q = 0; q < KS; ++q) fma4(out, win[r][q], wg[r * KS + q]);
                    st4(y + o, out);
                    out = stored4<T>(out);                       // statistics over the values the consumer will read
                    add4(acc_s1, out);
                    fma4(acc_s2, out, out);
                } else {
#pragma unroll
                    for (int r = 0; r < KS; ++r)
#pragma unroll
                        for (int q = 0; q < KS; ++q) fma4(wacc[r * KS + q], win[r][q], d);
                }
            }
        }
    }

    if (parts == nullptr) return;
    // deterministic block reduction over the `ppb` pixel slots, fixed order
    if (MODE == 0) {
        red[tid * 2 + 0] = acc_s1;
        red[tid * 2 + 1] = acc_s2;
        __syncthreads();
        if (pix == 0 && cvalid) {
            float4 a = f4zero(), b = f4zero();
            for (int p = 0; p < ppb; ++p) { add4(a, red[(p * g.cgb + cgl) * 2]); add4(b, red[(p * g.cgb + cgl) * 2 + 1]); }
            float* dst = parts + (int64_t)blockIdx.x * 2 * g.C;
            st4(dst + c, a);
            st4(dst + g.C + c, b);
        }
    } else {
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            __syncthreads();
            red[tid] = wacc[t];
            __syncthreads();
            if (pix == 0 && cvalid) {
                float4 a = f4zero();
                for (int p = 0; p < ppb; ++p) add4(a, red[p * g.cgb + cgl]);
                float* dst = parts + (int64_t)blockIdx.x * g.C * KK;
                dst[(c + 0) * KK + t] = a.x; dst[(c + 1) * KK + t] = a.y;
                dst[(c + 2) * KK + t] = a.z; dst[(c + 3) * KK + t] = a.w;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 forward, second generation (stride 1 | 2; also the stride-1 backward-data as a correlation with the flipped filter).
//
// dw_slide_kernel's inner loop is 161 VALU instructions per output row of a wave (ISA count, fp32 3x3 s1 + ReLU6 view): every
// tap is transformed separately (fma + mul/max/min + a padding select per element), the window is shifted with 28 v_mov, every
// tap address is a 64-bit multiply.  At 4 cycles per wave64 VALU instruction that is 644 cycles per 2 KB of algorithmic
// traffic and SIMD = a 7.8 TB/s ceiling for the whole chip, i.e. the "HBM-bound" kernel ran at 55-60 % VALU utilisation with
// only 4 waves/SIMD to hide memory latency behind it.  This kernel keeps the thread mapping (thread = 4 channels x one output
// column sliding down a strip; lanes = channel groups, then columns: every wave-level access is a contiguous run of 16-B
// vectors) and removes the instructions:
//   * packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): a float4 is two register pairs, every fma covers 2 lanes' worth;
//   * ReLU6 as one v_med3_f32; the view transform is templated (none / ReLU6 / h-swish / generic clamp);
//   * no per-tap padding selects: the transformed row is multiplied by a 0/1 mask per column (row inside the image x column
//     inside the image: one packed multiply per register pair);
//   * the 3-row window rotates through a x3-unrolled loop by renaming instead of moves;
//   * three column pointers bumped by the row pitch instead of per-tap 64-bit index arithmetic.
// ~60 VALU per row.  Strips are handed to workgroups XCD-contiguously (workgroup i lands on XCD i % 8): the column / row halo
// a workgroup shares with its neighbours is then in the same 4-MB L2 instead of going out to the fabric again.

// A row can be REQUESTED one step before it is used (template parameter PF of the forward kernels): the request keeps the bytes as
// loaded (`raw`: 16 B fp32 / 8 B bf16 per lane) and the widening shifts of bf16 storage run only when the row is consumed, so nothing
// in the requesting step waits for it.  Why: a thread keeps one row of loads (3 or 5 x 16 B / 8 B) in flight and 3 workgroups per CU
// are resident — with 8-B lanes that is ~4.7 MB in flight over the chip, i.e. ~2.3 TB/s at the ~2 us a dependent row takes on the
// short strips of the 16^2 / 32^2 / 64^2 layers (MobileNetV3 512^2 bs 64 measured 1.1-2.6 TB/s); two rows in flight double it.
template <typename T> struct dw_raw { typedef float4 type; };
template <> struct dw_raw<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 dw_ldraw(const float* p) { return ld4(p); }
__device__ __forceinline__ uint2 dw_ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 dw_widen(float4 v) { return v; }
__device__ __forceinline__ float4 dw_widen(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// XF: 0 none (value used as is), 1 ReLU6, 2 h-swish, 4 max(z, slope*z) = leaky / relu / scale-only (no upper clip)
template <int XF>
__device__ __forceinline__ F4P dw_xf(float4 v, v2f sc_lo, v2f sc_hi, v2f sh_lo, v2f sh_hi, float slope, float hi_clip) {
    F4P z;
    if (XF == 0) { z.lo = v2f{v.x, v.y}; z.hi = v2f{v.z, v.w}; return z; }
    z.lo = __builtin_elementwise_fma(v2f{v.x, v.y}, sc_lo, sh_lo);
    z.hi = __builtin_elementwise_fma(v2f{v.z, v.w}, sc_hi, sh_hi);
    if (XF == 1) {
        z.lo = v2f{__builtin_amdgcn_fmed3f(z.lo.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.lo.y, 0.f, 6.f)};
        z.hi = v2f{__builtin_amdgcn_fmed3f(z.hi.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.hi.y, 0.f, 6.f)};
    } else if (XF == 2) {
        const v2f three = v2f{3.f, 3.f}, sixth = v2f{1.f / 6.f, 1.f / 6.f};
        v2f a = z.lo + three, b = z.hi + three;
        a = v2f{__builtin_amdgcn_fmed3f(a.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(a.y, 0.f, 6.f)};
        b = v2f{__builtin_amdgcn_fmed3f(b.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(b.y, 0.f, 6.f)};
        z.lo = z.lo * a * sixth;                      // same operation order as act_fwd: z * clamp(z + 3) / 6 -> (z * h) * (1/6) differs by <= 1 ulp
        z.hi = z.hi * b * sixth;
    } else if (XF == 4) {                                // leaky / relu / scale-only: no upper clip
        const v2f sl = v2f{slope, slope};
        const v2f a = z.lo * sl, b = z.hi * sl;
        z.lo = v2f{fmaxf(z.lo.x, a.x), fmaxf(z.lo.y, a.y)};
        z.hi = v2f{fmaxf(z.hi.x, b.x), fmaxf(z.hi.y, b.y)};
    }
    return z;
}

template <typename T, int S, int XF, bool ADD, bool NT, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void dw3_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act,
    const float* __restrict__ w, int flip, const T* __restrict__ addend, T* __restrict__ y, float* __restrict__ parts, DwGeom g) {
    __shared__ float4 red[256 * 2];
    const int tid = threadIdx.x;
    const int cgl = tid % g.cgb;
    const int pix = tid / g.cgb;
    const int ppb = blockDim.x / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    const bool cvalid = cg < g.cg_total;
    const int c = cg * 4;
    // XCD-contiguous strip order: hardware places workgroup b on XCD b % 8 (gridDim.x is a multiple of 8 whenever it exceeds 8)
    const int gx = gridDim.x;
    const int lb = (g.xcd && (gx & 7) == 0) ? (int)(blockIdx.x & 7) * (gx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    F4P acc1 = f4p0(), acc2 = f4p0();
    if (cvalid) {
        v2f sc_lo = v2f{1.f, 1.f}, sc_hi = sc_lo, sh_lo = v2f{0.f, 0.f}, sh_hi = sh_lo;
        if (in_scale != nullptr) {
            const float4 a = ld4(in_scale + c), b = ld4(in_shift + c);
            sc_lo = v2f{a.x, a.y}; sc_hi = v2f{a.z, a.w}; sh_lo = v2f{b.x, b.y}; sh_hi = v2f{b.z, b.w};
        }
        const float slope = act_slope(in_act), hi_clip = act_hi(in_act);
        // filter taps of this thread's 4 channels (36 contiguous floats), loaded once per thread
        F4P wt[9];
        {
            float raw[36];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float4 v = ld4(w + (int64_t)c * 9 + 4 * i);
                raw[4 * i] = v.x; raw[4 * i + 1] = v.y; raw[4 * i + 2] = v.z; raw[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                wt[t].lo = v2f{flip ? raw[8 - t] : raw[t], flip ? raw[9 + 8 - t] : raw[9 + t]};
                wt[t].hi = v2f{flip ? raw[18 + 8 - t] : raw[18 + t], flip ? raw[27 + 8 - t] : raw[27 + t]};
            }
        }
        const int64_t pitch = (int64_t)g.W * g.C;               // input row pitch (elements)
        const int64_t opitch = (int64_t)g.Wo * g.C;
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < g.nstrips; strip += (int64_t)gx * ppb) {
            const int wo = (int)(strip % g.Wo);
            const int hs = (int)((strip / g.Wo) % g.nHS);
            const int n = (int)(strip / ((int64_t)g.Wo * g.nHS));
            const int ho0 = hs * g.TH;
            const int ho1 = min(ho0 + g.TH, g.Ho);
            const float ml = (wo * S - 1 >= 0) ? 1.f : 0.f, mr = (wo * S + 1 < g.W) ? 1.f : 0.f;     // column taps outside the image
            const int wl = max(wo * S - 1, 0), wm = wo * S, wr = min(wo * S + 1, g.W - 1);
            const T* xn = x + (int64_t)n * g.H * pitch + c;
            // row pointers of the three columns; `hi_next` = input row the next load fetches
            int hi_next = ho0 * S - 1;
            const T* pl = xn + (int64_t)max(hi_next, 0) * pitch + (int64_t)wl * g.C;
            const T* pm = xn + (int64_t)max(hi_next, 0) * pitch + (int64_t)wm * g.C;
            const T* pr = xn + (int64_t)max(hi_next, 0) * pitch + (int64_t)wr * g.C;
            T* yo = y + (((int64_t)n * g.Ho + ho0) * g.Wo + wo) * g.C + c;
            const T* ad = ADD ? addend + (((int64_t)n * g.Ho + ho0) * g.Wo + wo) * g.C + c : nullptr;

            // load + transform one input row into r[3]; rows outside the image come out as zeros
            // fetch = the three 16-B loads of one input row (+ pointer bump); finish = view transform, zero if the row is outside
            typedef typename dw_raw<T>::type RW;
            int nfl = (ho1 - ho0) * S + (3 - S);                 // input rows this strip still has to request: TH + 2 (stride 1), 2 TH + 1 (stride 2)
            auto fetch = [&](RW (&raw)[3], float& m) {
                m = (hi_next >= 0 && hi_next < g.H) ? 1.f : 0.f;
                raw[0] = dw_ldraw(pl); raw[1] = dw_ldraw(pm); raw[2] = dw_ldraw(pr);
                // advance unless the next row would leave the image (the clamped re-read is zeroed by its own `m`) or, PF = 1, the strip
                // needs no further row: the look-ahead request past the strip's end then re-reads the line it just loaded (a cache hit that
                // is never consumed) — a branch around it would put a wait for the newest row in front of the branch (measured in the ISA)
                --nfl;
                const int64_t step = (hi_next >= 0 && hi_next + 1 < g.H && (PF == 0 || nfl > 0)) ? pitch : 0;
                pl += step; pm += step; pr += step;
                ++hi_next;
            };
            auto finish = [&](const RW (&raw)[3], float m, F4P (&r)[3]) {
                const float mq[3] = {m * ml, m, m * mr};          // row outside the image / column outside the image -> zeros
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const v2f m2 = v2f{mq[q], mq[q]};
                    r[q] = dw_xf<XF>(dw_widen(raw[q]), sc_lo, sc_hi, sh_lo, sh_hi, slope, hi_clip);
                    r[q].lo *= m2; r[q].hi *= m2;
                }
            };
            // PF = 1: the row(s) of the NEXT step are requested before this step's row is transformed (pa / pb hold them raw)
            RW pa[3], pb[3];
            float pma = 0.f, pmb = 0.f;
            auto load_row = [&](F4P (&r)[3]) {
                if constexpr (PF == 0) { RW raw[3]; float m; fetch(raw, m); finish(raw, m, r); }
                else {
                    RW c[3] = {pa[0], pa[1], pa[2]};
                    const float cm = pma;
                    fetch(pa, pma);
                    finish(c, cm, r);
                }
            };
            // stride 2: both rows' loads are issued before either is transformed (6 x 16 B in flight per thread)
            auto load_rows2 = [&](F4P (&ra)[3], F4P (&rb)[3]) {
                if constexpr (PF == 0) {
                    RW wa[3], wb[3]; float ma, mb;
                    fetch(wa, ma); fetch(wb, mb);
                    __builtin_amdgcn_sched_barrier(0);
                    finish(wa, ma, ra); finish(wb, mb, rb);
                } else {
                    RW ca[3] = {pa[0], pa[1], pa[2]}, cb[3] = {pb[0], pb[1], pb[2]};
                    const float cma = pma, cmb = pmb;
                    fetch(pa, pma); fetch(pb, pmb);
                    __builtin_amdgcn_sched_barrier(0);
                    finish(ca, cma, ra); finish(cb, cmb, rb);
                }
            };
            auto emit = [&](const F4P (&top)[3], const F4P (&mid)[3], const F4P (&bot)[3]) {
                F4P o = f4p0();
                if (ADD) { o = f4p(ld4(ad)); ad += opitch; }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    o.lo = __builtin_elementwise_fma(top[q].lo, wt[q].lo, o.lo);         o.hi = __builtin_elementwise_fma(top[q].hi, wt[q].hi, o.hi);
                    o.lo = __builtin_elementwise_fma(mid[q].lo, wt[3 + q].lo, o.lo);     o.hi = __builtin_elementwise_fma(mid[q].hi, wt[3 + q].hi, o.hi);
                    o.lo = __builtin_elementwise_fma(bot[q].lo, wt[6 + q].lo, o.lo);     o.hi = __builtin_elementwise_fma(bot[q].hi, wt[6 + q].hi, o.hi);
                }
                const float4 of = make_float4(o.lo.x, o.lo.y, o.hi.x, o.hi.y);
                if (NT) st4_stream(yo, of); else st4(yo, of);
                yo += opitch;
                const F4P os = f4p(stored4<T>(of));               // statistics over the values the consumer will read
                acc1.lo += os.lo; acc1.hi += os.hi;
                acc2.lo = __builtin_elementwise_fma(os.lo, os.lo, acc2.lo);
                acc2.hi = __builtin_elementwise_fma(os.hi, os.hi, acc2.hi);
                __builtin_amdgcn_sched_barrier(0);               // keep the three unrolled steps apart: interleaving them costs 100+ VGPRs
            };
            F4P r0[3], r1[3], r2[3];
            int left = ho1 - ho0;
            if constexpr (PF != 0) {
                fetch(pa, pma);               // the strip's first row; every load_row / load_rows2 below requests the one(s) after its own
                if (S == 2) {                 // (stride 2: the single leading row is consumed alone, then pairs)
                    RW c[3] = {pa[0], pa[1], pa[2]};
                    const float cm = pma;
                    fetch(pa, pma); fetch(pb, pmb);
                    finish(c, cm, r0);
                }
            }
            if (S == 1) {
                load_row(r0);                 // row ho0-1
                load_row(r1);                 // row ho0
                // each step: load the row below, emit, rotate by renaming
#pragma clang loop unroll(disable)
                for (; left >= 3; left -= 3) {
                    load_row(r2); emit(r0, r1, r2);
                    load_row(r0); emit(r1, r2, r0);
                    load_row(r1); emit(r2, r0, r1);
                }
                if (left >= 1) { load_row(r2); emit(r0, r1, r2); }
                if (left >= 2) { load_row(r0); emit(r1, r2, r0); }
            } else {
                if constexpr (PF == 0) load_row(r0);                 // row 2*ho0-1
#pragma clang loop unroll(disable)
                for (; left >= 3; left -= 3) {
                    load_rows2(r1, r2); emit(r0, r1, r2);
                    load_rows2(r0, r1); emit(r2, r0, r1);
                    load_rows2(r2, r0); emit(r1, r2, r0);
                }
                if (left >= 1) { load_rows2(r1, r2); emit(r0, r1, r2); }
                if (left >= 2) { load_rows2(r0, r1); emit(r2, r0, r1); }
            }
        }
    }
    if (parts == nullptr) return;
    red[tid * 2 + 0] = make_float4(acc1.lo.x, acc1.lo.y, acc1.hi.x, acc1.hi.y);
    red[tid * 2 + 1] = make_float4(acc2.lo.x, acc2.lo.y, acc2.hi.x, acc2.hi.y);
    __syncthreads();
    if (pix == 0 && cvalid) {
        float4 a = f4zero(), b = f4zero();
        for (int p = 0; p < ppb; ++p) { add4(a, red[(p * g.cgb + cgl) * 2]); add4(b, red[(p * g.cgb + cgl) * 2 + 1]); }
        float* dst = parts + (int64_t)blockIdx.x * 2 * g.C;
        st4(dst + c, a);
        st4(dst + g.C + c, b);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 5x5 forward (stride 1 | 2) and stride-1 backward-data (flipped filter) — MobileNetV3's depthwise layers.
//
// The sliding-window kernel keeps a 5x5 register window (100 VGPRs) next to the 25 taps (100 VGPRs): 250-266 VGPRs, two waves
// per SIMD with half of them spilling, 0.6-1.5 TB/s on the bf16 configuration.  Here the window is gone: every INPUT row (five
// columns, transformed once) is scattered into the partial sums of the output rows it belongs to — stride 1: rows hi-2..hi+2
// (five accumulators), stride 2: an even row feeds three outputs (taps 4, 2, 0), an odd row two (taps 3, 1) — and an output row
// is stored as soon as its last input row has passed.  Per thread (4 channels x one output column): 25 taps + 3-5 accumulators
// + one row, ~170 VGPRs, no spills, two resident workgroups per CU (512 blocks).  Same lane mapping, XCD-contiguous strip order,
// masks and streaming stores as dw3_fwd_kernel.
template <typename T, int S, int XF, bool ADD, bool NT, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void dw5_fwd_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act,
    const float* __restrict__ w, int flip, const T* __restrict__ addend, T* __restrict__ y, float* __restrict__ parts, DwGeom g) {
    __shared__ float4 red[256 * 2];
    const int tid = threadIdx.x;
    const int cgl = tid % g.cgb;
    const int pix = tid / g.cgb;
    const int ppb = blockDim.x / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    const bool cvalid = cg < g.cg_total;
    const int c = cg * 4;
    const int gx = gridDim.x;
    const int lb = (g.xcd && (gx & 7) == 0) ? (int)(blockIdx.x & 7) * (gx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;

    F4P acc1 = f4p0(), acc2 = f4p0();
    if (cvalid) {
        v2f sc_lo = v2f{1.f, 1.f}, sc_hi = sc_lo, sh_lo = v2f{0.f, 0.f}, sh_hi = sh_lo;
        if (in_scale != nullptr) {
            const float4 a = ld4(in_scale + c), b = ld4(in_shift + c);
            sc_lo = v2f{a.x, a.y}; sc_hi = v2f{a.z, a.w}; sh_lo = v2f{b.x, b.y}; sh_hi = v2f{b.z, b.w};
        }
        const float slope = act_slope(in_act), hi_clip = act_hi(in_act);
        F4P wt[25];                                                // taps of this thread's 4 channels (100 contiguous floats)
        {
            const float* wp = w + (int64_t)c * 25;
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                const int s = flip ? 24 - t : t;
                wt[t].lo = v2f{wp[s], wp[25 + s]};
                wt[t].hi = v2f{wp[50 + s], wp[75 + s]};
            }
        }
        const int64_t pitch = (int64_t)g.W * g.C, opitch = (int64_t)g.Wo * g.C;
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < g.nstrips; strip += (int64_t)gx * ppb) {
            const int wo = (int)(strip % g.Wo);
            const int hs = (int)((strip / g.Wo) % g.nHS);
            const int n = (int)(strip / ((int64_t)g.Wo * g.nHS));
            const int ho0 = hs * g.TH;
            const int ho1 = min(ho0 + g.TH, g.Ho);
            int coff[5];
            float cm[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int wi = wo * S - 2 + q;
                cm[q] = (wi >= 0 && wi < g.W) ? 1.f : 0.f;
                coff[q] = min(max(wi, 0), g.W - 1) * g.C;
            }
            const T* xn = x + (int64_t)n * g.H * pitch + c;
            T* yo = y + (((int64_t)n * g.Ho + ho0) * g.Wo + wo) * g.C + c;
            const T* ad = ADD ? addend + (((int64_t)n * g.Ho + ho0) * g.Wo + wo) * g.C + c : nullptr;

            typedef typename dw_raw<T>::type RW;
            auto request = [&](int hi, RW (&raw)[5], float& rm) {  // the five 16-B / 8-B loads of input row hi (clamped into the image)
                rm = (hi >= 0 && hi < g.H) ? 1.f : 0.f;
                const T* p = xn + (int64_t)min(max(hi, 0), g.H - 1) * pitch;
#pragma unroll
                for (int q = 0; q < 5; ++q) raw[q] = dw_ldraw(p + coff[q]);
            };
            auto finish = [&](const RW (&raw)[5], float rm, F4P (&r)[5]) {     // transformed; zeros outside the image
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float m = rm * cm[q];
                    const v2f m2 = v2f{m, m};
                    r[q] = dw_xf<XF>(dw_widen(raw[q]), sc_lo, sc_hi, sh_lo, sh_hi, slope, hi_clip);
                    r[q].lo *= m2; r[q].hi *= m2;
                }
            };
            // PF = 1: `pend` holds the raw row requested one step ahead; load_row(hi, next, r) consumes it and requests row `next` (the last
            // step names its own row again: a cache hit that is never consumed, see dw3_fwd_kernel)
            RW pend[5];
            float pend_m = 0.f;
            auto load_row = [&](int hi, int next, F4P (&r)[5]) {
                if constexpr (PF == 0) { RW raw[5]; float rm; request(hi, raw, rm); finish(raw, rm, r); }
                else {
                    RW c[5] = {pend[0], pend[1], pend[2], pend[3], pend[4]};
                    const float cmk = pend_m;
                    request(next, pend, pend_m);
                    finish(c, cmk, r);
                }
            };
            auto tap_row = [&](F4P& acc, const F4P (&r)[5], int kr) {
#pragma unroll
                for (int q = 0; q < 5; ++q) pfma(acc, r[q], wt[kr * 5 + q]);
            };
            auto emit = [&](F4P o) {
                if (ADD) { const F4P a = f4p(ld4(ad)); o.lo += a.lo; o.hi += a.hi; ad += opitch; }
                const float4 of = f4u(o);
                if (NT) st4_stream(yo, of); else st4(yo, of);
                yo += opitch;
                const F4P os = f4p(stored4<T>(of));
                acc1.lo += os.lo; acc1.hi += os.hi;
                acc2.lo = __builtin_elementwise_fma(os.lo, os.lo, acc2.lo);
                acc2.hi = __builtin_elementwise_fma(os.hi, os.hi, acc2.hi);
                __builtin_amdgcn_sched_barrier(0);
            };
            F4P r[5];
            if (S == 1) {
                // accumulator j holds output row (hi - 2 + j) while input row hi is being scattered: kr = 4 - j
                F4P P0 = f4p0(), P1 = f4p0(), P2 = f4p0(), P3 = f4p0(), P4 = f4p0();
                if constexpr (PF != 0) request(ho0 - 2, pend, pend_m);
                for (int hi = ho0 - 2; hi <= ho1 + 1; ++hi) {
                    load_row(hi, min(hi + 1, ho1 + 1), r);
                    tap_row(P0, r, 4); tap_row(P1, r, 3); tap_row(P2, r, 2); tap_row(P3, r, 1); tap_row(P4, r, 0);
                    if (hi - 2 >= ho0) emit(P0);                    // output row hi-2 has seen its last input row (hi-2 < ho1 by the loop bound)
                    P0 = P1; P1 = P2; P2 = P3; P3 = P4; P4 = f4p0();
                }
            } else {
                // output row t collects input rows 2t-2 .. 2t+2: an even row 2t feeds outputs t-1, t, t+1 with taps 4, 2, 0, an odd
                // row 2t+1 feeds t, t+1 with taps 3, 1; output t-1 is complete after row 2t
                F4P P0 = f4p0(), P1 = f4p0(), P2 = f4p0();
                if constexpr (PF != 0) request(2 * (ho0 - 1), pend, pend_m);
                for (int t = ho0 - 1; t <= ho1; ++t) {
                    load_row(2 * t, 2 * t + 1, r);
                    tap_row(P0, r, 4); tap_row(P1, r, 2); tap_row(P2, r, 0);
                    if (t - 1 >= ho0) emit(P0);                     // (t-1 < ho1 by the loop bound)
                    load_row(2 * t + 1, min(2 * t + 2, 2 * ho1 + 1), r);
                    tap_row(P1, r, 3); tap_row(P2, r, 1);
                    P0 = P1; P1 = P2; P2 = f4p0();
                }
            }
        }
    }
    if (parts == nullptr) return;
    red[tid * 2 + 0] = f4u(acc1);
    red[tid * 2 + 1] = f4u(acc2);
    __syncthreads();
    if (pix == 0 && cvalid) {
        float4 a = f4zero(), b = f4zero();
        for (int p = 0; p < ppb; ++p) { add4(a, red[(p * g.cgb + cgl) * 2]); add4(b, red[(p * g.cgb + cgl) * 2 + 1]); }
        float* dst = parts + (int64_t)blockIdx.x * 2 * g.C;
        st4(dst + c, a);
        st4(dst + g.C + c, b);
    }
}

// 5x5 weight gradient, the dual of dw5_fwd_kernel: dW[kr][q] = sum over outputs of dY[ho][wo] * a[S*ho - 2 + kr][S*wo - 2 + q].
// The thread (4 channels x one output column) walks the INPUT rows of its strip once; input row hi meets the centre-column dY
// of the output rows it belongs to (stride 1: hi-2..hi+2, kept as a 5-entry history; stride 2: t-1, t, t+1 for an even row 2t,
// t, t+1 for the odd one) — 25 accumulators + one input row + the dY history, no 5x5 window (the sliding-window kernel needed
// 250+ VGPRs and spilled).  A strip owns its OUTPUT rows: dY of other strips' rows enters as zero.
template <typename T, int S, int XF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void dw5_wgrad_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act,
    const T* __restrict__ dy, float* __restrict__ parts, DwGeom g) {
    __shared__ float4 red[256];
    const int tid = threadIdx.x;
    const int cgl = tid % g.cgb;
    const int pix = tid / g.cgb;
    const int ppb = blockDim.x / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    const bool cvalid = cg < g.cg_total;
    const int c = cg * 4;
    const int gx = gridDim.x;
    const int lb = (g.xcd && (gx & 7) == 0) ? (int)(blockIdx.x & 7) * (gx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    F4P acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f4p0();
    if (cvalid) {
        v2f sc_lo = v2f{1.f, 1.f}, sc_hi = sc_lo, sh_lo = v2f{0.f, 0.f}, sh_hi = sh_lo;
        if (in_scale != nullptr) {
            const float4 a = ld4(in_scale + c), b = ld4(in_shift + c);
            sc_lo = v2f{a.x, a.y}; sc_hi = v2f{a.z, a.w}; sh_lo = v2f{b.x, b.y}; sh_hi = v2f{b.z, b.w};
        }
        const float slope = act_slope(in_act), hi_clip = act_hi(in_act);
        const int64_t pitch = (int64_t)g.W * g.C, opitch = (int64_t)g.Wo * g.C;
        for (int64_t strip = (int64_t)lb * ppb + pix; strip < g.nstrips; strip += (int64_t)gx * ppb) {
            const int wo = (int)(strip % g.Wo);
            const int hs = (int)((strip / g.Wo) % g.nHS);
            const int n = (int)(strip / ((int64_t)g.Wo * g.nHS));
            const int ho0 = hs * g.TH;
            const int ho1 = min(ho0 + g.TH, g.Ho);
            int coff[5];
            float cm[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int wi = wo * S - 2 + q;
                cm[q] = (wi >= 0 && wi < g.W) ? 1.f : 0.f;
                coff[q] = min(max(wi, 0), g.W - 1) * g.C;
            }
            const T* xn = x + (int64_t)n * g.H * pitch + c;
            const T* dn = dy + ((int64_t)n * g.Ho * g.Wo + wo) * g.C + c;
            auto load_row = [&](int hi, F4P (&r)[5]) {
                const float rm = (hi >= 0 && hi < g.H) ? 1.f : 0.f;
                const T* p = xn + (int64_t)min(max(hi, 0), g.H - 1) * pitch;
                float4 raw[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) raw[q] = ld4(p + coff[q]);
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float m = rm * cm[q];
                    const v2f m2 = v2f{m, m};
                    r[q] = dw_xf<XF>(raw[q], sc_lo, sc_hi, sh_lo, sh_hi, slope, hi_clip);
                    r[q].lo *= m2; r[q].hi *= m2;
                }
            };
            auto load_dy = [&](int ho) {                            // centre-column dY of output row ho; zero outside this strip's rows
                const float m = (ho >= ho0 && ho < ho1) ? 1.f : 0.f;
                F4P d = f4p(ld4(dn + (int64_t)min(max(ho, 0), g.Ho - 1) * opitch));
                d.lo *= v2f{m, m}; d.hi *= v2f{m, m};
                return d;
            };
            auto tap_row = [&](const F4P (&r)[5], F4P d, int kr) {
#pragma unroll
                for (int q = 0; q < 5; ++q) pfma(acc[kr * 5 + q], r[q], d);
            };
            F4P r[5];
            if (S == 1) {
                // d[j] = dY of output row hi - 2 + j while input row hi is processed (kr = 4 - j)
                F4P d0 = f4p0(), d1 = f4p0(), d2 = load_dy(ho0 - 2), d3 = load_dy(ho0 - 1), d4 = load_dy(ho0);
                for (int hi = ho0 - 2; hi <= ho1 + 1; ++hi) {
                    load_row(hi, r);
                    tap_row(r, d0, 4); tap_row(r, d1, 3); tap_row(r, d2, 2); tap_row(r, d3, 1); tap_row(r, d4, 0);
                    d0 = d1; d1 = d2; d2 = d3; d3 = d4; d4 = load_dy(hi + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                F4P dm = f4p0(), dc = load_dy(ho0 - 1), dp = load_dy(ho0);      // outputs t-1, t, t+1 for t = ho0 - 1
                for (int t = ho0 - 1; t <= ho1; ++t) {
                    load_row(2 * t, r);
                    tap_row(r, dm, 4); tap_row(r, dc, 2); tap_row(r, dp, 0);
                    load_row(2 * t + 1, r);
                    tap_row(r, dc, 3); tap_row(r, dp, 1);
                    dm = dc; dc = dp; dp = load_dy(t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // deterministic block reduction over the pixel slots, one tap at a time
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        __syncthreads();
        red[tid] = f4u(acc[t]);
        __syncthreads();
        if (pix == 0 && cvalid) {
            float4 a = f4zero();
            for (int p = 0; p < ppb; ++p) add4(a, red[p * g.cgb + cgl]);
            float* dst = parts + (int64_t)blockIdx.x * g.C * 25;
            dst[(c + 0) * 25 + t] = a.x; dst[(c + 1) * 25 + t] = a.y;
            dst[(c + 2) * 25 + t] = a.z; dst[(c + 3) * 25 + t] = a.w;
        }
    }
}

// stride-2 backward-data as a gather over the (at most ceil(K/2)^2) contributing taps
template <typename T, int KS>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                             const T* __restrict__ addend, T* __restrict__ dx,
                                                             int N, int H, int W, int C, int Ho, int Wo, int cgb, int cg_total) {
    constexpr int P = KS / 2;
    constexpr int KK = KS * KS;
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    float4 wg[KK];
    load_weights<KS>(w, c, false, wg);
    const int64_t npix = (int64_t)N * H * W;
    for (int64_t p = (int64_t)blockIdx.x * ppb + pix; p < npix; p += (int64_t)gridDim.x * ppb) {
        const int wi = (int)(p % W);
        const int hi = (int)((p / W) % H);
        const int n = (int)(p / ((int64_t)W * H));
        float4 out = addend ? ld4(addend + p * C + c) : f4zero();
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int th = hi + P - kh;
            if (th < 0 || (th & 1)) continue;
            const int ho = th >> 1;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int tw = wi + P - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int wo = tw >> 1;
                if (wo >= Wo) continue;
                fma4(out, ld4(dy + (((int64_t)n * Ho + ho) * Wo + wo) * C + c), wg[kh * KS + kw]);
            }
        }
        st4(dx + p * C + c, out);
    }
}

// 3x3 stride-2 backward-data, branch-free: a thread owns the 2x2 input quad (2i..2i+1, 2j..2j+1) of 4 channels.  With
// pad 1 the quad depends on exactly dy[i..i+1][j..j+1], each with a statically known filter tap:
//   dx[2i  ][2j  ] = dy[i][j]*w11
//   dx[2i  ][2j+1] = dy[i][j]*w12 + dy[i][j+1]*w10
//   dx[2i+1][2j  ] = dy[i][j]*w21 + dy[i+1][j]*w01
//   dx[2i+1][2j+1] = dy[i][j]*w22 + dy[i][j+1]*w20 + dy[i+1][j]*w02 + dy[i+1][j+1]*w00
// so the 4 loads are unconditional (clamped address, zeroed by select) and there is no control flow.
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_s2k3_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                               const T* __restrict__ addend, T* __restrict__ dx,
                                                               int N, int H, int W, int C, int Ho, int Wo, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    float4 wg[9];
    load_weights<3>(w, c, false, wg);
    const int Hq = (H + 1) / 2, Wq = (W + 1) / 2;
    const int64_t nq = (int64_t)N * Hq * Wq;
    for (int64_t q = (int64_t)blockIdx.x * ppb + pix; q < nq; q += (int64_t)gridDim.x * ppb) {
        const int j = (int)(q % Wq), i = (int)((q / Wq) % Hq);
        const int64_t n = q / ((int64_t)Wq * Hq);
        const T* dn = dy + n * Ho * Wo * C + c;
        const int i1 = min(i + 1, Ho - 1), j1 = min(j + 1, Wo - 1), i0 = min(i, Ho - 1), j0 = min(j, Wo - 1);
        float4 d00 = ld4(dn + ((int64_t)i0 * Wo + j0) * C), d01 = ld4(dn + ((int64_t)i0 * Wo + j1) * C);
        float4 d10 = ld4(dn + ((int64_t)i1 * Wo + j0) * C), d11 = ld4(dn + ((int64_t)i1 * Wo + j1) * C);
        const bool vi0 = i < Ho, vj0 = j < Wo, vi1 = i + 1 < Ho, vj1 = j + 1 < Wo;
        if (!(vi0 && vj0)) d00 = f4zero();
        if (!(vi0 && vj1)) d01 = f4zero();
        if (!(vi1 && vj0)) d10 = f4zero();
        if (!(vi1 && vj1)) d11 = f4zero();
        float4 o00 = f4zero(), o01 = f4zero(), o10 = f4zero(), o11 = f4zero();
        fma4(o00, d00, wg[4]);
        fma4(o01, d00, wg[5]); fma4(o01, d01, wg[3]);
        fma4(o10, d00, wg[7]); fma4(o10, d10, wg[1]);
        fma4(o11, d00, wg[8]); fma4(o11, d01, wg[6]); fma4(o11, d10, wg[2]); fma4(o11, d11, wg[0]);
        const int h0 = 2 * i, w0 = 2 * j;
        const int64_t base = ((n * H + h0) * W + w0) * C + c;
        const bool hv = h0 + 1 < H, wv2 = w0 + 1 < W;
        if (addend) {
            add4(o00, ld4(addend + base));
            if (wv2) add4(o01, ld4(addend + base + C));
            if (hv) add4(o10, ld4(addend + base + (int64_t)W * C));
            if (hv && wv2) add4(o11, ld4(addend + base + (int64_t)W * C + C));
        }
        st4_stream(dx + base, o00);
        if (wv2) st4_stream(dx + base + C, o01);
        if (hv) st4_stream(dx + base + (int64_t)W * C, o10);
        if (hv && wv2) st4_stream(dx + base + (int64_t)W * C + C, o11);
    }
}

// 5x5 stride-2 backward-data, the same idea: a thread owns the 2x2 input quad (2i..2i+1, 2j..2j+1) of 4 channels.  With pad 2,
// dx[hi][wi] collects dy[ho][wo] * w[hi+2-2ho][wi+2-2wo]; even rows/columns take taps 4,2,0 from dy[i-1..i+1], odd ones taps 3,1 from
// dy[i..i+1] -> 9 unconditional loads (clamped, zeroed by select), 25 statically chosen taps, no control flow.  (The generic
// gather kernel above tested parity and bounds per tap: 0.4 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void dw_bwd_data_s2k5_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                               const T* __restrict__ addend, T* __restrict__ dx,
                                                               int N, int H, int W, int C, int Ho, int Wo, int cgb, int cg_total) {
    const int cgl = threadIdx.x % cgb, pix = threadIdx.x / cgb, ppb = blockDim.x / cgb;
    const int cg = blockIdx.y * cgb + cgl;
    if (cg >= cg_total) return;
    const int c = cg * 4;
    float4 wg[25];
    load_weights<5>(w, c, false, wg);
    const int Hq = (H + 1) / 2, Wq = (W + 1) / 2;
    const int64_t nq = (int64_t)N * Hq * Wq;
    for (int64_t q = (int64_t)blockIdx.x * ppb + pix; q < nq; q += (int64_t)gridDim.x * ppb) {
        const int j = (int)(q % Wq), i = (int)((q / Wq) % Hq);
        const int64_t n = q / ((int64_t)Wq * Hq);
        const T* dn = dy + n * Ho * Wo * C + c;
        float4 d[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int r = i + a - 1, q2 = j + b - 1;
                const float4 v = ld4(dn + ((int64_t)min(max(r, 0), Ho - 1) * Wo + min(max(q2, 0), Wo - 1)) * C);
                d[a][b] = (r >= 0 && r < Ho && q2 >= 0 && q2 < Wo) ? v : f4zero();
            }
        float4 o00 = f4zero(), o01 = f4zero(), o10 = f4zero(), o11 = f4zero();
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                fma4(o00, d[a][b], wg[(4 - 2 * a) * 5 + (4 - 2 * b)]);                        // even row, even column: taps 4,2,0
                if (b >= 1) fma4(o01, d[a][b], wg[(4 - 2 * a) * 5 + (5 - 2 * b)]);            // odd column: taps 3,1 from dy[j], dy[j+1]
                if (a >= 1) fma4(o10, d[a][b], wg[(5 - 2 * a) * 5 + (4 - 2 * b)]);
                if (a >= 1 && b >= 1) fma4(o11, d[a][b], wg[(5 - 2 * a) * 5 + (5 - 2 * b)]);
            }
        const int h0 = 2 * i, w0 = 2 * j;
        const int64_t base = ((n * H + h0) * W + w0) * C + c;
        const bool hv = h0 + 1 < H, wv2 = w0 + 1 < W;
        if (addend) {
            add4(o00, ld4(addend + base));
            if (wv2) add4(o01, ld4(addend + base + C));
            if (hv) add4(o10, ld4(addend + base + (int64_t)W * C));
            if (hv && wv2) add4(o11, ld4(addend + base + (int64_t)W * C + C));
        }
        st4_stream(dx + base, o00);
        if (wv2) st4_stream(dx + base + C, o01);
        if (hv) st4_stream(dx + base + (int64_t)W * C, o10);
        if (hv && wv2) st4_stream(dx + base + (int64_t)W * C + C, o11);
    }
}

static bool dw_use_v2(int K, int mode) {
    static const bool v1 = getenv("MNY_DW_V1") != nullptr;       // A/B: the first-generation sliding-window kernel
    return ((K == 3 || K == 5) && mode == 0 && !v1) || (K == 5 && mode == 1 && !v1);      // 5x5 weight gradient: dw5_wgrad_kernel
}

static int dw_geom(DwGeom& g, CgLayout& L, int& gx, int N, int H, int W, int C, int K, int stride, int mode = 0) {
    MNY_REQUIRE(K == 3 || K == 5, "dw: kernel size %d unsupported (3 or 5)", K);
    MNY_REQUIRE(stride == 1 || stride == 2, "dw: stride %d unsupported", stride);
    MNY_REQUIRE(C % 4 == 0 && C > 0, "dw: C=%d must be a positive multiple of 4", C);
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "dw: empty tensor");
    const int P = K / 2;
    g.N = N; g.H = H; g.W = W; g.C = C;
    g.Ho = (H + 2 * P - K) / stride + 1;
    g.Wo = (W + 2 * P - K) / stride + 1;
    L = make_stencil_layout(C);
    g.cg_total = L.cg_total; g.cgb = L.cgb;
    // gx * chunks workgroups = one resident round: 4 per CU (first generation, <= 128 VGPRs), 3 per CU for the second-generation
    // forward (<= 168 VGPRs: the h-swish / leaky variants spilled inside the row loop at 128; tools/probe/dw_probe.hip shows 2, 3
    // and 4 waves per SIMD stream at the same rate)
    static const int res_env = getenv("MNY_DW_RES") ? atoi(getenv("MNY_DW_RES")) : 768;
    const int resident = dw_use_v2(K, mode) ? (K == 5 ? 512 : res_env) : kMaxParts;          // 5x5: ~170 VGPRs, two workgroups per CU
    int cap = resident / L.chunks > 0 ? resident / L.chunks : 1;
    if (cap > 8) cap &= ~7;                                               // whole XCD rounds (workgroup b runs on XCD b % 8)
    static const int th_env = getenv("MNY_DW_TH") ? atoi(getenv("MNY_DW_TH")) : 0;          // > 0: strip height; -1: balance search
    static const int xcd_env = getenv("MNY_DW_XCD") ? atoi(getenv("MNY_DW_XCD")) : 1;
    static const int nt_env = getenv("MNY_DW_NT") ? atoi(getenv("MNY_DW_NT")) : 1;
    g.xcd = xcd_env;
    g.nt = nt_env;
    int ns;
    if (dw_use_v2(K, mode) && th_env < 0) {
        // strips per column: a workgroup walks ceil(want / cap) strips, the last round partly empty, and every strip pays
        // (K - stride) halo rows of loads without an output row -> pick the split with the best product of both efficiencies
        double best = -1.0;
        ns = 1;
        for (int cand = 1; cand <= g.Ho && g.Ho / cand >= 4; ++cand) {
            const int th = (int)cdiv(g.Ho, cand);
            const int64_t want = cdiv((int64_t)N * g.Wo * cdiv(g.Ho, th), L.ppb);
            const double rounds = (double)want / cap;
            const double fill = rounds <= 1.0 ? 1.0 : rounds / (double)cdiv(want, cap);
            const double eff = fill * th / (th + (K - stride) * 0.75);
            if (eff > best + 1e-9) { best = eff; ns = cand; }
        }
    } else {
        const int th = th_env > 0 ? th_env : 16;                          // first generation: 16 re-reads 2/16 halo rows (8: 2/8); 4.15 -> 4.05 ms
        ns = (int)cdiv(g.Ho, th);
    }
    g.TH = (int)cdiv(g.Ho, ns);
    g.nHS = (int)cdiv(g.Ho, g.TH);
    g.nstrips = (int64_t)N * g.Wo * g.nHS;
    int64_t want = cdiv(g.nstrips, L.ppb);
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < cap ? want : cap);
    return MNY_OK;
}

template <int MODE, typename T>
static int dw_launch(const T* x, const float* sc, const float* sh, int act, const float* w, int flip,
                     const T* addend, T* y, const T* dy, float* parts,
                     int N, int H, int W, int C, int K, int stride, hipStream_t st) {
    DwGeom g; CgLayout L; int gx;
    int rc = dw_geom(g, L, gx, N, H, W, C, K, stride, MODE);
    if (rc) return rc;
    dim3 grid(gx, L.chunks), block(L.threads);
    if (MODE == 1 && dw_use_v2(K, MODE)) {
        const int xf5 = (sc == nullptr && act == MNY_ACT_NONE) ? 0 : (act == MNY_ACT_RELU6 ? 1 : (act == MNY_ACT_HSWISH ? 2 : 4));
#define MNY_W5(S_, X_) hipLaunchKernelGGL((dw5_wgrad_kernel<T, S_, X_>), grid, block, 0, st, x, sc, sh, act, dy, parts, g)
#define MNY_W5S(S_) do { if (xf5 == 0) MNY_W5(S_, 0); else if (xf5 == 1) MNY_W5(S_, 1); else if (xf5 == 2) MNY_W5(S_, 2); else MNY_W5(S_, 4); } while (0)
        if (stride == 1) MNY_W5S(1); else MNY_W5S(2);
#undef MNY_W5S
#undef MNY_W5
        return check_launch("dw5_wgrad_kernel");
    }
    if (MODE == 0 && dw_use_v2(K, MODE) && (addend == nullptr || (sc == nullptr && act == MNY_ACT_NONE))) {     // an addend only occurs without a view (backward-data)
        const int xf2 = (sc == nullptr && act == MNY_ACT_NONE) ? 0 : (act == MNY_ACT_RELU6 ? 1 : (act == MNY_ACT_HSWISH ? 2 : 4));
        // rows requested one step ahead (PF = 1): default for bf16 storage (8-B lanes: see dw_raw); MNY_DW_PF=0/1 forces it for both storage types
        static const int pf_env = getenv("MNY_DW_PF") ? atoi(getenv("MNY_DW_PF")) : -1;
        const int pf = pf_env >= 0 ? (pf_env != 0) : (std::is_same<T, bf16_t>::value ? 1 : 0);
#define MNY_DW2P(S_, X_, A_, N_) do { if (pf) hipLaunchKernelGGL((dw3_fwd_kernel<T, S_, X_, A_, N_, 1>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, parts, g); \
        else hipLaunchKernelGGL((dw3_fwd_kernel<T, S_, X_, A_, N_, 0>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, parts, g); } while (0)
#define MNY_DW2(S_, X_, A_) do { if (g.nt) MNY_DW2P(S_, X_, A_, true); else MNY_DW2P(S_, X_, A_, false); } while (0)
#define MNY_DW2S(S_) do { if (xf2 == 0) { if (addend) MNY_DW2(S_, 0, true); else MNY_DW2(S_, 0, false); } else if (xf2 == 1) MNY_DW2(S_, 1, false); \
        else if (xf2 == 2) MNY_DW2(S_, 2, false); else MNY_DW2(S_, 4, false); } while (0)
        if (K == 3) { if (stride == 1) MNY_DW2S(1); else MNY_DW2S(2); }
#undef MNY_DW2
#undef MNY_DW2P
#define MNY_DW2P(S_, X_, A_, N_) do { if (pf) hipLaunchKernelGGL((dw5_fwd_kernel<T, S_, X_, A_, N_, 1>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, parts, g); \
        else hipLaunchKernelGGL((dw5_fwd_kernel<T, S_, X_, A_, N_, 0>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, parts, g); } while (0)
#define MNY_DW2(S_, X_, A_) do { if (g.nt) MNY_DW2P(S_, X_, A_, true); else MNY_DW2P(S_, X_, A_, false); } while (0)
        if (K == 5) { if (stride == 1) MNY_DW2S(1); else MNY_DW2S(2); }
#undef MNY_DW2S
#undef MNY_DW2
#undef MNY_DW2P
        return check_launch(K == 3 ? "dw3_fwd_kernel" : "dw5_fwd_kernel");
    }
    const int xf = (sc == nullptr && act == MNY_ACT_NONE) ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
#define MNY_DW(KS_, S_) do { if (xf == 0) hipLaunchKernelGGL((dw_slide_kernel<T, KS_, S_, MODE, 0>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, dy, parts, g); \
        else if (xf == 1) hipLaunchKernelGGL((dw_slide_kernel<T, KS_, S_, MODE, 1>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, dy, parts, g); \
        else hipLaunchKernelGGL((dw_slide_kernel<T, KS_, S_, MODE, 2>), grid, block, 0, st, x, sc, sh, act, w, flip, addend, y, dy, parts, g); } while (0)
    if (K == 3 && stride == 1) MNY_DW(3, 1);
    else if (K == 3 && stride == 2) MNY_DW(3, 2);
    else if (K == 5 && stride == 1) MNY_DW(5, 1);
    else MNY_DW(5, 2);
#undef MNY_DW
    return check_launch("dw_slide_kernel");
}

}  // namespace mny

using namespace mny;

extern "C" int mny_dw_stat_parts(int N, int H, int W, int C, int K, int stride) {
    DwGeom g; CgLayout L; int gx;
    if (dw_geom(g, L, gx, N, H, W, C, K, stride)) return MNY_EINVAL;
    return gx;
}
// flags bit 0: bf16 storage (the tile form of dwtile.hip has its own grid)
extern "C" int mny_dw_stat_parts_x(int N, int H, int W, int C, int K, int stride, int flags) {
    if ((K == 3 || K == 5) && C > 0 && C % 4 == 0 && N > 0 && H > 0 && W > 0 && dwt_fwd_use(K, stride, flags & 1, C)) return dwt_fwd_parts(N, H, W, C, K);
    return mny_dw_stat_parts(N, H, W, C, K, stride);
}
extern "C" int mny_dw_wgrad_parts(int N, int H, int W, int C, int K, int stride) {
    DwGeom g; CgLayout L; int gx;
    if (dw_geom(g, L, gx, N, H, W, C, K, stride, 1)) return MNY_EINVAL;
    return gx;
}

template <typename T>
static int dw_fwd_impl(const T* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                       T* y, float* stats, int N, int H, int W, int C, int K, int stride, void* stream) {
    MNY_REQUIRE(x && w && y, "dw_fwd: null pointer");
    if (dwt_fwd_use(K, stride, sizeof(T) == 2 ? 1 : 0, C))                                       // tile form (dwtile.hip)
        return dwt_fwd_launch(sizeof(T) == 2 ? 1 : 0, x, in_scale, in_shift, in_act, w, y, stats, N, H, W, C, K, stream);
    return dw_launch<0, T>(x, in_scale, in_shift, in_act, w, 0, nullptr, y, nullptr, stats, N, H, W, C, K, stride, (hipStream_t)stream);
}
extern "C" int mny_dw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                          float* y, float* stats, int N, int H, int W, int C, int K, int stride, void* stream) {
    MNY_REQUIRE(x && w && y, "dw_fwd: null pointer");
    return dw_fwd_impl<float>(x, in_scale, in_shift, in_act, w, y, stats, N, H, W, C, K, stride, stream);
}
extern "C" int mny_dw_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                               void* y, float* stats, int N, int H, int W, int C, int K, int stride, void* stream) {
    return dw_fwd_impl<bf16_t>((const bf16_t*)x, in_scale, in_shift, in_act, w, (bf16_t*)y, stats, N, H, W, C, K, stride, stream);
}

template <typename T>
static int dw_bwd_data_impl(const T* dy, const float* w, const T* addend, T* dx,
                            int N, int H, int W, int C, int K, int stride, void* stream) {
    MNY_REQUIRE(dy && w && dx, "dw_bwd_data: null pointer");
    if (stride == 1)   // transposed conv == correlation with the flipped filter
        return dw_launch<0, T>(dy, nullptr, nullptr, MNY_ACT_NONE, w, 1, addend, dx, nullptr, nullptr, N, H, W, C, K, 1, (hipStream_t)stream);
    MNY_REQUIRE(stride == 2 && (K == 3 || K == 5) && C % 4 == 0, "dw_bwd_data: unsupported K=%d stride=%d C=%d", K, stride, C);
    const int P = K / 2;
    const int Ho = (H + 2 * P - K) / 2 + 1, Wo = (W + 2 * P - K) / 2 + 1;
    CgLayout L = make_stencil_layout(C);
    int64_t want = cdiv((int64_t)N * H * W, L.ppb);
    dim3 grid((unsigned)(want < 8192 ? want : 8192), L.chunks), block(L.threads);
    int64_t wq = cdiv((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2), L.ppb);
    dim3 gridq((unsigned)(wq < 8192 ? wq : 8192), L.chunks);
    static const bool gather5 = getenv("MNY_DW_S2K5_GATHER") != nullptr;      // A/B: the generic parity-gather kernel
    if (K == 3)
        hipLaunchKernelGGL((dw_bwd_data_s2k3_kernel<T>), gridq, block, 0, (hipStream_t)stream, dy, w, addend, dx, N, H, W, C, Ho, Wo, L.cgb, L.cg_total);
    else if (!gather5)
        hipLaunchKernelGGL((dw_bwd_data_s2k5_kernel<T>), gridq, block, 0, (hipStream_t)stream, dy, w, addend, dx, N, H, W, C, Ho, Wo, L.cgb, L.cg_total);
    else hipLaunchKernelGGL((dw_bwd_data_s2_kernel<T, 5>), grid, block, 0, (hipStream_t)stream, dy, w, addend, dx, N, H, W, C, Ho, Wo, L.cgb, L.cg_total);
    return check_launch("dw_bwd_data_s2_kernel");
}
extern "C" int mny_dw_bwd_data(const float* dy, const float* w, const float* addend, float* dx,
                               int N, int H, int W, int C, int K, int stride, void* stream) {
    return dw_bwd_data_impl<float>(dy, w, addend, dx, N, H, W, C, K, stride, stream);
}
extern "C" int mny_dw_bwd_data_bf16(const void* dy, const float* w, const void* addend, void* dx,
                                    int N, int H, int W, int C, int K, int stride, void* stream) {
    return dw_bwd_data_impl<bf16_t>((const bf16_t*)dy, w, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, K, stride, stream);
}

template <typename T>
static int dw_bwd_weight_impl(const T* x, const float* in_scale, const float* in_shift, int in_act, const T* dy,
                              float* dw, float* ws, int N, int H, int W, int C, int K, int stride, void* stream) {
    MNY_REQUIRE(x && dy && ws, "dw_bwd_weight: null pointer");
    int rc = dw_launch<1, T>(x, in_scale, in_shift, in_act, nullptr, 0, nullptr, nullptr, dy, ws, N, H, W, C, K, stride, (hipStream_t)stream);
    if (rc || !dw) return rc;                       // dw == NULL: partials only (combined later by mny_reduce_batch)
    const int parts = mny_dw_wgrad_parts(N, H, W, C, K, stride);
    return launch_reduce_parts(ws, parts, C * K * K, dw, (hipStream_t)stream);
}
extern "C" int mny_dw_bwd_weight(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                                 float* dw, float* ws, int N, int H, int W, int C, int K, int stride, void* stream) {
    return dw_bwd_weight_impl<float>(x, in_scale, in_shift, in_act, dy, dw, ws, N, H, W, C, K, stride, stream);
}
extern "C" int mny_dw_bwd_weight_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* dy,
                                      float* dw, float* ws, int N, int H, int W, int C, int K, int stride, void* stream) {
    return dw_bwd_weight_impl<bf16_t>((const bf16_t*)x, in_scale, in_shift, in_act, (const bf16_t*)dy, dw, ws, N, H, W, C, K, stride, stream);
}
