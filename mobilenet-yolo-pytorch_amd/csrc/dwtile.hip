// Fused backward (and the 5x5 forward) of a depthwise KxK stride-1 conv+BN+activation unit, TILE form: every 5x5 unit, and on bf16 storage the 3x3 units
// it is faster on (dwt_use() holds the rule; MNY_DWT3=0 / 1 forces it off / on for 3x3).
//
// The register form of dwbwd.hip (thread = 4 channels x one column, every thread rebuilds dY at all K columns it meets) does not
// carry over to 5x5: 25 tap accumulators x 4 channels + a 5-column dY history is > 250 VGPRs, and each element of G, Y would be
// loaded and pushed through the BN-backward / activation arithmetic five times.  Un-fused, a 5x5 unit of MobileNetV3 costs
// bn_bwd_apply + dw5_wgrad (1.0-1.4 TB/s: five transformed input columns per row) + dw_bwd_data + the producer's bn_bwd_reduce.
//
// Here a workgroup owns a tile of PC output columns x CG channel groups and walks down a strip of rows:
//   * every thread loads G, Y of ITS column(s) once per row, rebuilds dY = ca*G*act'(sc*Y+sh) + cb*Y + cc once and parks it (fp32) in
//     a ring of K+1 rows in LDS; the 2*(K/2) halo columns of the tile are rebuilt by one extra pass of one wave (the duty rotates
//     over the whole waves with the row index);
//   * after one barrier per row each thread reads the KxK window of dY around its output pixel(s) from LDS and uses every element
//     twice: dX += w_flipped * D (data gradient as a gather) and dW[tap] += a * D with a = the activated input at the thread's
//     own pixel (weight gradient summed over the INPUT pixels the thread owns: no window of the input is needed);
//   * 3x3: a thread owns 4 channels x 1 column, the 9 taps and 9 accumulators in registers;
//     5x5: 2 channels x 2 ADJACENT columns (the two share the accumulators and 4 of 6 window columns: 30 window reads per 2 outputs instead of
//     50), the 25 accumulator pairs in registers and the 25 taps read from LDS next to the window — taps AND accumulators in registers is
//     100 VGPRs of the 168 three waves per SIMD leave;
//   * the next row's G, Y, X (and addend) are requested before the current row is consumed (raw registers, widened when used);
//   * RED (the input is the raw output of a conv+BN+act unit consumed only here): the unit's BN-backward sums leave with dX.
// Partial rows [gridDim.x][C*K*K] (taps) and [gridDim.x][2][C] (RED) are combined by the usual fixed-order launches.
// Measurements, the routing rule (which units take this form) and what did not work: dwt_use() below, LAB_NOTES.md R5.5,
// profiles/r05_pmc_dwtile.txt.
//
// replaces, for these units, the autograd backward of nn.Conv2d(groups=C, kernel 5, stride 1) + nn.BatchNorm2d + ReLU / h-swish
// (models/mobilenetv3.py:54-56,68-69).
#include <stdlib.h>

#include "common.h"

namespace mny {

struct DwtGeom {
    int N, H, W, C;
    int TH, nHS;          // strip height, strips per column tile
    int PC, nWT;          // output columns per tile, column tiles per row
    int CG, cg_total;     // channel groups (of CPT channels) per workgroup / in the tensor
    int64_t ntiles;       // N * nHS * nWT
    int xcd;
};

template <int CPT> struct VC { v2f v[CPT / 2]; };

template <int CPT> __device__ __forceinline__ VC<CPT> vc_zero() {
    VC<CPT> r;
#pragma unroll
    for (int j = 0; j < CPT / 2; ++j) r.v[j] = v2f{0.f, 0.f};
    return r;
}
template <int CPT> __device__ __forceinline__ VC<CPT> vc_lds(const float* p) {
    VC<CPT> r;
    if constexpr (CPT == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        r.v[0] = v2f{t.x, t.y}; r.v[1] = v2f{t.z, t.w};
    } else {
        r.v[0] = *reinterpret_cast<const v2f*>(p);
    }
    return r;
}
template <int CPT> __device__ __forceinline__ void vc_sts(float* p, VC<CPT> v) {
    if constexpr (CPT == 4) *reinterpret_cast<float4*>(p) = make_float4(v.v[0].x, v.v[0].y, v.v[1].x, v.v[1].y);
    else *reinterpret_cast<v2f*>(p) = v.v[0];
}

// raw (as stored) image of CPT channels: what a prefetched load holds until it is used
template <typename T, int CPT> struct RawC { uint32_t u[CPT * sizeof(T) / 4]; };
template <typename T, int CPT> __device__ __forceinline__ RawC<T, CPT> raw_ld(const T* p) {
    constexpr int DW = CPT * sizeof(T) / 4;
    RawC<T, CPT> r;
    if constexpr (DW == 4) { const uint4 t = *reinterpret_cast<const uint4*>(p); r.u[0] = t.x; r.u[1] = t.y; r.u[2] = t.z; r.u[3] = t.w; }
    else if constexpr (DW == 2) { const uint2 t = *reinterpret_cast<const uint2*>(p); r.u[0] = t.x; r.u[1] = t.y; }
    else r.u[0] = *reinterpret_cast<const uint32_t*>(p);
    return r;
}
template <typename T, int CPT> __device__ __forceinline__ VC<CPT> raw_widen(RawC<T, CPT> r) {
    VC<CPT> o;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int j = 0; j < CPT / 2; ++j) o.v[j] = v2f{__uint_as_float(r.u[2 * j]), __uint_as_float(r.u[2 * j + 1])};
    } else {
#pragma unroll
        for (int j = 0; j < CPT / 2; ++j) o.v[j] = v2f{__uint_as_float(r.u[j] << 16), __uint_as_float(r.u[j] & 0xffff0000u)};
    }
    return o;
}
template <typename T, int CPT> __device__ __forceinline__ void vc_store_stream(T* p, VC<CPT> v) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (CPT == 4) {
            const mny_f4v t = {v.v[0].x, v.v[0].y, v.v[1].x, v.v[1].y};
            __builtin_nontemporal_store(t, reinterpret_cast<mny_f4v*>(p));
        } else {
            __builtin_nontemporal_store(v.v[0], reinterpret_cast<v2f*>(p));
        }
    } else {
        if constexpr (CPT == 4) {
            const mny_u2v t = {pack_bf16x2(v.v[0].x, v.v[0].y), pack_bf16x2(v.v[1].x, v.v[1].y)};
            __builtin_nontemporal_store(t, reinterpret_cast<mny_u2v*>(p));
        } else {
            __builtin_nontemporal_store(pack_bf16x2(v.v[0].x, v.v[0].y), reinterpret_cast<uint32_t*>(p));
        }
    }
}

// constants in LDS, [k][CG][CPT]: 0 scale, 1 shift (this unit's BN), 2..4 ca, cb, cc (mny_bn_bwd_finalize), 5, 6 the input view's scale / shift,
// 7, 8 mean / invstd of the unit that produced the input (RED)
constexpr int kDwtConsts = 9;
// channel groups per workgroup: 16 (5x5: 32 channels, 3x3: 64): the 2R halo columns of a row are at most one wave's worth of elements,
// a pixel is 64-256 contiguous bytes, and the window / constant reads use immediate LDS offsets
constexpr int kDwtCG = 16;
#ifndef MNY_DWT_PD
#define MNY_DWT_PD 1
#endif
#ifndef MNY_DWT_ALLW2
#define MNY_DWT_ALLW2 0
#endif

// KS: 3 / 5.  CPT: channels per thread.  NC: adjacent output columns per thread (they share the taps, the tap accumulators and most of the
// window reads).  TLDS: the taps are read from LDS next to the window (5x5: 25 x CPT registers the kernel does not have) instead of registers.
template <typename T, int KS, int CPT, int NC, bool TLDS, bool RED, int AM, int XF, int WPE, int PD, int CG_>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void dwb_tile_kernel(
    const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    const float* __restrict__ coef, const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    int in_act, const float* __restrict__ w, const T* __restrict__ addend, T* __restrict__ dx, float* __restrict__ parts,
    const float* __restrict__ in_mean, const float* __restrict__ in_invstd, float* __restrict__ in_red, DwtGeom gm) {
    constexpr int am = AM, xf = XF;             // AM: 0 none, 1 clamp family, 2 h-swish; XF: 0 as is, 1 ReLU6, 2 h-swish, 3 max(z, slope z)
    constexpr int R = KS / 2, KK = KS * KS, NV = CPT / 2, S = KS + 1, WC = NC + 2 * R;
    using V = VC<CPT>;
    using Raw = RawC<T, CPT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CG = CG_;                      // channel groups per workgroup: 16; 4 / 18 for the 16- / 72-channel layers (3x3, bf16 storage)
    const int PT = gm.PC / NC;                   // thread columns of a tile; PC = PT * NC output columns
    const int PC = gm.PC, SW = PC + 2 * R;
    constexpr int colf = CG * CPT;               // floats per column of a ring row
    const int rowf = SW * colf;                  // floats per ring row
    constexpr int NCST = kDwtConsts + (TLDS ? KK : 0);
    float* const cst = smem;                     // [kDwtConsts (+ KK flipped taps)][CG][CPT]
    float* const ring = smem + NCST * colf;      // [S][SW][CG][CPT]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int cgl = tid % CG, pt = tid / CG;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NWD = nthr >= 64 ? nthr >> 6 : 1;  // whole waves: the halo duty rotates over these (a trailing partial wave has too few lanes)
    const bool cvalid = (int)blockIdx.y * CG + cgl < gm.cg_total;
    const int c = min((int)blockIdx.y * CG + cgl, gm.cg_total - 1) * CPT;      // threads past the last channel group repeat it and store nothing

    if (pt == 0) {
        auto put = [&](int k, const float* src, float fill) {
#pragma unroll
            for (int e = 0; e < CPT; ++e) cst[(k * CG + cgl) * CPT + e] = src ? src[c + e] : fill;
        };
        put(0, scale, 1.f); put(1, shift, 0.f);
        put(2, coef, 0.f); put(3, coef + gm.C, 0.f); put(4, coef + 2 * gm.C, 0.f);
        put(5, xf != 0 ? in_scale : nullptr, 1.f); put(6, xf != 0 ? in_shift : nullptr, 0.f);
        put(7, RED ? in_mean : nullptr, 0.f); put(8, RED ? in_invstd : nullptr, 1.f);
    }
    // taps, flipped: window element u = (dr, dq) meets tap KK-1-u in both products
    V tap[TLDS ? 1 : KK];
    if constexpr (TLDS) {
        for (int u = pt; u < KK; u += (nthr / CG))
#pragma unroll
            for (int e = 0; e < CPT; ++e) cst[((kDwtConsts + u) * CG + cgl) * CPT + e] = w[(c + e) * KK + (KK - 1 - u)];
    } else {
#pragma unroll
        for (int u = 0; u < KK; ++u)
#pragma unroll
            for (int j = 0; j < NV; ++j) tap[u].v[j] = v2f{w[(c + 2 * j) * KK + (KK - 1 - u)], w[(c + 2 * j + 1) * KK + (KK - 1 - u)]};
    }
    V wacc[KK];
#pragma unroll
    for (int u = 0; u < KK; ++u) wacc[u] = vc_zero<CPT>();
    V rs1 = vc_zero<CPT>(), rs2 = vc_zero<CPT>();

    // halo elements of a row: e in [0, 2R*CG) <= 64, one per lane of the wave on duty; e < R*CG: left columns (ring column e / CG), else right
    // (ring column PC + e / CG)
    constexpr int HE = 2 * R * CG;
    const bool h_v = lane < HE;
    const int h_e = min(lane, HE - 1);
    const int h_s = h_e / CG;
    const int h_cgl = h_e - h_s * CG;
    const int h_idx = (h_s < R ? h_e : PC * CG + h_e) * CPT;                // float offset inside a ring row
    const int h_col = h_s < R ? h_s - R : PC + h_s - R;                     // image column relative to the tile's first output column
    const int h_c = min((int)blockIdx.y * CG + h_cgl, gm.cg_total - 1) * CPT;
    __syncthreads();

    const float slope = act_slope(act), hi = act_hi(act), xslope = act_slope(in_act);
    // dY of CPT channels from (G, Y) and the constants of channel group `q` of this workgroup
    auto dyf = [&](V G, V Y, int q, float mask) {
        const float* cb_ = cst + q * CPT;
        const V sc = vc_lds<CPT>(cb_), sh = vc_lds<CPT>(cb_ + colf), ca = vc_lds<CPT>(cb_ + 2 * colf), cb = vc_lds<CPT>(cb_ + 3 * colf),
                cc = vc_lds<CPT>(cb_ + 4 * colf);
        V d;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v2f dd = G.v[j];
            if (am != 0) {
                const v2f z = __builtin_elementwise_fma(Y.v[j], sc.v[j], sh.v[j]);
                v2f f;
                if (am == 2) {
                    f.x = z.x <= -3.f ? 0.f : (z.x >= 3.f ? 1.f : (2.f * z.x + 3.f) * (1.f / 6.f));
                    f.y = z.y <= -3.f ? 0.f : (z.y >= 3.f ? 1.f : (2.f * z.y + 3.f) * (1.f / 6.f));
                } else {
                    f.x = (z.x > 0.f ? 1.f : slope) * (z.x < hi ? 1.f : 0.f);
                    f.y = (z.y > 0.f ? 1.f : slope) * (z.y < hi ? 1.f : 0.f);
                }
                dd = dd * f;
            }
            d.v[j] = __builtin_elementwise_fma(ca.v[j], dd, __builtin_elementwise_fma(cb.v[j], Y.v[j], cc.v[j])) * v2f{mask, mask};
        }
        return d;
    };

    const int gxd = gridDim.x;
    const int lb = (gm.xcd && (gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t pitch = (int64_t)gm.W * gm.C;            // elements per image row (an image stays below 2^31 bytes: 32-bit offsets inside it)
    for (int64_t tile = lb; tile < gm.ntiles; tile += gxd) {
        const int wt = (int)(tile % gm.nWT);
        const int hs = (int)((tile / gm.nWT) % gm.nHS);
        const int n = (int)(tile / ((int64_t)gm.nWT * gm.nHS));
        const int w0 = wt * PC, h0 = hs * gm.TH, h1 = min(h0 + gm.TH, gm.H);
        const int64_t img = (int64_t)n * gm.H * pitch;
        const T* const gi = g + img; const T* const yi = y + img; const T* const xi = x + img;
        const T* const ai = addend ? addend + img : nullptr;
        T* const di = dx + img;
        bool colok[NC];
        float colm[NC];
        unsigned own_b[NC];                                 // byte offsets inside an image row
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int wo = w0 + pt * NC + j;
            colok[j] = wo < gm.W;
            colm[j] = colok[j] ? 1.f : 0.f;
            own_b[j] = (unsigned)((min(wo, gm.W - 1) * gm.C + c) * (int)sizeof(T));
        }
        const int hwi = w0 + h_col;
        const float hm = (h_v && hwi >= 0 && hwi < gm.W) ? 1.f : 0.f;
        const unsigned h_b = (unsigned)((min(max(hwi, 0), gm.W - 1) * gm.C + h_c) * (int)sizeof(T));
        __syncthreads();                                   // the previous tile's window reads are done: the ring may be overwritten

        const int r_begin = h0 - R, r_end = h1 + R;          // dY rows produced for this strip
        int duty = wave < NWD ? ((wave - r_begin) % NWD + NWD) % NWD : -1;       // 0: this wave rebuilds the halo columns of the row
        // PD rows of G, Y (+ halo), X, addend are in flight: one register set per row of the queue, the row loop is unrolled PD times so that
        // every set keeps its registers (shifting the queue would wait for the newest load)
        struct RowSet { Raw g[NC], y[NC], x[NC], a[NC], hg, hy; };
        RowSet rs_[PD];
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int q = 0; q < (int)(sizeof(Raw) / 4); ++q) {
                rs_[k].hg.u[q] = 0u; rs_[k].hy.u[q] = 0u;
#pragma unroll
                for (int j = 0; j < NC; ++j) rs_[k].a[j].u[q] = 0u;
            }
        const int pdm = PD % NWD;
        auto duty_at = [&](int d, int ahead_mod) {            // the duty counter `ahead` rows later (it counts down with the row index)
            if (d < 0) return -1;
            const int t = d - ahead_mod;
            return t < 0 ? t + NWD : t;
        };
        auto issue_gy = [&](RowSet& q_, int r, int k) {
            const int64_t ro = (int64_t)min(max(r, 0), gm.H - 1) * pitch;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                q_.g[j] = raw_ld<T, CPT>(at_bytes(gi + ro, own_b[j]));
                q_.y[j] = raw_ld<T, CPT>(at_bytes(yi + ro, own_b[j]));
            }
            if (k == 0) {
                q_.hg = raw_ld<T, CPT>(at_bytes(gi + ro, h_b));
                q_.hy = raw_ld<T, CPT>(at_bytes(yi + ro, h_b));
            }
        };
        auto issue_xa = [&](RowSet& q_, int i) {
            const int64_t ro = (int64_t)min(max(i, 0), gm.H - 1) * pitch;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                q_.x[j] = raw_ld<T, CPT>(at_bytes(xi + ro, own_b[j]));
                if (ai) q_.a[j] = raw_ld<T, CPT>(at_bytes(ai + ro, own_b[j]));
            }
        };
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            issue_gy(rs_[k], r_begin + k, duty_at(duty, k % NWD));
            issue_xa(rs_[k], r_begin + k - R);
        }
        int slot_w = 0;
        auto step = [&](RowSet& q_, int r) {
            const float rowm = (r >= 0 && r < gm.H) ? 1.f : 0.f;
            float* const wrow = ring + slot_w * rowf;
            {
                int q = cgl;
                asm volatile("" : "+v"(q));                  // opaque per iteration: the constant reads stay in the loop
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const V d = dyf(raw_widen<T, CPT>(q_.g[j]), raw_widen<T, CPT>(q_.y[j]), q, rowm * colm[j]);
                    vc_sts<CPT>(wrow + ((pt * NC + j + R) * CG + cgl) * CPT, d);
                }
            }
            if (duty == 0) {
                const V d = dyf(raw_widen<T, CPT>(q_.hg), raw_widen<T, CPT>(q_.hy), h_cgl, rowm * hm);
                if (h_v) vc_sts<CPT>(wrow + h_idx, d);
            }
            issue_gy(q_, r + PD, duty_at(duty, pdm));          // this set's next row (past the strip at the end: clamped, unused)
            duty = duty < 0 ? -1 : (duty == 0 ? NWD - 1 : duty - 1);
            __syncthreads();
            const int i = r - R;                             // output row whose window [i-R, i+R] is complete now
            if (i >= h0) {
                V xr[NC], acc[NC], A[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    xr[j] = raw_widen<T, CPT>(q_.x[j]);
                    acc[j] = vc_zero<CPT>();
                    if (ai) acc[j] = raw_widen<T, CPT>(q_.a[j]);
                }
                issue_xa(q_, i + PD);
                int q = cgl;
                asm volatile("" : "+v"(q));
                const float* cb_ = cst + q * CPT;
                const V xsc = vc_lds<CPT>(cb_ + 5 * colf), xsh = vc_lds<CPT>(cb_ + 6 * colf);
#pragma unroll
                for (int jc = 0; jc < NC; ++jc) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        v2f a = xr[jc].v[j];
                        if (xf != 0) {
                            const v2f z = __builtin_elementwise_fma(xr[jc].v[j], xsc.v[j], xsh.v[j]);
                            if (xf == 1) a = v2f{__builtin_amdgcn_fmed3f(z.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.y, 0.f, 6.f)};
                            else if (xf == 3) { const v2f t = z * v2f{xslope, xslope}; a = v2f{fmaxf(z.x, t.x), fmaxf(z.y, t.y)}; }
                            else {
                                const v2f t = z + v2f{3.f, 3.f};
                                a = z * v2f{__builtin_amdgcn_fmed3f(t.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(t.y, 0.f, 6.f)} * v2f{1.f / 6.f, 1.f / 6.f};
                            }
                        }
                        A[jc].v[j] = a * v2f{colm[jc], colm[jc]};
                    }
                }
                // the window: thread column pt covers ring columns pt*NC .. pt*NC + NC - 1 + 2R; one row of reads ahead of the products
                const float* wb = ring + (pt * NC * CG + cgl) * CPT;
                const float* tb = cb_ + kDwtConsts * colf;
                auto row_base = [&](int dr) {
                    int sl = slot_w + 2 + dr;                // ring slot of dY row i - R + dr (S = 2R + 2 slots)
                    sl = sl >= S ? sl - S : sl;
                    return wb + sl * rowf;
                };
                if constexpr (TLDS) {
                    // taps next to the window, one kernel row at a time and nothing read ahead (the registers that would take are the fourth wave
                    // per SIMD, which hides the LDS latency better than a deeper queue in three)
#pragma unroll
                    for (int dr = 0; dr < KS; ++dr) {
                        const float* rb = row_base(dr);
                        V D[WC], Tc[KS];
#pragma unroll
                        for (int dq = 0; dq < WC; ++dq) D[dq] = vc_lds<CPT>(rb + dq * colf);
#pragma unroll
                        for (int dq = 0; dq < KS; ++dq) Tc[dq] = vc_lds<CPT>(tb + (dr * KS + dq) * colf);
#pragma unroll
                        for (int dq = 0; dq < KS; ++dq) {
                            const int u = dr * KS + dq;
#pragma unroll
                            for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                                for (int j = 0; j < NV; ++j) {
                                    acc[jc].v[j] = __builtin_elementwise_fma(Tc[dq].v[j], D[jc + dq].v[j], acc[jc].v[j]);
                                    wacc[u].v[j] = __builtin_elementwise_fma(A[jc].v[j], D[jc + dq].v[j], wacc[u].v[j]);
                                }
                        }
                        // the products stay HERE: without the pin the compiler sinks the whole dX accumulation into the (colok && cvalid) store
                        // branch and keeps all 55 LDS results alive up to it; the next kernel row's reads stay behind this row's products
#pragma unroll
                        for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                            for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(acc[jc].v[j]));
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    V Dn[WC];
                    {
                        const float* rb = row_base(0);
#pragma unroll
                        for (int dq = 0; dq < WC; ++dq) Dn[dq] = vc_lds<CPT>(rb + dq * colf);
                    }
#pragma unroll
                    for (int dr = 0; dr < KS; ++dr) {
                        V D[WC];
#pragma unroll
                        for (int dq = 0; dq < WC; ++dq) D[dq] = Dn[dq];
                        if (dr + 1 < KS) {
                            const float* rb = row_base(dr + 1);
#pragma unroll
                            for (int dq = 0; dq < WC; ++dq) Dn[dq] = vc_lds<CPT>(rb + dq * colf);
                        }
#pragma unroll
                        for (int dq = 0; dq < KS; ++dq) {
                            const int u = dr * KS + dq;
#pragma unroll
                            for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                                for (int j = 0; j < NV; ++j) {
                                    acc[jc].v[j] = __builtin_elementwise_fma(tap[u].v[j], D[jc + dq].v[j], acc[jc].v[j]);
                                    wacc[u].v[j] = __builtin_elementwise_fma(A[jc].v[j], D[jc + dq].v[j], wacc[u].v[j]);
                                }
                        }
#pragma unroll
                        for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                            for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(acc[jc].v[j]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                V mu, is;
                if (RED) { mu = vc_lds<CPT>(cb_ + 7 * colf); is = vc_lds<CPT>(cb_ + 8 * colf); }
#pragma unroll
                for (int jc = 0; jc < NC; ++jc) {
                    if (colok[jc] && cvalid) vc_store_stream<T, CPT>(at_bytes(di + (int64_t)i * pitch, own_b[jc]), acc[jc]);
                    if (RED) {
#pragma unroll
                        for (int j = 0; j < NV; ++j) {
                            const v2f gq = v2f{stored<T>(acc[jc].v[j].x), stored<T>(acc[jc].v[j].y)} * v2f{colm[jc], colm[jc]};
                            const v2f z = __builtin_elementwise_fma(xr[jc].v[j], xsc.v[j], xsh.v[j]);
                            v2f f;
                            if (xf == 1) f = v2f{(z.x > 0.f ? 1.f : 0.f) * (z.x < 6.f ? 1.f : 0.f), (z.y > 0.f ? 1.f : 0.f) * (z.y < 6.f ? 1.f : 0.f)};
                            else if (xf == 2) f = v2f{z.x <= -3.f ? 0.f : (z.x >= 3.f ? 1.f : (2.f * z.x + 3.f) * (1.f / 6.f)),
                                                      z.y <= -3.f ? 0.f : (z.y >= 3.f ? 1.f : (2.f * z.y + 3.f) * (1.f / 6.f))};
                            else f = v2f{z.x > 0.f ? 1.f : xslope, z.y > 0.f ? 1.f : xslope};
                            const v2f dz = gq * f;
                            rs1.v[j] += dz;
                            rs2.v[j] = __builtin_elementwise_fma(dz, (xr[jc].v[j] - mu.v[j]) * is.v[j], rs2.v[j]);
                        }
                    }
                }
            } else {
                issue_xa(q_, i + PD);
            }
            slot_w = slot_w + 1 == S ? 0 : slot_w + 1;
        };
        for (int r = r_begin; r < r_end; r += PD) {
#pragma unroll
            for (int k = 0; k < PD; ++k)
                if (r + k < r_end) step(rs_[k], r + k);
        }
    }

    // deterministic block reductions over the thread columns (the ring is free now), several taps per round
    float* const red = ring;
    const int per = nthr * CPT;                              // floats one tap of the whole block takes
    int tpr = (S * rowf) / per;
    tpr = tpr < 1 ? 1 : (tpr > KK ? KK : tpr);
    for (int u0 = 0; u0 < KK; u0 += tpr) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KK; ++u)
            if (u >= u0 && u < u0 + tpr) vc_sts<CPT>(red + (u - u0) * per + tid * CPT, wacc[u]);
        __syncthreads();
        const int cnt = min(tpr, KK - u0);
        for (int j = pt; j < cnt; j += PT) {                 // thread (cgl, pt) sums tap u0 + j of its channel group over the thread columns
            if (!cvalid) continue;
            V a = vc_zero<CPT>();
            for (int p = 0; p < PT; ++p) {
                const V t = vc_lds<CPT>(red + j * per + (p * CG + cgl) * CPT);
#pragma unroll
                for (int q = 0; q < NV; ++q) a.v[q] += t.v[q];
            }
            float* dst = parts + (int64_t)blockIdx.x * gm.C * KK;
            const int tp = KK - 1 - (u0 + j);
#pragma unroll
            for (int q = 0; q < NV; ++q) { dst[(c + 2 * q) * KK + tp] = a.v[q].x; dst[(c + 2 * q + 1) * KK + tp] = a.v[q].y; }
        }
    }
    if (RED) {
        __syncthreads();
        vc_sts<CPT>(red + tid * CPT, rs1);
        vc_sts<CPT>(red + per + tid * CPT, rs2);
        __syncthreads();
        if (pt < 2 && cvalid) {
            V a = vc_zero<CPT>();
            for (int p = 0; p < PT; ++p) {
                const V t = vc_lds<CPT>(red + pt * per + (p * CG + cgl) * CPT);
#pragma unroll
                for (int q = 0; q < NV; ++q) a.v[q] += t.v[q];
            }
            float* dst = in_red + (int64_t)blockIdx.x * 2 * gm.C + pt * gm.C;
#pragma unroll
            for (int q = 0; q < NV; ++q) { dst[c + 2 * q] = a.v[q].x; dst[c + 2 * q + 1] = a.v[q].y; }
        }
    }
}

// ---- forward, tile form -------------------------------------------------------------------------------------------------------
// y = depthwise KxK stride-1 correlation of the ACTIVATED input view + the per-channel (sum, sum of squares) of the stored output (the unit's
// BatchNorm statistics), bf16 storage.  The register forms (dw3_fwd_kernel / dw5_fwd_kernel) load and transform K input columns per thread and
// row (1.1-1.4 TB/s for 5x5, 1.3-3.4 TB/s for 3x3 on the MobileNetV3 512x512 shapes); here every thread loads and transforms ITS column(s) once,
// parks the fp32 result in the LDS ring and reads the window back.  4 channels per thread, taps in registers (no accumulators to compete with),
// NC = 2 adjacent output columns per thread for 5x5 (30 window reads per 2 outputs instead of 50).
// replaces nn.Conv2d(groups=C, stride 1) in front of its BatchNorm (models/mobilenetv3.py:54-55, 68-69; models/mobilenetv2.py:65-66, 79-80).
template <typename T, int KS, int NC, int XF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void dwf_tile_kernel(
    const T* __restrict__ x, const float* __restrict__ in_scale, const float* __restrict__ in_shift, int in_act, const float* __restrict__ w,
    T* __restrict__ y, float* __restrict__ parts, DwtGeom gm) {
    constexpr int CPT = 4, R = KS / 2, KK = KS * KS, NV = CPT / 2, S = KS + 1, WC = NC + 2 * R;
    using V = VC<CPT>;
    using Raw = RawC<T, CPT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CG = kDwtCG;
    const int PT = gm.PC / NC;
    const int PC = gm.PC, SW = PC + 2 * R;
    constexpr int colf = CG * CPT;
    const int rowf = SW * colf;
    float* const cst = smem;                     // [2][CG][CPT]: the input view's scale / shift
    float* const ring = smem + 2 * colf;         // [S][SW][CG][CPT]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int cgl = tid % CG, pt = tid / CG;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NWD = nthr >= 64 ? nthr >> 6 : 1;
    const bool cvalid = (int)blockIdx.y * CG + cgl < gm.cg_total;
    const int c = min((int)blockIdx.y * CG + cgl, gm.cg_total - 1) * CPT;
    if (pt == 0) {
#pragma unroll
        for (int e = 0; e < CPT; ++e) {
            cst[cgl * CPT + e] = (XF != 0 && in_scale) ? in_scale[c + e] : 1.f;
            cst[colf + cgl * CPT + e] = (XF != 0 && in_scale) ? in_shift[c + e] : 0.f;
        }
    }
    V tap[KK];
#pragma unroll
    for (int u = 0; u < KK; ++u)
#pragma unroll
        for (int j = 0; j < NV; ++j) tap[u].v[j] = v2f{w[(c + 2 * j) * KK + u], w[(c + 2 * j + 1) * KK + u]};
    V st1 = vc_zero<CPT>(), st2 = vc_zero<CPT>();
    constexpr int HE = 2 * R * CG;
    const bool h_v = lane < HE;
    const int h_e = min(lane, HE - 1);
    const int h_s = h_e / CG;
    const int h_cgl = h_e - h_s * CG;
    const int h_idx = (h_s < R ? h_e : PC * CG + h_e) * CPT;
    const int h_col = h_s < R ? h_s - R : PC + h_s - R;
    const int h_c = min((int)blockIdx.y * CG + h_cgl, gm.cg_total - 1) * CPT;
    __syncthreads();
    const float xslope = act_slope(in_act);
    auto xform = [&](V v, int q, float mask) {
        const V sc = vc_lds<CPT>(cst + q * CPT), sh = vc_lds<CPT>(cst + colf + q * CPT);
        V o;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v2f a = v.v[j];
            if (XF != 0) {
                const v2f z = __builtin_elementwise_fma(v.v[j], sc.v[j], sh.v[j]);
                if (XF == 1) a = v2f{__builtin_amdgcn_fmed3f(z.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(z.y, 0.f, 6.f)};
                else if (XF == 3) { const v2f t = z * v2f{xslope, xslope}; a = v2f{fmaxf(z.x, t.x), fmaxf(z.y, t.y)}; }
                else {
                    const v2f t = z + v2f{3.f, 3.f};
                    a = z * v2f{__builtin_amdgcn_fmed3f(t.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(t.y, 0.f, 6.f)} * v2f{1.f / 6.f, 1.f / 6.f};
                }
            }
            o.v[j] = a * v2f{mask, mask};                      // zero padding applies to the ACTIVATED input
        }
        return o;
    };
    const int gxd = gridDim.x;
    const int lb = (gm.xcd && (gxd & 7) == 0) ? (int)(blockIdx.x & 7) * (gxd >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t pitch = (int64_t)gm.W * gm.C;
    for (int64_t tile = lb; tile < gm.ntiles; tile += gxd) {
        const int wt = (int)(tile % gm.nWT);
        const int hs = (int)((tile / gm.nWT) % gm.nHS);
        const int n = (int)(tile / ((int64_t)gm.nWT * gm.nHS));
        const int w0 = wt * PC, h0 = hs * gm.TH, h1 = min(h0 + gm.TH, gm.H);
        const int64_t img = (int64_t)n * gm.H * pitch;
        const T* const xi = x + img;
        T* const yi = y + img;
        bool colok[NC];
        float colm[NC];
        unsigned own_b[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int wo = w0 + pt * NC + j;
            colok[j] = wo < gm.W;
            colm[j] = colok[j] ? 1.f : 0.f;
            own_b[j] = (unsigned)((min(wo, gm.W - 1) * gm.C + c) * (int)sizeof(T));
        }
        const int hwi = w0 + h_col;
        const float hm = (h_v && hwi >= 0 && hwi < gm.W) ? 1.f : 0.f;
        const unsigned h_b = (unsigned)((min(max(hwi, 0), gm.W - 1) * gm.C + h_c) * (int)sizeof(T));
        __syncthreads();
        const int r_begin = h0 - R, r_end = h1 + R;          // input rows this strip reads
        int duty = wave < NWD ? ((wave - r_begin) % NWD + NWD) % NWD : -1;
        Raw px[NC], phx;
#pragma unroll
        for (int q = 0; q < (int)(sizeof(Raw) / 4); ++q) phx.u[q] = 0u;
        auto issue = [&](int r, int k) {
            const int64_t ro = (int64_t)min(max(r, 0), gm.H - 1) * pitch;
#pragma unroll
            for (int j = 0; j < NC; ++j) px[j] = raw_ld<T, CPT>(at_bytes(xi + ro, own_b[j]));
            if (k == 0) phx = raw_ld<T, CPT>(at_bytes(xi + ro, h_b));
        };
        issue(r_begin, duty);
        int slot_w = 0;
        for (int r = r_begin; r < r_end; ++r) {
            const float rowm = (r >= 0 && r < gm.H) ? 1.f : 0.f;
            float* const wrow = ring + slot_w * rowf;
            {
                int q = cgl;
                asm volatile("" : "+v"(q));
#pragma unroll
                for (int j = 0; j < NC; ++j) vc_sts<CPT>(wrow + ((pt * NC + j + R) * CG + cgl) * CPT, xform(raw_widen<T, CPT>(px[j]), q, rowm * colm[j]));
            }
            if (duty == 0) {
                const V a = xform(raw_widen<T, CPT>(phx), h_cgl, rowm * hm);
                if (h_v) vc_sts<CPT>(wrow + h_idx, a);
            }
            duty = duty < 0 ? -1 : (duty == 0 ? NWD - 1 : duty - 1);
            issue(r + 1, duty);
            __syncthreads();
            const int i = r - R;
            if (i >= h0) {
                V acc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[j] = vc_zero<CPT>();
                const float* wb = ring + (pt * NC * CG + cgl) * CPT;
#pragma unroll
                for (int dr = 0; dr < KS; ++dr) {
                    int sl = slot_w + 2 + dr;
                    sl = sl >= S ? sl - S : sl;
                    const float* rb = wb + sl * rowf;
                    V D[WC];
#pragma unroll
                    for (int dq = 0; dq < WC; ++dq) D[dq] = vc_lds<CPT>(rb + dq * colf);
#pragma unroll
                    for (int dq = 0; dq < KS; ++dq)
#pragma unroll
                        for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                            for (int j = 0; j < NV; ++j) acc[jc].v[j] = __builtin_elementwise_fma(tap[dr * KS + dq].v[j], D[jc + dq].v[j], acc[jc].v[j]);
#pragma unroll
                    for (int jc = 0; jc < NC; ++jc)
#pragma unroll
                        for (int j = 0; j < NV; ++j) asm volatile("" : "+v"(acc[jc].v[j]));
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int jc = 0; jc < NC; ++jc) {
                    if (colok[jc] && cvalid) vc_store_stream<T, CPT>(at_bytes(yi + (int64_t)i * pitch, own_b[jc]), acc[jc]);
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        const v2f q = v2f{stored<T>(acc[jc].v[j].x), stored<T>(acc[jc].v[j].y)} * v2f{colm[jc], colm[jc]};
                        st1.v[j] += q;
                        st2.v[j] = __builtin_elementwise_fma(q, q, st2.v[j]);
                    }
                }
            }
            slot_w = slot_w + 1 == S ? 0 : slot_w + 1;
        }
    }
    if (parts == nullptr) return;
    float* const red = ring;
    const int per = nthr * CPT;
    __syncthreads();
    vc_sts<CPT>(red + tid * CPT, st1);
    vc_sts<CPT>(red + per + tid * CPT, st2);
    __syncthreads();
    if (pt < 2 && cvalid) {
        V a = vc_zero<CPT>();
        for (int p = 0; p < PT; ++p) {
            const V t = vc_lds<CPT>(red + pt * per + (p * CG + cgl) * CPT);
#pragma unroll
            for (int q = 0; q < NV; ++q) a.v[q] += t.v[q];
        }
        float* dst = parts + (int64_t)blockIdx.x * 2 * gm.C + pt * gm.C;
#pragma unroll
        for (int q = 0; q < NV; ++q) { dst[c + 2 * q] = a.v[q].x; dst[c + 2 * q + 1] = a.v[q].y; }
    }
}

// ---- geometry -------------------------------------------------------------------------------------------------------------
static inline int dwt_cpt(int K) { return K == 5 ? 2 : 4; }
static inline int dwt_nc(int K) { return K == 5 ? 2 : 1; }
// channel groups per workgroup of the backward tile kernel: 16, except the 3x3 layers whose channel count fills 16-group chunks badly:
// C = 16 (4 groups: a pixel is 32 contiguous bytes, a wave covers 16 columns) and C = 72 (18 groups in one chunk)
static inline int dwt_pick_cg(int K, int C, int bf) {
    if (bf && K == 3 && C == 16) return 4;
    if (bf && K == 3 && C == 72) return 18;
    return kDwtCG;
}

static int dwt_geom(DwtGeom& g, int& gx, int& chunks, int& threads, size_t& lds, int N, int H, int W, int C, int K, int bf, bool fwd = false) {
    MNY_REQUIRE(K == 3 || K == 5, "dw_bnbwd (tile form): K=%d is not 3 or 5", K);
    const int CPT = fwd ? 4 : dwt_cpt(K), R = K / 2;
    MNY_REQUIRE(C % 4 == 0 && C > 0, "dw_bnbwd: C=%d must be a positive multiple of 4", C);
    MNY_REQUIRE(N > 0 && H > 0 && W > 0, "dw_bnbwd: empty tensor");
    g.N = N; g.H = H; g.W = W; g.C = C;
    g.cg_total = C / CPT;
    // CG = 16 channel groups per workgroup (compile-time), <= 16 thread columns of NC output columns: the column tiles of a row are made equal;
    // at least 64 / CG thread columns where the halo of a row is a full wave's worth of elements (5x5)
    const int CG = fwd ? kDwtCG : dwt_pick_cg(K, C, bf), NC = dwt_nc(K);      // (the forward uses the same column split: 2 columns per thread for 5x5)
    chunks = (int)cdiv(g.cg_total, CG);
    const int nwt = (int)cdiv(W, (256 / CG) * NC);
    int PT = (int)cdiv(cdiv(W, nwt), NC);
    const int pt_min = (int)cdiv(64, CG);                    // at least one whole wave: the halo duty needs 2R*CG <= 64 lanes of it
    if (PT < pt_min) PT = pt_min;
    if (PT * NC < 2 * R) PT = (int)cdiv(2 * R, NC);
    const int PC = PT * NC;
    g.CG = CG; g.PC = PC; threads = CG * PT;
    g.nWT = (int)cdiv(W, PC);
    // strip height: balance whole rounds of resident workgroups against the halo rows
    static const int res = getenv("MNY_DWT_RES") ? atoi(getenv("MNY_DWT_RES")) : 768;
    static const int force_th = getenv("MNY_DWT_TH") ? atoi(getenv("MNY_DWT_TH")) : 0;
    int cap = res / chunks > 0 ? res / chunks : 1;
    if (cap > 8) cap &= ~7;
    double bs = -1.0;
    int bTH = H;
    for (int ns = 1; ns <= H; ++ns) {
        const int TH = (int)cdiv(H, ns);
        if (TH < 2 * R && ns > 1) break;
        const int64_t tiles = (int64_t)N * cdiv(H, TH) * g.nWT;
        const double rounds = (double)tiles / cap;
        const double effr = rounds <= 1.0 ? rounds : rounds / (double)cdiv(tiles, cap);
        const double effh = (double)TH / (TH + R);            // 2R halo rows of G, Y per strip = half the streams
        const double sc2 = effr * effh;
        if (force_th > 0 ? (TH <= force_th && bs < 0) : sc2 > bs) { bs = sc2; bTH = TH; if (force_th > 0) break; }
    }
    g.TH = bTH;
    g.nHS = (int)cdiv(H, g.TH);
    g.ntiles = (int64_t)N * g.nHS * g.nWT;
    g.xcd = 1;
    int64_t want = g.ntiles;
    if (want > 8) want = (want + 7) & ~(int64_t)7;
    gx = (int)(want < cap ? want : cap);
    if ((int64_t)gx > g.ntiles) gx = (int)g.ntiles;
    lds = (size_t)((fwd ? 2 : kDwtConsts + (K == 5 ? K * K : 0)) * CG * CPT + (K + 1) * (PC + 2 * R) * CG * CPT) * sizeof(float);
    return MNY_OK;
}

// Which form runs a K x K stride-1 unit.  5x5: always the tile form (MNY_NO_DWT5=1: none, the entry point reports 5x5 unsupported).
// 3x3: the register form of dwbwd.hip, except on bf16 storage where the tile form measured faster (round 5, MobileNetV3 512x512 bs 64 shapes,
// tools/bench_dwbwd.py): without producer sums C120 0.142 -> 0.080 ms, C160 0.064 -> 0.038, C480 0.138 -> 0.080, C672 0.176 -> 0.134, C960 0.078 ->
// 0.055; with producer sums only the narrow layers win (C120 0.153 -> 0.125, C160 / C184 0.066 -> 0.054; C672 0.179 -> 0.207: the epilogue is vector-ALU
// work the tile form has no idle cycles for).  Channel counts that leave the last workgroup chunk (64 channels) under 3/4 full stay on the register
// form (C72: 0.304 vs 0.332 ms), fp32 storage too (C384 @22x22: 0.221 vs 0.247 ms).  MNY_DWT3=1 / 0 forces the tile / register form for every 3x3 unit.
bool dwt_use(int K, int bf, int red, int C) {
    static const bool no5 = getenv("MNY_NO_DWT5") != nullptr && atoi(getenv("MNY_NO_DWT5")) != 0;
    static const int env3 = getenv("MNY_DWT3") ? atoi(getenv("MNY_DWT3")) : -1;
    if (K == 5) return !no5;
    if (K != 3) return false;
    if (env3 >= 0) return env3 != 0;
    if (!bf) return false;
    if (C == 16 || C == 72) return !red;                    // own chunk widths (dwt_pick_cg): C16 @256x256 0.280 -> 0.192 ms, C72 @128x128 0.313 -> 0.237; with producer sums 0.292 vs 0.294 / 0.319 vs 0.350
    if (C < 120) return false;
    const int cg = C / 4, chunks = (cg + kDwtCG - 1) / kDwtCG;
    if (4 * cg < 3 * chunks * kDwtCG) return false;
    return !red || C <= 192;
}

int dwt_parts(int N, int H, int W, int C, int K, int bf) {
    DwtGeom g; int gx, chunks, threads; size_t lds;
    if (dwt_geom(g, gx, chunks, threads, lds, N, H, W, C, K, bf)) return MNY_EINVAL;
    return gx;
}

template <typename T, int KS>
static int dwt_launch_t(const T* g, const T* y, const float* scale, const float* shift, int act, const float* coef, const T* x, const float* in_scale,
                        const float* in_shift, int in_act, const float* w, const T* addend, T* dx, float* dw, float* ws, int N, int H, int W, int C,
                        hipStream_t st, const float* in_mean, const float* in_invstd, float* in_red) {
    constexpr int CPT = KS == 5 ? 2 : 4, NC = KS == 5 ? 2 : 1;
    constexpr bool TLDS = KS == 5;
    DwtGeom gm; int gx, chunks, threads; size_t lds;
    int rc = dwt_geom(gm, gx, chunks, threads, lds, N, H, W, C, KS, sizeof(T) == 2 ? 1 : 0);
    if (rc) return rc;
    const int am = act == MNY_ACT_NONE ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
    const int xf = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : (in_act == MNY_ACT_RELU6 ? 1 : 3));
    dim3 grid(gx, chunks), block(threads);
    // waves per SIMD the register allocation aims at: 3 (168 VGPRs); the 5x5 form with producer sums needs ~205 and runs spill-free at 2
    // (same box, bf16: C672 @32x32 0.323 vs 0.256 ms, C960 @16x16 0.163 vs 0.107)
#define MNY_DWT_LC(RED_, A_, X_, CG_) do { auto k = dwb_tile_kernel<T, KS, CPT, NC, TLDS, RED_, A_, X_, (MNY_DWT_ALLW2 || (KS == 5 && RED_)) ? 2 : 3, MNY_DWT_PD, CG_>; \
        hipLaunchKernelGGL(k, grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, ws, in_mean, in_invstd, in_red, gm); } while (0)
#define MNY_DWT_L(RED_, A_, X_) do { if constexpr (KS == 3 && sizeof(T) == 2) { if (gm.CG == 4) { MNY_DWT_LC(RED_, A_, X_, 4); break; } if (gm.CG == 18) { MNY_DWT_LC(RED_, A_, X_, 18); break; } } \
        auto k = dwb_tile_kernel<T, KS, CPT, NC, TLDS, RED_, A_, X_, (MNY_DWT_ALLW2 || (KS == 5 && RED_)) ? 2 : 3, MNY_DWT_PD, kDwtCG>; \
        hipLaunchKernelGGL(k, grid, block, lds, st, g, y, scale, shift, act, coef, x, in_scale, in_shift, in_act, w, addend, dx, ws, in_mean, in_invstd, in_red, gm); } while (0)
#define MNY_DWT_S(RED_) switch (am * 4 + xf) { \
        case 0: MNY_DWT_L(RED_, 0, 0); break; case 1: MNY_DWT_L(RED_, 0, 1); break; case 2: MNY_DWT_L(RED_, 0, 2); break; case 3: MNY_DWT_L(RED_, 0, 3); break; \
        case 4: MNY_DWT_L(RED_, 1, 0); break; case 5: MNY_DWT_L(RED_, 1, 1); break; case 6: MNY_DWT_L(RED_, 1, 2); break; case 7: MNY_DWT_L(RED_, 1, 3); break; \
        case 8: MNY_DWT_L(RED_, 2, 0); break; case 9: MNY_DWT_L(RED_, 2, 1); break; case 10: MNY_DWT_L(RED_, 2, 2); break; default: MNY_DWT_L(RED_, 2, 3); break; }
    if (in_red) { MNY_DWT_S(true) } else { MNY_DWT_S(false) }
#undef MNY_DWT_S
#undef MNY_DWT_L
#undef MNY_DWT_LC
    rc = check_launch("dwb_tile_kernel");
    if (rc || !dw) return rc;                       // dw == NULL: partials only (combined later by mny_reduce_batch)
    return launch_reduce_parts(ws, gx, C * KS * KS, dw, st);
}

// bf: 1 = bf16 storage.  Same argument meaning as mny_dw_bnbwd / mny_dw_bnbwd_red (in_red == NULL: no producer sums).
int dwt_launch(int bf, const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef, const void* x,
               const float* in_scale, const float* in_shift, int in_act, const float* w, const void* addend, void* dx, float* dw, float* ws, int N, int H,
               int W, int C, int K, void* stream, const float* in_mean, const float* in_invstd, float* in_red) {
    hipStream_t st = (hipStream_t)stream;
#define MNY_DWT(T_, K_) dwt_launch_t<T_, K_>((const T_*)g, (const T_*)y, scale, shift, act, coef, (const T_*)x, in_scale, in_shift, in_act, w, (const T_*)addend, \
                                             (T_*)dx, dw, ws, N, H, W, C, st, in_mean, in_invstd, in_red)
    if (K == 5) return bf ? MNY_DWT(bf16_t, 5) : MNY_DWT(float, 5);
    return bf ? MNY_DWT(bf16_t, 3) : MNY_DWT(float, 3);
#undef MNY_DWT
}


// ---- forward routing ----------------------------------------------------------------------------------------------------------
// bf16 storage, stride 1, 5x5 only, the same channel-count rule as the backward (>= 120 channels, last 64-channel chunk at least 3/4 full).
// Measured round 5 (tools/bench_dwfwd.py, MobileNetV3 512x512 bs 64 shapes, register form -> tile form): 5x5 C672 @32x32 0.127 -> 0.080 ms,
// C120 @64x64 0.079 -> 0.069, C960 @16x16 0.047 -> 0.044; 3x3 is SLOWER on the tile form (C480 0.042 -> 0.048, C672 0.053 -> 0.062,
// C320 @32x32 0.026 -> 0.037: three loads per row are no burden for the register form, the barrier per row is one for the tile form).
// MNY_DWTF=0 / 1: never / every stride-1 launch (A/B).
bool dwt_fwd_use(int K, int stride, int bf, int C) {
    static const int env = getenv("MNY_DWTF") ? atoi(getenv("MNY_DWTF")) : -1;
    if (stride != 1 || (K != 3 && K != 5) || C % 4 != 0) return false;
    if (env >= 0) return env != 0;
    if (!bf || C < 120 || K != 5) return false;
    const int cg = C / 4, chunks = (cg + kDwtCG - 1) / kDwtCG;
    return 4 * cg >= 3 * chunks * kDwtCG;
}
int dwt_fwd_parts(int N, int H, int W, int C, int K) {
    DwtGeom g; int gx, chunks, threads; size_t lds;
    if (dwt_geom(g, gx, chunks, threads, lds, N, H, W, C, K, 1, true)) return MNY_EINVAL;
    return gx;
}
template <typename T, int KS>
static int dwt_fwd_launch_t(const T* x, const float* in_scale, const float* in_shift, int in_act, const float* w, T* y, float* stats, int N, int H, int W,
                            int C, hipStream_t st) {
    constexpr int NC = KS == 5 ? 2 : 1;
    DwtGeom gm; int gx, chunks, threads; size_t lds;
    int rc = dwt_geom(gm, gx, chunks, threads, lds, N, H, W, C, KS, sizeof(T) == 2 ? 1 : 0, true);
    if (rc) return rc;
    const int xf = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : (in_act == MNY_ACT_RELU6 ? 1 : 3));
    MNY_REQUIRE(in_act != MNY_ACT_HSIGMOID, "dw_fwd (tile form): h-sigmoid views are not supported");
    dim3 grid(gx, chunks), block(threads);
#define MNY_DWF(X_) do { auto k = dwf_tile_kernel<T, KS, NC, X_>; \
        if (lds > 64 * 1024 && !allow_lds((const void*)k, lds)) { set_error("dw_fwd: hipFuncSetAttribute failed"); return MNY_EHIP; } \
        hipLaunchKernelGGL(k, grid, block, lds, st, x, in_scale, in_shift, in_act, w, y, stats, gm); } while (0)
    switch (xf) { case 0: MNY_DWF(0); break; case 1: MNY_DWF(1); break; case 2: MNY_DWF(2); break; default: MNY_DWF(3); break; }
#undef MNY_DWF
    return check_launch("dwf_tile_kernel");
}
int dwt_fwd_launch(int bf, const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w, void* y, float* stats, int N, int H,
                   int W, int C, int K, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (K == 5) return bf ? dwt_fwd_launch_t<bf16_t, 5>((const bf16_t*)x, in_scale, in_shift, in_act, w, (bf16_t*)y, stats, N, H, W, C, st)
                          : dwt_fwd_launch_t<float, 5>((const float*)x, in_scale, in_shift, in_act, w, (float*)y, stats, N, H, W, C, st);
    return bf ? dwt_fwd_launch_t<bf16_t, 3>((const bf16_t*)x, in_scale, in_shift, in_act, w, (bf16_t*)y, stats, N, H, W, C, st)
              : dwt_fwd_launch_t<float, 3>((const float*)x, in_scale, in_shift, in_act, w, (float*)y, stats, N, H, W, C, st);
}

}  // namespace mny
