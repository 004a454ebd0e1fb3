// Depthwise 3x3 forward (stride 1 | 2), fp32, input rows staged through LDS by DMA.
//
// dw_slide_kernel (dwconv.hip) loads every tap straight from global memory into registers: three loads per thread and row,
// the three column taps of neighbouring threads overlapping, and no way to run further ahead than the registers allow — it
// sits at ~4.0 TB/s algorithmic with ~1.5-2x read amplification at the L2->fabric counters.  Here a workgroup owns a span of PW
// output columns x CG channel groups and walks DOWN the rows of a strip:
//   * each input row of the span (IW = (PW-1)*S + 3 pixels x CG float4) is copied HBM -> LDS once, by global_load_lds_dwordx4
//     (the [pixel][channel-group] image is lane-linear = what a DMA writes); a ring of NS rows keeps NS-1 rows in flight per
//     workgroup with no VGPR cost — counted `s_waitcnt vmcnt` + raw `s_barrier`, as in the GEMM kernels;
//   * a thread (channel group, output column) keeps the usual 3x3 register window and reads only the NEW row's three taps from
//     LDS (raw ds_read_b128: a compiler-visible LDS read after a DMA would drain the queue), applies the producer's
//     BN-apply + activation, and emits one output pixel per S input rows; BN statistics ride in the epilogue as before.
// Column halo is paid once per workgroup ((PW*S + 2) / (PW*S) of the row), not once per thread.
//
// replaces nn.Conv2d(groups=C, k=3) forward at models/mobilenetv2.py:65,79 and models/mbv2_yolo.py:22.
#include "common.h"

namespace mny {

struct DwsGeom {
    int N, H, W, C, Ho, Wo;
    int CG, PW, IW;            // channel groups / output columns per workgroup, input pixels per staged row
    int TH, nHS, nWS;          // output rows per strip, strips per column span, column spans per image row
    int64_t nwork;             // N * nWS * nHS
    int cg_total;
};

template <int S, int XF>
__global__ __launch_bounds__(256) void dw_fwd_dma_kernel(const float* __restrict__ x, const float* __restrict__ in_scale,
                                                         const float* __restrict__ in_shift, int in_act, const float* __restrict__ w,
                                                         float* __restrict__ y, float* __restrict__ parts, DwsGeom g) {
    constexpr int NS = 6;                          // ring slots = staged input rows
    constexpr int LPW = S == 1 ? 2 : 3;            // DMA instructions per wave and row (a row is <= 5 / 9 KiB)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RQ = g.IW * g.CG;                    // float4 per staged row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgl = tid % g.CG, pw = tid / g.CG;
    const int cg0 = blockIdx.y * g.CG;
    const bool cvalid = cg0 + cgl < g.cg_total && pw < g.PW;
    const int c = (cg0 + (cg0 + cgl < g.cg_total ? cgl : 0)) * 4;

    // DMA descriptors.  A DMA instruction writes LANE-LINEARLY from its wave-uniform LDS base (lane l -> base + 16 l, whatever
    // the lane's own address says), so instruction j always fills float4 slots 64j .. 64j+63 of the ring slot: the slot stride is
    // rounded up to whole instructions (SQ), lanes past the row image fetch a valid filler, and the surplus instructions a wave
    // issues to keep the per-wave count fixed (vmcnt bookkeeping) are exact duplicates of instruction j - NI.
    const int NI = (RQ + 63) >> 6, SQ = NI * 64;
    int d_pix[LPW], d_coff[LPW], d_lds[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wv + 4 * i;
        while (j >= NI) j -= NI;
        int q = j * 64 + lane;
        d_lds[i] = q * 4;
        if (q >= RQ) q = RQ - 1;
        const int pix = q / g.CG, cq = q % g.CG;
        d_pix[i] = pix;
        d_coff[i] = (cg0 + (cg0 + cq < g.cg_total ? cq : 0)) * 4;
    }

    float4 wg[9];
    float4 sc = f4one(), sh = f4zero();
    if (cvalid) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wg[t] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
        if (XF != 0 && in_scale) { sc = ld4(in_scale + c); sh = ld4(in_shift + c); }
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) wg[t] = f4zero();
    }
    const float slope = act_slope(in_act), hi_clip = act_hi(in_act);
    float4 acc_s1 = f4zero(), acc_s2 = f4zero();

    for (int64_t work = blockIdx.x; work < g.nwork; work += gridDim.x) {
        const int ws = (int)(work % g.nWS);
        const int hs = (int)((work / g.nWS) % g.nHS);
        const int n = (int)(work / ((int64_t)g.nWS * g.nHS));
        const int wo0 = ws * g.PW, ho0 = hs * g.TH;
        const int ho1 = min(ho0 + g.TH, g.Ho);
        const int nrows = (ho1 - ho0 - 1) * S + 3;         // input rows ho0*S-1 .. (ho1-1)*S+1
        const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;
        const float* xn = x + (int64_t)n * g.H * g.W * g.C;

        auto issue = [&](int t) {
            float* stage = smem + (t % NS) * (SQ * 4);
            const int hc = min(max(hi0 + t, 0), g.H - 1);
            const float* rowp = xn + (int64_t)hc * g.W * g.C;
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                const int wc = min(max(wi0 + d_pix[i], 0), g.W - 1);
                const float* src = rowp + (int64_t)wc * g.C + d_coff[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
            }
        };

        float4 win[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) win[r][q] = f4zero();
        const int wo = wo0 + pw;
        const bool ovalid = cvalid && wo < g.Wo;

        auto consume = [&](int t) {
            const float* stage = smem + (t % NS) * (SQ * 4);
            const int hi = hi0 + t;
            const bool rok = hi >= 0 && hi < g.H;
            v4f_t tap[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) tap[q] = lds_read_f4(stage + ((min(pw, g.PW - 1) * S + q) * g.CG + cgl) * 4);
            MNY_LGKM_WAIT(tap[0]);
            MNY_LGKM_DEP(tap[1]);
            MNY_LGKM_DEP(tap[2]);
#pragma unroll
            for (int q = 0; q < 3; ++q) { win[0][q] = win[1][q]; win[1][q] = win[2][q]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float4 v = make_float4(tap[q].x, tap[q].y, tap[q].z, tap[q].w);
                if (XF != 0) {
                    float z[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        z[e] = XF == 1 ? fminf(fmaxf(z[e], slope * z[e]), hi_clip) : z[e] * fminf(fmaxf(z[e] + 3.f, 0.f), 6.f) / 6.f;
                    v = make_float4(z[0], z[1], z[2], z[3]);
                }
                const int wi = wo * S - 1 + q;
                win[2][q] = (rok && wi >= 0 && wi < g.W) ? v : f4zero();     // padding taps are 0 in the ACTIVATED domain
            }
            if (t >= 2 && (t - 2) % S == 0 && ovalid) {
                const int ho = ho0 + (t - 2) / S;
                float4 out = f4zero();
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 3; ++q) fma4(out, win[r][q], wg[r * 3 + q]);
                st4(y + (((int64_t)n * g.Ho + ho) * g.Wo + wo) * g.C + c, out);
                add4(acc_s1, out);
                fma4(acc_s2, out, out);
            }
        };

        const int pre = nrows < NS - 1 ? nrows : NS - 1;
        for (int t = 0; t < pre; ++t) issue(t);
        const int steady = nrows - pre;
        for (int t = 0; t < steady; ++t) {
            wait_vmcnt<LPW*(NS - 2)>();                      // my share of the oldest staged row has landed
            __builtin_amdgcn_s_barrier();                    // ... for every wave; the slot refilled below is fully consumed
            issue(t + pre);
            consume(t);
        }
        for (int t = steady; t < nrows; ++t) {
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            consume(t);
        }
        __builtin_amdgcn_s_barrier();                        // the ring may be refilled by the next work item
    }

    if (parts == nullptr) return;
    // deterministic block reduction over the PW column slots, fixed order
    wait_vmcnt<0>();
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid * 2 + 0] = acc_s1;
    red[tid * 2 + 1] = acc_s2;
    __syncthreads();
    if (pw == 0 && cg0 + cgl < g.cg_total) {
        float4 a = f4zero(), b = f4zero();
        const int ppb = 256 / g.CG < g.PW ? 256 / g.CG : g.PW;
        for (int p = 0; p < ppb; ++p) {
            if ((p * g.CG + cgl) < (int)blockDim.x) { add4(a, red[(p * g.CG + cgl) * 2]); add4(b, red[(p * g.CG + cgl) * 2 + 1]); }
        }
        float* dst = parts + (int64_t)blockIdx.x * 2 * g.C;
        st4(dst + c, a);
        st4(dst + g.C + c, b);
    }
}

static int dws_geom(DwsGeom& g, dim3& grid, int& threads, size_t& lds, int N, int H, int W, int C, int stride) {
    g.N = N; g.H = H; g.W = W; g.C = C;
    g.Ho = (H + 2 - 3) / stride + 1;
    g.Wo = (W + 2 - 3) / stride + 1;
    g.cg_total = C / 4;
    const int chunks = (int)cdiv(g.cg_total, 32);            // <= 32 channel groups (128 channels) per workgroup
    g.CG = (int)cdiv(g.cg_total, chunks);
    g.PW = 256 / g.CG;
    if (g.PW > g.Wo) g.PW = g.Wo;
    threads = g.CG * g.PW;
    threads = (int)cdiv(threads, 64) * 64;                   // whole waves: every wave takes part in the DMA + barriers
    g.IW = (g.PW - 1) * stride + 3;
    const int ns = (int)cdiv(g.Ho, 32);
    g.TH = (int)cdiv(g.Ho, ns);
    g.nHS = (int)cdiv(g.Ho, g.TH);
    g.nWS = (int)cdiv(g.Wo, g.PW);
    g.nwork = (int64_t)N * g.nWS * g.nHS;
    const int cap = kMaxParts / chunks > 0 ? kMaxParts / chunks : 1;
    grid = dim3((unsigned)(g.nwork < cap ? g.nwork : cap), chunks);
    lds = (size_t)6 * cdiv((int64_t)g.IW * g.CG, 64) * 64 * 16;      // ring slots rounded up to whole 1-KiB DMA instructions
    if (lds < (size_t)threads * 2 * 16) lds = (size_t)threads * 2 * 16;
    return MNY_OK;
}

// eligibility of the staged kernel: fp32, 3x3, a row image of at most 8 / 12 KiB-instructions, enough columns to amortise the halo
bool dws_supported(int N, int H, int W, int C, int K, int stride) {
    if (K != 3 || (stride != 1 && stride != 2) || C % 4 || N <= 0) return false;
    DwsGeom g; dim3 grid; int threads; size_t lds;
    dws_geom(g, grid, threads, lds, N, H, W, C, stride);
    const int instrs = (int)cdiv((int64_t)g.IW * g.CG, 64);
    return g.PW >= 4 && instrs <= (stride == 1 ? 8 : 12) && threads == 256 && lds <= 60 * 1024;
}

int dws_parts(int N, int H, int W, int C, int stride) {
    DwsGeom g; dim3 grid; int threads; size_t lds;
    dws_geom(g, grid, threads, lds, N, H, W, C, stride);
    return (int)grid.x;
}

int dws_launch(const float* x, const float* sc, const float* sh, int act, const float* w, float* y, float* parts, int N, int H, int W,
               int C, int stride, hipStream_t st) {
    DwsGeom g; dim3 grid; int threads; size_t lds;
    dws_geom(g, grid, threads, lds, N, H, W, C, stride);
    const int xf = (sc == nullptr && act == MNY_ACT_NONE) ? 0 : (act == MNY_ACT_HSWISH ? 2 : 1);
#define MNY_L(S_, X_) hipLaunchKernelGGL((dw_fwd_dma_kernel<S_, X_>), grid, dim3(threads), lds, st, x, sc, sh, act, w, y, parts, g)
    switch (stride * 10 + xf) {
        case 10: MNY_L(1, 0); break; case 11: MNY_L(1, 1); break; case 12: MNY_L(1, 2); break;
        case 20: MNY_L(2, 0); break; case 21: MNY_L(2, 1); break; default: MNY_L(2, 2); break;
    }
#undef MNY_L
    return check_launch("dw_fwd_dma_kernel");
}

}  // namespace mny
