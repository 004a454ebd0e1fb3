// Weight gradient of a pointwise conv with one NARROW side (64 ... 320 channels, a multiple of 32) and a wide one (>= 2x) — the expand /
// project convs of the 22x22 and 11x11 bottlenecks: dW[N][K] = sum_m dY[m][:]^T act(X[m][:]) — as a barrier-free stream kernel on the
// bf16 matrix cores.
//
// The LDS-DMA weight-gradient kernels (pwgemm.hip) run these shapes at 70-79 TFLOP/s, 1.7-2.8 TB/s: a 16-row chunk behind a barrier,
// fp32 MFMAs (the six-product form needs whole 16-row chunks per wave, the all-waves-share-the-tile mode hands a wave 4 rows).  Here a
// wave owns whole 16-row chunks of its workgroup's row slice (chunks wave, wave + 4, ...), loads both operands STRAIGHT from the
// row-major tensors in the reduction-major pattern (lane (column c, half h) -> rows 2e + h, e = 0..7, of column c: eight dword loads
// per 32-column block, each instruction = two full 128-B row segments; tools/probe/rowfrag_probe.hip: 5.3-6.3 TB/s), cuts them into
// three bf16 pieces (x6_split) and re-requests the registers for its NEXT chunk as soon as they are cut.  A workgroup = the narrow
// side (TB or TA blocks of 32) x a 64-column slice of the wide side; its four waves' accumulators are combined through LDS once, at
// the end, into one partial row of [splits][N][K] (the layout of the other weight-gradient kernels; combined by the same reducers).
// fp32 storage, M % 16 == 0; no loop branch around a load, no store in the loop.
#include "common.h"
#include "x6.h"

namespace mny {

struct WgsArgs {
    const float* X; const float* in_scale; const float* in_shift; int in_act;
    const float* dY; float* partial;
    int64_t M; int K, N;
    int slices_a, slices_b;               // slices of the dY columns (32*TA each) and of the X columns (32*TB each): one of them is 1
    int64_t rows_per_block;               // multiple of 64
    int splits, xcd_groups;
};

template <int TA, int TB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_wgrad_stream_kernel(WgsArgs p) {
    __shared__ __attribute__((aligned(16))) float red[3 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    // block -> (slice, row split z): the slices of one row split are consecutive within one XCD — they re-read the same rows of the narrow
    // operand, which then comes from that XCD's L2 (dispatched round-robin over the XCDs, 9 slices re-read it from HBM: 1.5x counted traffic)
    // (p.xcd_groups == 0: few row splits — many slices, small M, everything L2-resident anyway —: plain order, slice fastest)
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int nsl = p.slices_a * p.slices_b;
    const int slice = p.xcd_groups ? local % nsl : bid % nsl, z = p.xcd_groups ? (local / nsl) * 8 + xcd : bid / nsl;
    if (z >= p.splits) return;
    const int sa = slice % p.slices_a, sb = slice / p.slices_a;
    const int ca0 = sa * 32 * TA, cb0 = sb * 32 * TB;
    const int K = p.K, N = p.N;
    const int64_t m_begin = (int64_t)z * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const int nchunks = (int)((m_end - m_begin) / 16);           // whole chunks: M % 16 == 0, rows_per_block % 64 == 0
    const bool has_xf = p.in_scale != nullptr;
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);

    float sc[TB], sh[TB];
    unsigned offa[TA], offb[TB];                                 // the lane's byte offset inside a chunk (row kk, its column); rows 2e are scalar steps
#pragma unroll
    for (int i = 0; i < TA; ++i) {
        const int co = ca0 + 32 * i + li;
        offa[i] = ((unsigned)kk * (unsigned)N + (unsigned)(co < N ? co : 0)) * 4u;      // past N: any finite column, the result is not stored
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int ci = cb0 + 32 * j + li;
        offb[j] = ((unsigned)kk * (unsigned)K + (unsigned)(ci < K ? ci : 0)) * 4u;
        sc[j] = (has_xf && ci < K) ? p.in_scale[ci] : 1.f;
        sh[j] = (has_xf && ci < K) ? p.in_shift[ci] : 0.f;
    }

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[TA][8], rb[TB][8];
    auto load_a = [&](int c, int i) {                            // chunk c of this workgroup's slice, column block i of dY
        const float* base = p.dY + (m_begin + 16 * (int64_t)c) * N;
        unsigned off = offa[i];
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[i][e] = ld1(at_bytes(base + (int64_t)(2 * e) * N, off));
    };
    auto load_b = [&](int c, int j) {
        const float* base = p.X + (m_begin + 16 * (int64_t)c) * K;
        unsigned off = offb[j];
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[j][e] = ld1(at_bytes(base + (int64_t)(2 * e) * K, off));
    };

    int c = wave;
    if (c < nchunks) {
#pragma unroll
        for (int i = 0; i < TA; ++i) load_a(c, i);
#pragma unroll
        for (int j = 0; j < TB; ++j) load_b(c, j);
    }
    for (; c < nchunks; c += 4) {
        const int next = c + 4 < nchunks ? c + 4 : c;            // past the end: harmless re-reads
        bf16x8_t ah[TA], am[TA], al[TA];
#pragma unroll
        for (int i = 0; i < TA; ++i) {
            x6_split(v4f_t{ra[i][0], ra[i][1], ra[i][2], ra[i][3]}, v4f_t{ra[i][4], ra[i][5], ra[i][6], ra[i][7]}, ah[i], am[i], al[i]);
            // HERE, not earlier: without the fence every load of the next chunk is hoisted to the top of the loop (two chunks of registers)
            asm volatile("" : "+v"(ah[i]), "+v"(am[i]), "+v"(al[i]) :: "memory");
            load_a(next, i);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            float z[8];
            if (p.in_act >= MNY_ACT_HSWISH) {                    // wave-uniform: h-swish / h-sigmoid views (MobileNetV3) — ALU only inside the branch
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = act_fwd(fmaf(rb[j][e], sc[j], sh[j]), p.in_act);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(rb[j][e], sc[j], sh[j]);
                    z[e] = fminf(fmaxf(t, slope * t), hi);
                }
            }
            bf16x8_t bh, bm, bl;
            x6_split(v4f_t{z[0], z[1], z[2], z[3]}, v4f_t{z[4], z[5], z[6], z[7]}, bh, bm, bl);
            asm volatile("" : "+v"(bh), "+v"(bm), "+v"(bl) :: "memory");
            load_b(next, j);
#pragma unroll
            for (int i = 0; i < TA; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);      // small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // the four waves' accumulators -> one partial row (fixed order: wave 0 + 1 + 2 + 3)
    float* dst = p.partial + (int64_t)z * N * K;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[i][j][r];
            }
            __syncthreads();
            if (wave == 0) {
                const int ci = cb0 + 32 * j + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ((acc[i][j][r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                    const int co = ca0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < N && ci < K) dst[(int64_t)co * K + ci] = v;
                }
            }
        }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct WgsPlan { int TA, TB, slices_a, slices_b, splits, xcd_groups; int64_t rows_per_block; };

bool pw_wgs_ok(int64_t M, int K, int N) {
    static const bool off = getenv("MNY_NO_WGS") != nullptr || getenv("MNY_WGRAD_V1") != nullptr;
    const int thin = K < N ? K : N, wide = K < N ? N : K;
    static const int max_thin = getenv("MNY_WGS_MAXTHIN") ? atoi(getenv("MNY_WGS_MAXTHIN")) : 320;      // same-box A/B: 160 and 320 gain 20-25 %, 512 loses 5 %
    return !off && M >= 16384 && (M & 15) == 0 && thin >= 64 && thin <= max_thin && (thin & 31) == 0 && wide >= 2 * thin && (wide & 3) == 0;
}

static WgsPlan wgs_plan(int64_t M, int K, int N) {
    WgsPlan pl;
    // slices of the wide side: 96 columns next to a 64-channel narrow side when they divide it (6 accumulator tiles either way; the cuts —
    // the vector-ALU cost — are shared by more matrix work: 5 cuts per 6 tiles instead of 4 per 4), else 64
    static const int force_ws = getenv("MNY_WGS_SLICE") ? atoi(getenv("MNY_WGS_SLICE")) : 0;
    const int thin = K <= N ? K : N, wide = K <= N ? N : K;
    // a narrow side of more than 96 channels is taken in parts of 96 (the last one partly masked): 2-D tiling, both operands re-read from L2
    const int tblocks = thin / 32, tpart = tblocks <= 3 ? tblocks : ((tblocks % 2 == 0 && tblocks % 3 != 0) ? 2 : 3), tslices = (int)cdiv(tblocks, tpart);
    int ws = (tpart == 2 && wide % 96 == 0) ? 3 : 2;
    if (force_ws == 2 || (force_ws == 3 && tpart == 2)) ws = force_ws;
    if (K <= N) { pl.TB = tpart; pl.TA = ws; pl.slices_a = (int)cdiv(N, 32 * ws); pl.slices_b = tslices; }      // X narrow: slices of the dY columns
    else { pl.TA = tpart; pl.TB = ws; pl.slices_a = tslices; pl.slices_b = (int)cdiv(K, 32 * ws); }           // dY narrow: slices of the X columns
    static const int blocks = getenv("MNY_WGS_BLOCKS") ? atoi(getenv("MNY_WGS_BLOCKS")) : 512;      // two workgroups per CU
    const int slices = pl.slices_a * pl.slices_b;
    int64_t splits = blocks / slices;
    pl.xcd_groups = splits >= 32;                     // same-box A/B: with fewer row splits the grouping only costs parallelism (320->1280: 5 splits on 5 XCDs)
    if (pl.xcd_groups) splits = splits / 8 * 8;       // whole groups of 8 row splits: the same number of workgroups on every XCD
    if (splits < 1) splits = 1;
    const int64_t max_splits = cdiv(M, 256);
    if (splits > max_splits) splits = max_splits;
    pl.rows_per_block = cdiv(cdiv(M, splits), 64) * 64;
    pl.splits = (int)cdiv(M, pl.rows_per_block);
    return pl;
}

int pw_wgs_splits(int64_t M, int K, int N) { return wgs_plan(M, K, N).splits; }

int pw_wgs_launch(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* partial,
                  int64_t M, int K, int N, hipStream_t st) {
    MNY_REQUIRE(pw_wgs_ok(M, K, N), "pw_wgs: unsupported problem M=%lld K=%d N=%d", (long long)M, K, N);
    const WgsPlan pl = wgs_plan(M, K, N);
    WgsArgs a{x, in_scale, in_shift, in_act, dy, partial, M, K, N, pl.slices_a, pl.slices_b, pl.rows_per_block, pl.splits, pl.xcd_groups};
    const dim3 grid((unsigned)(pl.slices_a * pl.slices_b * (pl.xcd_groups ? cdiv(pl.splits, 8) * 8 : pl.splits))), block(256);
    if (pl.TA == 2 && pl.TB == 2) hipLaunchKernelGGL((pw_wgrad_stream_kernel<2, 2>), grid, block, 0, st, a);
    else if (pl.TA == 2 && pl.TB == 3) hipLaunchKernelGGL((pw_wgrad_stream_kernel<2, 3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((pw_wgrad_stream_kernel<3, 2>), grid, block, 0, st, a);
    return check_launch("pw_wgrad_stream_kernel");
}

}  // namespace mny
