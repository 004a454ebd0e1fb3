// Pointwise conv with a SHORT reduction feeding a WIDE output (K = 64..96 input channels, N >= K outputs: the expand convs of
// the 14x14 / 22x22 bottlenecks and the data gradients of their project convs) — barrier-free kernel on the bf16 matrix cores.
//
// Why a third GEMM kernel.  On these shapes the LDS-DMA ring kernel (pwgemm.hip) runs at 2.7-3.1 TB/s, half of what the
// tensors cost to stream: a tile is only 4-6 k-stages deep, so a workgroup meets a barrier + a DMA wait every ~200 matrix
// cycles and, in the reduction-epilogue form, adds a load round trip per column block (tools/ab_lib_detail.sh).  Here nothing
// in the loop is shared between waves:
//   * the weight block of the workgroup (K x 32*TNB) is cut ONCE into bf16 planes in LDS, in MFMA-operand order (x6_split,
//     the six-product form of x6.h) — K*BN*6 bytes, 37-74 KB, two workgroups per CU;
//   * a wave owns whole 32-row tiles: its A fragments come STRAIGHT from the row-major tensor (lane (row i, half h) loads the
//     two 16-B chunks 16s+4h.., 16s+8+4h.. of row i: streams at the copy rate, tools/probe/rowfrag_probe.hip), are transformed
//     (BN + activation of the producer unit) and cut once per tile, and the registers are re-requested for the wave's NEXT tile
//     the moment they are cut (rolling prefetch: a full tile of loads is always in flight);
//   * column blocks are the OUTER loop of a tile (u: 32 columns, 6*KS MFMAs, then their epilogue), so stores, epilogue VALU work
//     and the next block's matrix work interleave, and the reduction epilogue's operand (the fed unit's raw output, MFMA
//     layout) is requested one block ahead;
//   * epilogue = the DPP quad transpose of pwgemm.hip (lane jq of a quad: one row, 4 consecutive columns, 16-B streaming store).
// Partial rows ([gx][2][N]) have the layout of the other pointwise kernels.  fp32 storage only.
#include <mutex>
#include <vector>

#include "common.h"
#include "x6.h"

namespace mny {

struct WideArgs {
    const float* A; const float* in_scale; const float* in_shift; int in_act;
    const float* W;                       // [N][K] fp32
    float* C; float* stats;
    int64_t M; int K, N;
    int gx, n_blocks; int64_t ntiles;     // m-runs (= partial rows), column blocks, 32-row tiles
    const float* rY; const float* r_scale; const float* r_shift; const float* r_mean; const float* r_invstd; int r_act;
    const float* addend;
    const float* bias;                    // MODE 4: a per-column constant of the product (mny_pw_lr_fix)
};

// XF: 0 = A as is, 1 = scale/shift + clamp family, 2 = scale/shift + h-swish
// MODE: 0 = plain, 1 = column sums / sums of squares of C, 2 = BN-backward sums of the unit C is the gradient of, 3 = the same over C + addend
//       (MODE 2 / 3 take A as is; XF then names the activation family of THAT unit: 0 = clamp family, 2 = h-swish / h-sigmoid)
//       4 = MODE 3 behind a LINEAR view of A (in_scale / in_shift, no activation) and with a per-column constant `bias` in the product:
//           the low-rank BatchNorm-backward correction dx = view(x) Q + r + addend with the sums over it (mny_pw_lr_fix, csrc/lrbwd.hip)
template <int KS, int TNB, int XF, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_wide_kernel(WideArgs p) {
    constexpr int BN = 32 * TNB;
    constexpr bool TWO_ACC = false;                              // even / odd stages on two accumulators: measured 3-8 % slower (tools/ab/sweep_wide.sh)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sB = smem;                                            // [u][s][piece][64 lanes] 16-B slots
    float* sScale = sB + TNB * KS * 3 * 256;
    float* sShift = sScale + KS * 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, khalf = lane >> 5;
    // block -> (m-run x, column block y): the column blocks of one m-run are consecutive within one XCD (they re-read the same A rows)
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int y = local % p.n_blocks, x = (local / p.n_blocks) * 8 + xcd;
    if (x >= p.gx) return;
    const int n0 = y * BN;
    const int K = p.K, N = p.N;

    // ---- the workgroup's weight block, cut once ----
    for (int c = tid; c < TNB * KS * 64; c += 256) {
        const int sidx = c >> 6, l = c & 63, j = l & 31, h = l >> 5;
        const int u = sidx / KS, s = sidx - u * KS;
        const int n = n0 + u * 32 + j;                           // < N: the host only plans whole column blocks
        const int k0 = s * 16 + 4 * h, k1 = k0 + 8;
        const float* src = p.W + (int64_t)(n < N ? n : 0) * K;
        const float4 v0 = (n < N && k0 < K) ? ld4(src + k0) : f4zero();
        const float4 v1 = (n < N && k1 < K) ? ld4(src + k1) : f4zero();
        bf16x8_t hh, mm, ll;
        x6_split(v4f_t{v0.x, v0.y, v0.z, v0.w}, v4f_t{v1.x, v1.y, v1.z, v1.w}, hh, mm, ll);
        float* dst = sB + (sidx * 3) * 256 + l * 4;
        *reinterpret_cast<v4f_t*>(dst) = __builtin_bit_cast(v4f_t, hh);
        *reinterpret_cast<v4f_t*>(dst + 256) = __builtin_bit_cast(v4f_t, mm);
        *reinterpret_cast<v4f_t*>(dst + 512) = __builtin_bit_cast(v4f_t, ll);
    }
    if ((XF != 0 && MODE < 2) || MODE == 4) {
        const bool has = p.in_scale != nullptr;
        for (int k = tid; k < KS * 16; k += 256) {
            sScale[k] = k < K ? (has ? p.in_scale[k] : 1.f) : 0.f;
            sShift[k] = (k < K && has) ? p.in_shift[k] : 0.f;
        }
    }
    __syncthreads();

    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    const int quad = lrow >> 2, jq = lane & 3;
    const float rslope = act_slope(p.r_act), rhi = act_hi(p.r_act);
    const bool rhsig = p.r_act == MNY_ACT_HSIGMOID;
    float s1[TNB], s2[TNB];
#pragma unroll
    for (int u = 0; u < TNB; ++u) { s1[u] = 0.f; s2[u] = 0.f; }

    const float* bl = sB + lane * 4;                             // the lane's slot in every 1-KiB plane image: constant offsets from one register
    const float* scl = sScale + 4 * khalf;
    const float* shl = sShift + 4 * khalf;
    const int64_t nslots = (int64_t)p.gx * 4;
    const unsigned rowb = (unsigned)N * 4u;                      // bytes per output row
    float4 g[2 * KS];
    float ry[MODE >= 2 ? 16 : 1], ra[MODE >= 3 ? 16 : 1];
    auto load_g = [&](int64_t tile, int s) {
        const int64_t r = tile * 32 + lrow;
        const float* row = p.A + (r < p.M ? r : p.M - 1) * K;
        int k0 = 16 * s + 4 * khalf, k1 = k0 + 8;
        if (s == KS - 1) { k0 = k0 < K ? k0 : 0; k1 = k1 < K ? k1 : 0; }      // past K: any finite filler, the weight planes are zero there
        g[2 * s] = ld4(row + k0); g[2 * s + 1] = ld4(row + k1);
    };
    // operands of the reduction epilogue for column block u of a tile, in the MFMA layout (lane = column, 16 rows): wave-uniform row
    // base + one 32-bit byte offset per load
    auto load_r = [&](int64_t tile, int u) {
        const int col = n0 + u * 32 + lrow, cc = col < N ? col : 0;
        unsigned voff = (unsigned)(4 * khalf) * rowb + (unsigned)cc * 4u;        // the lane's part; the row of register r is a scalar added to the base
        asm volatile("" : "+v"(voff));
        const float* ub = p.rY + tile * 32 * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) ry[r] = ld1(at_bytes(ub + (int64_t)(8 * (r >> 2) + (r & 3)) * N, voff));
        if constexpr (MODE >= 3) {
            const float* ua = p.addend + tile * 32 * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) ra[r] = ld1(at_bytes(ua + (int64_t)(8 * (r >> 2) + (r & 3)) * N, voff));
        }
    };

    // One tile, branch-free (whole 32-row tiles, whole column blocks: the host's rule): gfx9 counts loads and stores in ONE in-order
    // counter, and the compiler can only count what is issued unconditionally — with predicated stores it has to assume none were
    // issued and the wait for the next tile's fragments (older than this tile's stores) degenerates into a wait for the stores.
    auto do_tile = [&](const int64_t tile, const int64_t next) {
        const int64_t wrow = tile * 32;
        // ---- transform + cut the tile's A fragments; each pair of registers is re-requested for the next tile as soon as it is cut ----
        bf16x8_t ah[KS], am[KS], al[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 a0 = g[2 * s], a1 = g[2 * s + 1];
            float z[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            if ((XF != 0 && MODE < 2) || MODE == 4) {
                // RULE (DESIGN 4 "Determinism"): in a kernel that keeps MFMAs in flight the vector ALU consumes an LDS result only
                // behind an EXPLICIT `s_waitcnt lgkmcnt(0)` that carries the destination registers — never behind the compiler's counted
                // lgkmcnt(N > 0) alone (the reduction form's table read, consumed behind a counted wait, returned run-to-run different
                // values on its 6-stage builds; cause unresolved).  The LDS-DMA kernels comply by construction (MNY_LGKM_WAIT), this is
                // the one compiler-visible LDS read next to this kernel's MFMAs.
                v4f_t c0 = *reinterpret_cast<const v4f_t*>(scl + 16 * s), c1 = *reinterpret_cast<const v4f_t*>(scl + 16 * s + 8);
                v4f_t h0 = *reinterpret_cast<const v4f_t*>(shl + 16 * s), h1 = *reinterpret_cast<const v4f_t*>(shl + 16 * s + 8);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(h0), "+v"(h1));
                const float sv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(z[e], sv[e], hv[e]);
                    z[e] = MODE == 4 ? t : (XF == 1 ? fminf(fmaxf(t, slope * t), hi) : t * fminf(fmaxf(t + 3.f, 0.f), 6.f) * (1.f / 6.f));
                }
            }
            x6_split(v4f_t{z[0], z[1], z[2], z[3]}, v4f_t{z[4], z[5], z[6], z[7]}, ah[s], am[s], al[s]);
            // HERE, not earlier: without the fence the compiler hoists every load of the next tile to the top of the loop (two tiles of registers)
            asm volatile("" : "+v"(ah[s]), "+v"(am[s]), "+v"(al[s]) :: "memory");
            load_g(next, s);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* cbase = p.C + wrow * N;                           // wave-uniform
#pragma unroll
        for (int u = 0; u < TNB; ++u) {
            // scale, shift, mean, invstd of the lane's column — from global memory (L1 / L2 hits), requested ahead of the block's matrix work,
            // NOT from an LDS table read next to the MFMAs.  With the table, the 6-stage builds returned BN sums that differed in single
            // elements in 1-15 of 40 runs (tools/ab/stress_red.py; dx always right).  What the ISA showed: the compiler had given the table
            // read the registers an MFMA issued two instructions earlier uses as SrcB, and the vector ALU consumed them right after the
            // (correctly counted) `s_waitcnt lgkmcnt(4)`; the same build with an explicit `lgkmcnt(0)` after the read is bit-stable.  So on
            // gfx950 the LDS counter can run ahead of the register write while an MFMA still in the pipe reads those registers: do not let the
            // vector ALU consume an LDS result that lands in operand registers of a just-issued MFMA.
            float4 rc = f4zero();
            float cbias = 0.f;
            if (MODE >= 2) {
                const int colx = n0 + u * 32 + lrow;
                rc = make_float4(p.r_scale[colx], p.r_shift[colx], p.r_mean[colx], p.r_invstd[colx]);
                if (MODE == 4) cbias = p.bias[colx];
            }
            f32x16 acc, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float* bsrc = bl + ((u * KS + s) * 3) * 256;
                const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc));
                const bf16x8_t bm = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 256));
                const bf16x8_t bl_ = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 512));
                f32x16& a = (TWO_ACC && (s & 1)) ? acc1 : acc;
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, a, 0, 0, 0);      // small terms first
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl_, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[s], bh, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bm, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, a, 0, 0, 0);
            }
            if (TWO_ACC) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
            }
            if (MODE >= 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float zz = fmaf(ry[r], rc.x, rc.y);
                    float dact;
                    if (XF == 0) dact = (zz > 0.f ? 1.f : rslope) * (zz < rhi ? 1.f : 0.f);      // clamp family (torch's subgradient choices, act_bwd)
                    else {                                       // h-swish / h-sigmoid: selects, no per-element branch on the wave-uniform kind
                        const float dsw = zz <= -3.f ? 0.f : (zz >= 3.f ? 1.f : (2.f * zz + 3.f) * (1.f / 6.f));
                        const float dsg = (zz > -3.f && zz < 3.f) ? (1.f / 6.f) : 0.f;
                        dact = rhsig ? dsg : dsw;
                    }
                    if (MODE >= 3) acc[r] += ra[r];              // the sums are over the COMPLETE gradient = product + addend (stored as such)
                    if (MODE == 4) acc[r] += cbias;
                    const float dz = acc[r] * dact;
                    s1[u] += dz; s2[u] = fmaf(dz, (ry[r] - rc.z) * rc.w, s2[u]);
                }
                // the operands of the next column block (of the next tile after the last one): in flight across this block's stores
                // and the next block's matrix work
                asm volatile("" : "+v"(acc) :: "memory");
                if (u + 1 < TNB) load_r(tile, u + 1); else load_r(next, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 1) {
                v2f a1 = v2f{0.f, 0.f}, a2 = v2f{0.f, 0.f};      // packed pairs
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const v2f q = v2f{acc[r], acc[r + 1]};
                    a1 += q;
                    a2 = __builtin_elementwise_fma(q, q, a2);
                }
                s1[u] += a1.x + a1.y; s2[u] += a2.x + a2.y;
            }
            // DPP quad transpose: lane jq of a quad ends up with row 8 gq + 4 khalf + jq, columns 4 quad .. 4 quad + 3 of the block
            unsigned off0 = (unsigned)(4 * khalf + jq) * rowb + (unsigned)(n0 + u * 32 + quad * 4) * 4u;
            asm volatile("" : "+v"(off0));
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float r0 = acc[gq * 4 + 0], r1 = acc[gq * 4 + 1], r2 = acc[gq * 4 + 2], r3 = acc[gq * 4 + 3];
                {
                    const bool odd = lane & 1;
                    const float xa = odd ? r0 : r1, xb = odd ? r2 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0xB1, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0xB1, 0xF, 0xF, true));
                    if (odd) { r0 = ya; r2 = yb; } else { r1 = ya; r3 = yb; }
                }
                {
                    const bool hi2 = lane & 2;
                    const float xa = hi2 ? r0 : r2, xb = hi2 ? r1 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0x4E, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0x4E, 0xF, 0xF, true));
                    if (hi2) { r0 = ya; r1 = yb; } else { r2 = ya; r3 = yb; }
                }
                st4(at_bytes(cbase + (int64_t)(8 * gq) * N, off0), make_float4(r0, r1, r2, r3));      // scalar row step, one lane offset per block
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int64_t tile = (int64_t)x * 4 + wave;
    if (tile < p.ntiles) {
#pragma unroll
        for (int s = 0; s < KS; ++s) load_g(tile, s);
        if (MODE >= 2) load_r(tile, 0);
    }
    // first tile peeled: the loop is then ENTERED with a tile's stores in flight, as on its back edge — merged with a store-free
    // entry the compiler's wait for the fragments would again be the conservative one
    if (tile < p.ntiles) {
        do_tile(tile, tile + nslots < p.ntiles ? tile + nslots : tile);      // (past the end: harmless re-reads)
        tile += nslots;
        for (; tile < p.ntiles; tile += nslots) do_tile(tile, tile + nslots < p.ntiles ? tile + nslots : tile);
    }

    if (MODE != 0) {
        __syncthreads();                                         // every wave is done with the weight planes
        float* red = smem;                                       // [4][BN][2]
#pragma unroll
        for (int u = 0; u < TNB; ++u) {
            const float a = s1[u] + __shfl_xor(s1[u], 32);
            const float b = s2[u] + __shfl_xor(s2[u], 32);
            if (khalf == 0) {
                red[(wave * BN + u * 32 + lrow) * 2 + 0] = a;
                red[(wave * BN + u * 32 + lrow) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
            p.stats[(int64_t)x * 2 * N + n0 + tid] = a;
            p.stats[(int64_t)x * 2 * N + N + n0 + tid] = b;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct WidePlan { int KS, TNB, n_blocks, gx, grid; int64_t ntiles; size_t lds; };

static bool wide_on() {
    static const bool off = getenv("MNY_NO_WIDE") != nullptr || getenv("MNY_GEMM_V1") != nullptr;
    return !off;
}

static int wide_tnb(int /*K*/, int N, bool red) {     // column block = 32 * TNB columns, whole blocks only; 0 = no kernel
    static const int force = getenv("MNY_WIDE_TNB") ? atoi(getenv("MNY_WIDE_TNB")) : 0;
    // (the reduction form takes blocks of at most 96 columns: its 128-column builds need scratch; every instantiation below is spill-free,
    // and tests/test_gpu_kernels.py pins run-to-run determinism of each one the nets use)
    const int cap = red ? 3 : 4;
    if (force >= 2 && force <= cap && N % (32 * force) == 0) return force;
    if (N % 96 == 0 && cap >= 3) return 3;
    if (N % 128 == 0 && cap >= 4) return 4;
    return N % 64 == 0 ? 2 : 0;
}

bool pw_wide_ok(int64_t M, int K, int N, bool red) {
    static const int min_ratio_x2 = getenv("MNY_WIDE_RATIO2") ? atoi(getenv("MNY_WIDE_RATIO2")) : 2;      // N >= ratio/2 * K
    return wide_on() && M >= 8192 && K >= 52 && K <= 96 && (K & 3) == 0 && N >= 64 && wide_tnb(K, N, red) != 0 && 2 * (int64_t)N >= (int64_t)min_ratio_x2 * K;
}

static WidePlan wide_plan(int64_t M, int K, int N, bool red) {
    WidePlan pl;
    pl.KS = (K + 15) / 16;
    pl.TNB = wide_tnb(K, N, red);
    pl.n_blocks = N / (32 * pl.TNB);
    pl.ntiles = M / 32;                                  // whole tiles; the last M % 32 rows go through the LDS-DMA kernel (one more partial row)
    static const int res = getenv("MNY_WIDE_RES") ? atoi(getenv("MNY_WIDE_RES")) : 512;     // resident workgroups: 2 per CU
    int64_t gx = res / pl.n_blocks;
    if (gx >= 8) gx = gx / 8 * 8;                     // whole groups of 8 m-runs: every XCD gets the same number of workgroups (85 runs of 6 blocks
                                                      // put 66 on five XCDs with 64 slots each: K96 N576 0.131 -> 0.120 ms with 80 runs)
    if (gx < 1) gx = 1;
    const int64_t max_gx = cdiv(pl.ntiles, 4);
    if (gx > max_gx) gx = max_gx;
    pl.gx = (int)gx;
    pl.grid = (int)(cdiv(gx, 8) * 8) * pl.n_blocks;
    pl.lds = (size_t)(pl.TNB * pl.KS * 3 * 256 + 2 * pl.KS * 16) * sizeof(float);
    const size_t need = (size_t)4 * 32 * pl.TNB * 2 * sizeof(float);
    if (pl.lds < need) pl.lds = need;
    return pl;
}

int pw_wide_parts(int64_t M, int K, int N, bool red) { return wide_plan(M, K, N, red).gx; }     // of the whole tiles

typedef void (*WideKernel)(WideArgs);
template <int KS, int TNB>
static WideKernel wide_pick_fwd(int xf, int mode) {
    switch (mode * 3 + xf) {
        case 0: return pw_wide_kernel<KS, TNB, 0, 0>; case 1: return pw_wide_kernel<KS, TNB, 1, 0>; case 2: return pw_wide_kernel<KS, TNB, 2, 0>;
        case 3: return pw_wide_kernel<KS, TNB, 0, 1>; case 4: return pw_wide_kernel<KS, TNB, 1, 1>; default: return pw_wide_kernel<KS, TNB, 2, 1>;
    }
}
template <int KS, int TNB>
static WideKernel wide_pick_red(int xf, int mode) {      // clamp-family units only: the h-swish / h-sigmoid builds of this form spill (see wide_tnb)
    if (mode == 4) {
        if constexpr (KS == TNB * 2) return pw_wide_kernel<KS, TNB, 0, 4>;       // (square products only: K = N = 64, 96)
        else return nullptr;
    }
    return mode == 2 ? pw_wide_kernel<KS, TNB, 0, 2> : pw_wide_kernel<KS, TNB, 0, 3>;
}
static WideKernel wide_pick(int KS, int TNB, int xf, int mode) {
    if (mode >= 2) switch (KS * 10 + TNB) {
        case 42: return wide_pick_red<4, 2>(xf, mode); case 43: return wide_pick_red<4, 3>(xf, mode);
        case 52: return wide_pick_red<5, 2>(xf, mode); case 53: return wide_pick_red<5, 3>(xf, mode);
        case 62: return wide_pick_red<6, 2>(xf, mode); default: return wide_pick_red<6, 3>(xf, mode);
    }
    switch (KS * 10 + TNB) {
        case 42: return wide_pick_fwd<4, 2>(xf, mode); case 43: return wide_pick_fwd<4, 3>(xf, mode); case 44: return wide_pick_fwd<4, 4>(xf, mode);
        case 52: return wide_pick_fwd<5, 2>(xf, mode); case 53: return wide_pick_fwd<5, 3>(xf, mode); case 54: return wide_pick_fwd<5, 4>(xf, mode);
        case 62: return wide_pick_fwd<6, 2>(xf, mode); case 63: return wide_pick_fwd<6, 3>(xf, mode); default: return wide_pick_fwd<6, 4>(xf, mode);
    }
}

// the low-rank BN-backward correction (mny_pw_lr_fix): dx = (in_scale o x + in_shift) Q + bias + addend and the sums of the unit dx is the gradient of
bool pw_wide_fix_ok(int64_t M, int K) { return pw_wide_ok(M, K, K, true) && (M & 31) == 0 && (K == 64 || K == 96); }
int pw_wide_fix_parts(int64_t M, int K) { return wide_plan(M, K, K, true).gx; }
int pw_wide_fix_launch(const float* A, const float* in_scale, const float* in_shift, const float* Q, const float* bias, const float* addend, float* C,
                       float* stats, int64_t M, int K, const float* rY, const float* r_scale, const float* r_shift, const float* r_mean,
                       const float* r_invstd, int r_act, hipStream_t st) {
    MNY_REQUIRE(pw_wide_fix_ok(M, K) && rY && addend && bias && stats && r_act < MNY_ACT_HSWISH, "pw_wide_fix: unsupported problem M=%lld K=%d", (long long)M, K);
    const WidePlan pl = wide_plan(M, K, K, true);
    WideArgs a{A, in_scale, in_shift, MNY_ACT_NONE, Q, C, stats, M, K, K, pl.gx, pl.n_blocks, pl.ntiles, rY, r_scale, r_shift, r_mean, r_invstd, r_act, addend, bias};
    const WideKernel k = wide_pick(pl.KS, pl.TNB, 0, 4);
    MNY_REQUIRE(k != nullptr, "pw_wide_fix: no kernel for K=%d", K);
    if (!allow_lds((const void*)k, 80 * 1024)) {
        set_error("pw_wide: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    hipLaunchKernelGGL(k, dim3(pl.grid), dim3(256), pl.lds, st, a);
    return check_launch("pw_wide_kernel<fix>");
}

// mode 0/1: forward (stats != null -> 1); mode 2/3: data gradient + BN-backward sums (addend != null -> 3)
int pw_wide_launch(const float* A, const float* in_scale, const float* in_shift, int in_act, const float* W, float* C, float* stats,
                   int64_t M, int K, int N, const float* rY, const float* r_scale, const float* r_shift, const float* r_mean,
                   const float* r_invstd, int r_act, const float* addend, hipStream_t st) {
    MNY_REQUIRE(pw_wide_ok(M, K, N, rY != nullptr) && (M & 31) == 0, "pw_wide: unsupported problem M=%lld K=%d N=%d (whole 32-row tiles: callers send the last M %% 32 rows elsewhere)", (long long)M, K, N);
    const WidePlan pl = wide_plan(M, K, N, rY != nullptr);
    MNY_REQUIRE(pl.KS >= 4 && pl.KS <= 6 && pl.TNB >= 2 && pl.TNB <= 4, "pw_wide: no kernel for K=%d, column block %d", K, 32 * pl.TNB);
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    const int mode = rY ? (addend ? 3 : 2) : (stats ? 1 : 0);
    const int XF = rY ? (r_act >= MNY_ACT_HSWISH ? 2 : 0) : (!xf ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1));
    MNY_REQUIRE(!(rY && xf), "pw_wide: the reduction form takes its A operand as is");
    MNY_REQUIRE(!rY || r_act < MNY_ACT_HSWISH, "pw_wide: the reduction form knows the clamp family only (mny_pw_dgrad_bnred_supported)");
    MNY_REQUIRE(rY || !addend, "pw_wide: a plain addend is not supported (callers route it to the LDS-DMA kernel)");
    MNY_REQUIRE(in_act != MNY_ACT_HSIGMOID, "pw_wide: h-sigmoid input transform is not supported");
    WideArgs a{A, in_scale, in_shift, in_act, W, C, stats, M, K, N, pl.gx, pl.n_blocks, pl.ntiles, rY, r_scale, r_shift, r_mean, r_invstd, r_act, addend, nullptr};
    const WideKernel k = wide_pick(pl.KS, pl.TNB, XF, mode);
    if (!allow_lds((const void*)k, 80 * 1024)) {      // > 64 KB of dynamic LDS needs an explicit opt-in per kernel (and device)
        set_error("pw_wide: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    hipLaunchKernelGGL(k, dim3(pl.grid), dim3(256), pl.lds, st, a);
    return check_launch("pw_wide_kernel");
}

}  // namespace mny
