// Pointwise (1x1) convolution as fp32 GEMMs on the gfx950 matrix cores.
//
//   forward / data-gradient : C[M,N] = act_in(A)[M,K] * B[N,K]^T (+bias)(+addend)      ("NT")
//   weight-gradient         : dW[N,K] = dY[M,N]^T * act_in(X)[M,K]                     ("TN", reduction over M)
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak = 64 cyc per instruction per SIMD).
// One MFMA consumes a single A and B dword per lane, so LDS/L1 bandwidth is never the limiter in fp32:
// the kernels are kept simple (one barrier per K tile) and spend their effort on (i) fusing the
// producer's BatchNorm-apply + activation into the A load, (ii) fusing the BatchNorm statistics of
// the output into the epilogue, and (iii) exposing enough independent workgroups (>> 256 CUs).
//
// NT kernel: 256 threads = 4 waves stacked along M (32 rows each, BM = 128), every wave spans the
// whole BN = 32*TN tile (TN = 1..4 accumulators of 32x32).  A and B tiles are staged
// global -> registers (float4, transform applied) -> LDS with a row pitch of BK+4 floats, which makes
// the ds_read_b128 fragment reads bank-conflict free (pitch 20 -> 16 distinct 4-bank slots per lane group).
// Fragment trick: lanes with k-half h read the float4 at k = 8*kc + 4*h; MFMA j of the chunk then
// contracts k in {8kc+j, 8kc+4+j} — any consistent permutation of K is a valid contraction order.
//
// TN kernel: operands are M-major, which is exactly the MFMA operand layout (lane = channel, k = row),
// so fragments are loaded straight from global memory (128-B contiguous per half-wave), no LDS.
//
// replaces nn.Conv2d(Cin,Cout,1) at models/mobilenetv2.py:48,69,75,83 and models/mbv2_yolo.py:20,82
// and their autograd backward (convolution_backward = 57.5 % of the reference's CPU step).
#include <stdlib.h>

#include <map>
#include <type_traits>
#include <mutex>

#include "common.h"
#include "x6.h"

namespace mny {

constexpr int BM = 128;

struct GemmArgs {   // A / addend / C are T* (float or bf16_t) of the kernel instantiation, B is T* as well
    const void* A; const float* in_scale; const float* in_shift; int in_act;
    const void* B; const float* bias; const void* addend; void* C; float* stats;
    int64_t M; int K; int N;
    int m_tiles; int tiles_per_block;
};

template <typename T, int TN, int BK>
__global__ __launch_bounds__(256) void pw_gemm_nt_kernel(GemmArgs p) {
    constexpr int BN = 32 * TN;
    const T* pA = (const T*)p.A;
    const T* pAdd = (const T*)p.addend;
    T* pC = (T*)p.C;
    constexpr int LDP = BK + 4;   // LDS row pitch (floats): 20 / 36 -> conflict-free ds_read_b128 fragments
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                           // [2][BM][LDP]
    float* Bs = smem + 2 * BM * LDP;            // [2][BN][LDP]
    float* sScale = Bs + 2 * BN * LDP;          // [Kpad]
    const int Kpad = (p.K + BK - 1) / BK * BK;
    float* sShift = sScale + Kpad;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int lrow = lane & 31;
    const int khalf = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const bool has_xf = p.in_scale != nullptr;
    const bool do_xf = has_xf || p.in_act != MNY_ACT_NONE;
    const bool kvec = (p.K & 3) == 0;   // rows 16-B aligned -> float4 loads; else scalar tail-safe loads

    if (do_xf) {                                  // (the host sizes the LDS without this area when !do_xf)
        for (int k = tid; k < Kpad; k += 256) {
            sScale[k] = (has_xf && k < p.K) ? p.in_scale[k] : 1.f;
            sShift[k] = (has_xf && k < p.K) ? p.in_shift[k] : 0.f;
        }
        __syncthreads();
    }

    // staging assignment: A tile = BM*BK/4 = 512 float4 -> 2 per thread; B tile = BN*BK/4 -> TN/2 per thread
    constexpr int A_PER = BM * BK / 4 / 256;
    constexpr int B_F4 = BN * BK / 4;
    constexpr int B_PER = (B_F4 + 255) / 256;
    const int nk = Kpad / BK;

    float s1[TN], s2[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

    const int mt_begin = blockIdx.x * p.tiles_per_block;
    const int mt_end = min(mt_begin + p.tiles_per_block, p.m_tiles);

    for (int mt = mt_begin; mt < mt_end; ++mt) {
        const int64_t m0 = (int64_t)mt * BM;
        f32x16 acc[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        float4 ra[A_PER], rb[B_PER];
        auto gload = [&](int kt) {
            const int k0 = kt * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BK / 4), kq = idx % (BK / 4);
                const int64_t m = m0 + row;
                const int k = k0 + kq * 4;
                float4 v = f4zero();
                if (m < p.M && k < p.K) {
                    const T* src = pA + m * p.K + k;
                    if (kvec) v = ld4(src);
                    else { v.x = ld1(src); if (k + 1 < p.K) v.y = ld1(src + 1); if (k + 2 < p.K) v.z = ld1(src + 2); if (k + 3 < p.K) v.w = ld1(src + 3); }
                    if (do_xf) {
                        v = xform4(v, ld4(sScale + k), ld4(sShift + k), p.in_act);
                        if (!kvec) { if (k + 1 >= p.K) v.y = 0.f; if (k + 2 >= p.K) v.z = 0.f; if (k + 3 >= p.K) v.w = 0.f; }
                    }
                }
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int idx = tid + i * 256;
                float4 v = f4zero();
                if (idx < B_F4) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
                    const int n = n0 + row;
                    const int k = k0 + kq * 4;
                    if (n < p.N && k < p.K) {
                        const T* src = (const T*)p.B + (int64_t)n * p.K + k;
                        if (kvec) v = ld4(src);
                        else { v.x = ld1(src); if (k + 1 < p.K) v.y = ld1(src + 1); if (k + 2 < p.K) v.z = ld1(src + 2); if (k + 3 < p.K) v.w = ld1(src + 3); }
                    }
                }
                rb[i] = v;
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BK / 4), kq = idx % (BK / 4);
                st4(As + (buf * BM + row) * LDP + kq * 4, ra[i]);
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int idx = tid + i * 256;
                if (idx < B_F4) {
                    const int row = idx / (BK / 4), kq = idx % (BK / 4);
                    st4(Bs + (buf * BN + row) * LDP + kq * 4, rb[i]);
                }
            }
        };

        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gload(kt + 1);
            const float* a_base = As + (buf * BM + wave * 32 + lrow) * LDP + khalf * 4;
            const float* b_base = Bs + (buf * BN + lrow) * LDP + khalf * 4;
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const float4 af = ld4(a_base + kc * 8);
                float4 bf[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) bf[t] = ld4(b_base + t * 32 * LDP + kc * 8);
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[t].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[t].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[t].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[t].w, acc[t], 0, 0, 0);
                }
            }
            if (kt + 1 < nk) lstore(buf ^ 1);
            __syncthreads();
        }

        // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = n0 + t * 32 + lrow;
            const bool cok = col < p.N;
            const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                float v = acc[t][r] + bv;
                if (cok && row < p.M) {
                    if (pAdd) v += ld1(pAdd + row * p.N + col);
                    st1(pC + row * p.N + col, v);
                    v = stored<T>(v);
                    s1[t] += v;
                    s2[t] = fmaf(v, v, s2[t]);
                }
            }
        }
    }

    if (p.stats) {
        // lanes l and l^32 hold the same column; then the 4 waves are summed in a fixed order via LDS
        __syncthreads();
        float* red = smem;   // [4][BN][2]
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float a = s1[t] + __shfl_xor(s1[t], 32);
            const float b = s2[t] + __shfl_xor(s2[t], 32);
            if (khalf == 0) {
                red[(wave * BN + t * 32 + lrow) * 2 + 0] = a;
                red[(wave * BN + t * 32 + lrow) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
            p.stats[(int64_t)blockIdx.x * 2 * p.N + n0 + tid] = a;
            p.stats[(int64_t)blockIdx.x * 2 * p.N + p.N + n0 + tid] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT GEMM v2: LDS-DMA pipeline.
//
// The register-staged kernel above exposes HBM/L2 latency: one k-tile of prefetch, two waves per SIMD and
// one M-tile per block leave each wave idle ~60 % of the time.  v2 fixes the structure, CDNA4-style:
//   * global_load_lds_dwordx4 (LDS-DMA) copies tiles HBM -> LDS without touching VGPRs, so a 3-stage ring
//     keeps two k-tiles in flight behind the one being multiplied (counted `s_waitcnt vmcnt(N)` + raw
//     `s_barrier`: the DMA queue is never drained inside the loop);
//   * with no staging registers the kernel fits 3 workgroups per CU (48 KB LDS, <=168 VGPR+AGPR);
//   * each block is persistent over a run of consecutive M-tiles and walks one flat (tile, k-tile) sequence,
//     so the next tile's loads are already in flight during the epilogue stores;
//   * BN-apply + activation of the producer is applied when the A fragment is READ from LDS (3 VALU per
//     element, hidden under 64-cycle MFMAs) because a DMA cannot transform;
//   * the LDS image is lane-linear per DMA instruction (hardware rule), so bank conflicts are removed by
//     XOR-swizzling the 16-B chunk index on the SOURCE address and on the fragment read (chunk ^ (row>>2)&3);
//   * block id -> (m-run, n-tile) puts the n-tiles that share an A panel on the same XCD (bid % 8).
// Requirements: K % 4 == 0 (16-B aligned input rows); other K (the 75-channel heads' data gradient, the 10-channel gate) take
// the register-staged kernel.  A ragged output width (N % 4 != 0) is handled in the TRANSPOSED epilogue: up to four element
// stores off one address per row.  (An earlier scalar tail on the un-transposed accumulators — 16*TN stores with 16*TN
// addresses, fully unrolled — cost ~40 VGPRs and made the TN >= 3 main loops spill to scratch.)
// Out-of-range rows are clamped to a valid row (their outputs are never stored); k-chunks past K re-read the
// row start and are annihilated by zeroed B fragments (and zero scale/shift).
// ------------------------------------------------------------------------------------------------
__device__ const float4 mny_zero16 = {0.f, 0.f, 0.f, 0.f};   // DMA source for B chunks past K

struct Gemm2Args {   // A / B / addend / C: float* (BF = 0) or bf16_t* (BF = 1)
    const void* A; const float* in_scale; const float* in_shift; int in_act;
    const void* B; const float* bias; const void* addend; void* C; float* stats;
    int64_t M; int K; int N;
    int m_tiles, tiles_per_block, gx, n_tiles;
    // RED = 1 (data-gradient GEMM feeding a BN unit): C is that unit's dL/d(output); its BN-backward sums
    // (sum dz, sum dz*xhat per column, dz = C * act'(r_scale*rY + r_shift)) leave through `stats` instead of the plain column sums
    const void* rY; const float* r_scale; const float* r_shift; const float* r_mean; const float* r_invstd; int r_act;
};

// XF: 0 = A used as is, 1 = scale/shift + min(max(z, slope*z), hi) activation, 2 = scale/shift + hswish
// BF: 0 = fp32 operands, v_mfma_f32_32x32x2_f32;  1 = bf16 operands (A, B, addend, C), v_mfma_f32_32x32x16_bf16.
//     The LDS image is the same in bytes (64-B rows of four 16-B chunks): a chunk is 4 fp32 or 8 bf16 k-values, a stage
//     covers 16 or 32 k, and the lane's 16-B fragment read IS the bf16 MFMA operand (k-block = lane>>5), so the bf16
//     main loop is 2 MFMAs per accumulator per stage instead of 16.
#ifndef MNY_W6_GROUP
#define MNY_W6_GROUP 4            // SWP: column blocks advanced together (an accumulator's next product is this many MFMA issues away)
#endif
#ifndef MNY_W6_SWP
#define MNY_W6_SWP 1              // planes mode without a reduction epilogue: the A fragment of stage t is read, transformed and cut BETWEEN the MFMAs of stage t-1 (0: one stage at a time)
#endif
// RB (RED kernels only): the product carries the per-column constant p.bias INSIDE the sums of the reduction epilogue (mny_pw_lr_fix); a template
// flag, not a run-time test: one more live register in the data-gradient instantiations broke the bf16 TN = 3 addend form (NaN outputs, round 6)
template <int TN, int XF, int BF, int RED = 0, int X6 = 0, int RB = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((X6 == 3 && RED == 0 && MNY_W6_SWP) ? 2 : 3, 8))) void pw_gemm_nt_dma_kernel(Gemm2Args p) {
    using T = typename std::conditional<BF != 0, bf16_t, float>::type;
    constexpr int EPC = BF ? 8 : 4;                                           // elements per 16-B chunk
    constexpr int BKE = 4 * EPC;                                              // k-values per stage
    const T* pA = (const T*)p.A;
    const T* pB = (const T*)p.B;
    const T* pAdd = (const T*)p.addend;
    T* pC = (T*)p.C;
    constexpr int BN = 32 * TN, BKD = 16, S = 3;
    // X6 == 3: B arrives PRE-CUT (mny_cut3_batch): three bf16 planes [N][nk][2][8], the eight k-values of a 16-B chunk in the order a lane
    // of this kernel holds them (k = 4h..4h+3, 8+4h..8+4h+3 for half h); a stage of B is 3 x 32*TN rows x 32 B, one DMA instruction per
    // (plane, 32-row block).  Lane l of that instruction fetches (column l & 31, half l >> 5), so the lane-linear LDS image is
    // [half][column] and lane (column, half) reads back ITS OWN 16 bytes: consecutive lanes, consecutive chunks, no bank conflict.
    // (Round 2 laid the image out [column][half]: the lanes of a half then read every OTHER chunk — SQ_LDS_BANK_CONFLICT was 40 % of
    // the LDS-active cycles of every planes kernel, profiles/r03_pmc_step_lds.txt.)
    constexpr int A_ST = BM * BKD, B_ST = X6 == 3 ? BN * 24 : BN * BKD;                          // floats
    constexpr int NA = BM / 16, NB = X6 == 3 ? 3 * TN : BN / 16, NL = NA + NB;                   // 1-KiB DMA instructions per stage
    constexpr int LPW = (NL + 3) / 4;                                         // per wave (surplus ones duplicate the last)
    // Ring layout: [S slots of A][SB slots of B][scale / shift cache].  Planes mode (round 3): the pre-cut weight planes are L2-resident
    // and need less lead than the activation rows from HBM, so their ring is TWO slots deep (stage t+1 is requested while stage t is
    // multiplied, A stays two stages ahead): 3 x 8 KB + 2 x 3 TN KB is exactly the LDS of the plain mode (3 x (8 + 2 TN) KB), so the planes
    // kernels keep the plain mode's THREE resident workgroups per CU — with a three-slot plane ring (60 KB at TN = 4) they ran two.
    // Software-pipelined form (round 3, SWP): the products of stage t-1 and the cut of stage t share an iteration, so a plane slot is
    // read one iteration after it landed: the plane ring is three slots again (24 + 9 TN KB per workgroup; this form runs two
    // workgroups per CU on registers — 200-250 VGPRs — anyway, 2 x 69 KB fits).
    constexpr bool SWP = X6 == 3 && RED == 0 && MNY_W6_SWP != 0;
    constexpr int SB = X6 == 3 ? (SWP ? 3 : 2) : S;
    constexpr int NA_W = NA / 4;                                              // A instructions per wave: slots [0, NA_W) are A, the rest B
    static_assert(NA % 4 == 0 && NA_W < LPW, "slot split");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sA = smem;
    float* const sB = smem + S * A_ST;
    float* sScale = sB + SB * B_ST;

    // block -> (m-run x, n-tile y): the n_tiles blocks of one m-run are consecutive within one XCD
    const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
    const int y = local % p.n_tiles, x = (local / p.n_tiles) * 8 + xcd;
    if (x >= p.gx) return;
    const int n0 = y * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, khalf = lane >> 5;
    const int swz = (lrow >> 2) & 3;
    const int nk = (p.K + BKE - 1) / BKE;
    const int Kpad = nk * BKE;
    float* sShift = sScale + Kpad;
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);
    if (XF != 0) {
        const bool has_xf = p.in_scale != nullptr;
        for (int k = tid; k < Kpad; k += 256) {
            sScale[k] = (k < p.K) ? (has_xf ? p.in_scale[k] : 1.f) : 0.f;
            sShift[k] = (k < p.K && has_xf) ? p.in_shift[k] : 0.f;
        }
    }
    const int mt_begin = x * p.tiles_per_block;
    const int mt_end = min(mt_begin + p.tiles_per_block, p.m_tiles);
    const int total = (mt_end - mt_begin) * nk;

    // DMA instruction j (wave-uniform, j = wv + 4i) covers tile rows 16j..16j+15, 4 lanes per row.  Per-lane state is kept
    // to the minimum — the row-in-group, ONE swizzled chunk offset (the swizzle depends on row & 15 only) and the B row
    // pointers; everything else is scalar.  (Per-instruction row/offset arrays pushed the TN >= 3 kernels over the
    // 168-VGPR budget of 3 waves/SIMD and the main loop spilled to scratch.)
    const int drow = lane >> 2;
    const int dk = ((lane & 3) ^ ((drow >> 2) & 3)) * EPC;       // swizzled source chunk, in elements
    int d_lds[LPW], d_row0[LPW];
    bool d_isA[LPW];
    const T* d_bptr[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wv + 4 * i;
        if (j >= NL) j = NL - 1;
        d_isA[i] = j < NA;                                        // (== i < NA_W: NA is a multiple of 4)
        d_row0[i] = (d_isA[i] ? j : j - NA) * 16;                 // scalar
        d_lds[i] = d_isA[i] ? j * 256 : (j - NA) * 256;           // scalar, relative to the slot of its own ring
        if constexpr (X6 == 3) {
            const int jb = d_isA[i] ? 0 : j - NA;                 // (plane, column block) of this instruction
            const int pl = jb / TN, u = jb - pl * TN;
            int n = n0 + u * 32 + (lane & 31);                    // LDS image of a (plane, column block): [k half][32 columns] 16-B chunks, lane-linear
            if (n >= p.N) n = p.N - 1;
            d_lds[i] = d_isA[i] ? j * 256 : jb * 256;
            d_bptr[i] = reinterpret_cast<const T*>(reinterpret_cast<const char*>(p.B) + ((int64_t)pl * p.N + n) * nk * 32 + (lane >> 5) * 16);
        } else {
            int n = n0 + d_row0[i] + drow;
            if (n >= p.N) n = p.N - 1;
            d_bptr[i] = pB + (int64_t)n * p.K + dk;
        }
    }
    const T* zero_src = reinterpret_cast<const T*>(&mny_zero16);
    const bool ragged_k = (p.K % BKE) != 0;

    // Running DMA sources (round 3): one pointer per slot, advanced by one k-stage per issue and re-seated when the flat walk enters a
    // new M tile.  The first cut recomputed every source from (tile, k-tile) per instruction — a 64-bit multiply-add, clamps and a
    // wave-uniform A / B branch per DMA instruction: ~30 vector + ~20 scalar instructions and 4-6 branches per stage in a loop whose
    // waves are latency-bound; now a stage's issue is LPW pointer bumps.
    const T* d_cur[LPW];
    int d_step[LPW];                                              // elements of T per k-stage (wave-uniform)
#pragma unroll
    for (int i = 0; i < LPW; ++i) d_step[i] = d_isA[i] ? BKE : (X6 == 3 ? 32 / (int)sizeof(T) : BKE);
    auto seat = [&](int mt) {                                     // sources of stage (mt, k-tile 0)
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            if (d_isA[i]) {
                int m = mt * BM + d_row0[i] + drow;
                if (m >= (int)p.M) m = (int)p.M - 1;
                d_cur[i] = pA + (int64_t)m * p.K + dk;
            } else {
                d_cur[i] = d_bptr[i];
            }
        }
    };
    auto issue_part = [&](int kt, float* stage, int i0, int i1) {  // slots [i0, i1) of this wave into `stage` (a slot of the A or the B ring)
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            if (i < i0 || i >= i1) continue;
            const T* src = d_cur[i];
            if (ragged_k) {                                       // kernel-uniform; K % 16 != 0 only
                const bool kout = kt == nk - 1 && kt * BKE + dk >= p.K;      // this lane's chunk lies past K
                if (kout) src = d_isA[i] ? d_cur[i] - (kt * BKE + dk) : (X6 == 3 ? src : zero_src);   // A: finite filler (row start), annihilated by zero B / zero scale
            }
            d_cur[i] += d_step[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };
    auto issue_a = [&](int kt, int slot) { issue_part(kt, sA + slot * A_ST, 0, NA_W); };
    auto issue_b = [&](int kt, int slot) { issue_part(kt, sB + slot * B_ST, NA_W, LPW); };

    f32x16 acc[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float s1[TN], s2[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { s1[t] = 0.f; s2[t] = 0.f; }

    auto compute = [&](int kt, int slot, int slot_b) {
        const float* stA = sA + slot * A_ST;
        const float* stB = sB + slot_b * B_ST;
        const float* a_row = stA + (wv * 32 + lrow) * BKD;
        if constexpr (BF != 0) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int chunk = kc * 2 + khalf;
                v4u_t au = lds_read_u4(a_row + ((chunk ^ swz) << 2));
                v4u_t bu[TN];
#pragma unroll
                for (int u = 0; u < TN; ++u) bu[u] = lds_read_u4(stB + (u * 32 + lrow) * BKD + ((chunk ^ swz) << 2));
                v4f_t sc0, sc1, sh0, sh1;
                if (XF != 0) {
                    const int kbase = kt * BKE + chunk * 8;
                    sc0 = lds_read_f4(sScale + kbase); sc1 = lds_read_f4(sScale + kbase + 4);
                    sh0 = lds_read_f4(sShift + kbase); sh1 = lds_read_f4(sShift + kbase + 4);
                }
                MNY_LGKM_WAIT(au);
#pragma unroll
                for (int u = 0; u < TN; ++u) MNY_LGKM_DEP(bu[u]);
                if (XF != 0) {
                    MNY_LGKM_DEP(sc0); MNY_LGKM_DEP(sc1); MNY_LGKM_DEP(sh0); MNY_LGKM_DEP(sh1);
                    float z[8] = {__uint_as_float(au.x << 16), __uint_as_float(au.x & 0xffff0000u), __uint_as_float(au.y << 16),
                                  __uint_as_float(au.y & 0xffff0000u), __uint_as_float(au.z << 16), __uint_as_float(au.z & 0xffff0000u),
                                  __uint_as_float(au.w << 16), __uint_as_float(au.w & 0xffff0000u)};
                    const float scv[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                    const float shv[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float zz = fmaf(z[e], scv[e], shv[e]);
                        z[e] = XF == 1 ? __builtin_amdgcn_fmed3f(zz, slope * zz, hi) : zz * __builtin_amdgcn_fmed3f(zz + 3.f, 0.f, 6.f) * (1.f / 6.f);
                    }
                    au = v4u_t{pack_bf16x2(z[0], z[1]), pack_bf16x2(z[2], z[3]), pack_bf16x2(z[4], z[5]), pack_bf16x2(z[6], z[7])};
                }
                const bf16x8_t a8 = __builtin_bit_cast(bf16x8_t, au);
#pragma unroll
                for (int u = 0; u < TN; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8_t, bu[u]), acc[u], 0, 0, 0);
            }
            return;
        }
        if constexpr (X6 != 0) {
            // the lane's eight k-values of this stage: chunks khalf and 2 + khalf (the same two the fp32 path reads one after the other)
            const int c0 = khalf, c1 = 2 + khalf;
            v4f_t a0 = lds_read_f4(a_row + ((c0 ^ swz) << 2)), a1 = lds_read_f4(a_row + ((c1 ^ swz) << 2));
            v4f_t b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;          // (never a copy of a0: its register is still in flight here)
            if constexpr (X6 != 3) { b0 = lds_read_f4(stB + lrow * BKD + ((c0 ^ swz) << 2)); b1 = lds_read_f4(stB + lrow * BKD + ((c1 ^ swz) << 2)); }
            v4f_t sc0, sc1, sh0, sh1;
            if (XF != 0) {
                sc0 = lds_read_f4(sScale + kt * BKD + c0 * 4); sc1 = lds_read_f4(sScale + kt * BKD + c1 * 4);
                sh0 = lds_read_f4(sShift + kt * BKD + c0 * 4); sh1 = lds_read_f4(sShift + kt * BKD + c1 * 4);
            }
            MNY_LGKM_WAIT(a0);
            MNY_LGKM_DEP(a1); MNY_LGKM_DEP(b0); MNY_LGKM_DEP(b1);
            if (XF != 0) {
                MNY_LGKM_DEP(sc0); MNY_LGKM_DEP(sc1); MNY_LGKM_DEP(sh0); MNY_LGKM_DEP(sh1);
                float z[8] = {fmaf(a0.x, sc0.x, sh0.x), fmaf(a0.y, sc0.y, sh0.y), fmaf(a0.z, sc0.z, sh0.z), fmaf(a0.w, sc0.w, sh0.w),
                              fmaf(a1.x, sc1.x, sh1.x), fmaf(a1.y, sc1.y, sh1.y), fmaf(a1.z, sc1.z, sh1.z), fmaf(a1.w, sc1.w, sh1.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = XF == 1 ? __builtin_amdgcn_fmed3f(z[e], slope * z[e], hi) : z[e] * __builtin_amdgcn_fmed3f(z[e] + 3.f, 0.f, 6.f) * (1.f / 6.f);
                a0 = v4f_t{z[0], z[1], z[2], z[3]}; a1 = v4f_t{z[4], z[5], z[6], z[7]};
            }
            bf16x8_t ah, am, al;
            x6_split(a0, a1, ah, am, al);
            if constexpr (X6 == 3) {
                const float* src0 = stB + (khalf * 32 + lrow) * 4;
                v4f_t ph = lds_read_f4(src0), pm = lds_read_f4(src0 + TN * 256), pl = lds_read_f4(src0 + 2 * TN * 256);
                MNY_LGKM_WAIT(ph); MNY_LGKM_DEP(pm); MNY_LGKM_DEP(pl);
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    v4f_t nh, nm, nl;
                    if (u + 1 < TN) {                             // the next column block's pieces are requested before this one's MFMAs
                        const float* src = stB + ((u + 1) * 64 + khalf * 32 + lrow) * 4;
                        nh = lds_read_f4(src); nm = lds_read_f4(src + TN * 256); nl = lds_read_f4(src + 2 * TN * 256);
                    }
                    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, ph), bm = __builtin_bit_cast(bf16x8_t, pm), bl = __builtin_bit_cast(bf16x8_t, pl);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u], 0, 0, 0);
                    if (u + 1 < TN) { MNY_LGKM_WAIT(nh); MNY_LGKM_DEP(nm); MNY_LGKM_DEP(nl); ph = nh; pm = nm; pl = nl; }
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < TN; ++u) {
                v4f_t n0, n1;
                if (u + 1 < TN) {                                 // next column block's fragment is requested before this one's MFMAs
                    n0 = lds_read_f4(stB + ((u + 1) * 32 + lrow) * BKD + ((c0 ^ swz) << 2));
                    n1 = lds_read_f4(stB + ((u + 1) * 32 + lrow) * BKD + ((c1 ^ swz) << 2));
                }
                bf16x8_t bh, bm, bl;
                x6_split(b0, b1, bh, bm, bl);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[u], 0, 0, 0);      // small terms first
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[u], 0, 0, 0);
                if (u + 1 < TN) { MNY_LGKM_WAIT(n0); MNY_LGKM_DEP(n1); b0 = n0; b1 = n1; }
            }
            return;
        }
#pragma unroll
        for (int kc = 0; kc < BKD / 8; ++kc) {
            const int chunk = kc * 2 + khalf;
            v4f_t af = lds_read_f4(a_row + ((chunk ^ swz) << 2));
            v4f_t bf[TN];
#pragma unroll
            for (int u = 0; u < TN; ++u) bf[u] = lds_read_f4(stB + (u * 32 + lrow) * BKD + ((chunk ^ swz) << 2));
            v4f_t sc, sh;
            if (XF != 0) {
                const int kbase = kt * BKD + chunk * 4;
                sc = lds_read_f4(sScale + kbase);
                sh = lds_read_f4(sShift + kbase);
            }
            MNY_LGKM_WAIT(af);
#pragma unroll
            for (int u = 0; u < TN; ++u) MNY_LGKM_DEP(bf[u]);
            if (XF != 0) {
                MNY_LGKM_DEP(sc);
                MNY_LGKM_DEP(sh);
                float z0 = fmaf(af.x, sc.x, sh.x), z1 = fmaf(af.y, sc.y, sh.y), z2 = fmaf(af.z, sc.z, sh.z), z3 = fmaf(af.w, sc.w, sh.w);
                if (XF == 1) {
                    af.x = fminf(fmaxf(z0, slope * z0), hi); af.y = fminf(fmaxf(z1, slope * z1), hi);
                    af.z = fminf(fmaxf(z2, slope * z2), hi); af.w = fminf(fmaxf(z3, slope * z3), hi);
                } else {
                    af.x = z0 * fminf(fmaxf(z0 + 3.f, 0.f), 6.f) * (1.f / 6.f); af.y = z1 * fminf(fmaxf(z1 + 3.f, 0.f), 6.f) * (1.f / 6.f);
                    af.z = z2 * fminf(fmaxf(z2 + 3.f, 0.f), 6.f) * (1.f / 6.f); af.w = z3 * fminf(fmaxf(z3 + 3.f, 0.f), 6.f) * (1.f / 6.f);
                }
            }
            // accumulators interleaved: consecutive MFMAs never depend on each other
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf[u].x, acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf[u].y, acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf[u].z, acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf[u].w, acc[u], 0, 0, 0);
        }
    };

    // Epilogue.  The MFMA layout gives a lane one COLUMN (16 rows of it); storing that directly is 64 four-byte
    // stores per lane.  Instead each group of 4 accumulator registers (rows b..b+3 of the lane's column) is transposed
    // across the lane's quad with two DPP butterfly stages, after which lane jq of a quad holds row b+jq, columns
    // 4q..4q+3 -> one 16-byte store (4x fewer store instructions, full 128-B row segments per quad-row).
    const int quad = lrow >> 2, jq = lane & 3;
    const bool nvec = (p.N & 3) == 0;
    auto epilogue = [&](int mt) {
        const int64_t m0 = (int64_t)mt * BM;
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int colq = n0 + u * 32 + quad * 4;              // first of this lane's 4 output columns
            const bool cok = colq < p.N;                          // N % 4 == 0: all four or none; ragged N: per element below
            float4 bv = f4zero();
            if (p.bias && cok) {
                if (nvec) bv = ld4(p.bias + colq);
                else {
                    bv.x = p.bias[colq];
                    if (colq + 1 < p.N) bv.y = p.bias[colq + 1];
                    if (colq + 2 < p.N) bv.z = p.bias[colq + 2];
                    if (colq + 3 < p.N) bv.w = p.bias[colq + 3];
                }
            }
            if (RED) {                                            // BN-backward sums of the unit this gradient belongs to
                const int col = n0 + u * 32 + lrow;
                const bool ccol = col < p.N;
                const int cc = ccol ? col : 0;
                const float rsc = p.r_scale[cc], rsh = p.r_shift[cc], rmu = p.r_mean[cc], ris = p.r_invstd[cc];
                float rbias = 0.f;                                // (mny_pw_lr_fix: the gradient carries a per-column constant; data gradients have none)
                if constexpr (RB != 0) rbias = p.bias[cc];
                const float rslope = act_slope(p.r_act), rhi = act_hi(p.r_act);
                const int64_t rbase = m0 + wv * 32 + 4 * khalf;
                const int rows_left = (int)max((int64_t)0, min((int64_t)64, p.M - rbase));   // rows rbase + 8*gq + j, 8*gq + j < rows_left, exist
                const T* ybase = (const T*)p.rY + (rows_left > 0 ? rbase : 0) * p.N + cc;
                const T* abase = RED == 2 ? pAdd + (rows_left > 0 ? rbase : 0) * p.N + cc : nullptr;   // RED = 2: earlier contributions to the same gradient
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {                  // four rows at a time: the loads of a group are all that is in flight
                    float yv[4], av[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) yv[j] = ld1(ybase + (ccol && 8 * gq + j < rows_left ? 8 * gq + j : 0) * p.N);
                    if (RED == 2) {                               // the sums are over the COMPLETE gradient = product + addend
#pragma unroll
                        for (int j = 0; j < 4; ++j) av[j] = ld1(abase + (ccol && 8 * gq + j < rows_left ? 8 * gq + j : 0) * p.N);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float z = fmaf(yv[j], rsc, rsh);
                        const float dact = p.r_act >= MNY_ACT_HSWISH ? act_bwd(z, p.r_act) : (z > 0.f ? 1.f : rslope) * (z < rhi ? 1.f : 0.f);
                        const float dz = stored<T>(RB != 0 ? acc[u][gq * 4 + j] + rbias + av[j] : acc[u][gq * 4 + j] + av[j]) * dact;
                        if (ccol && 8 * gq + j < rows_left) { s1[u] += dz; s2[u] = fmaf(dz, (yv[j] - rmu) * ris, s2[u]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (p.stats) {                                 // column sums come from the un-transposed registers
                const bool ccol = n0 + u * 32 + lrow < p.N;
                if (m0 + BM <= p.M) {                             // whole tile inside M (all but the last one): packed pairs, no row tests
                    v2f a1 = v2f{0.f, 0.f}, a2 = v2f{0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const v2f q = v2f{stored<T>(acc[u][r]), stored<T>(acc[u][r + 1])};
                        a1 += q;
                        a2 = __builtin_elementwise_fma(q, q, a2);
                    }
                    if (ccol) { s1[u] += a1.x + a1.y; s2[u] += a2.x + a2.y; }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = m0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                        const float q = stored<T>(acc[u][r]);
                        if (ccol && row < p.M) { s1[u] += q; s2[u] = fmaf(q, q, s2[u]); }
                    }
                }
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float r0 = acc[u][gq * 4 + 0], r1 = acc[u][gq * 4 + 1], r2 = acc[u][gq * 4 + 2], r3 = acc[u][gq * 4 + 3];
                {   // stage A: exchange with lane^1 inside the quad
                    const bool odd = lane & 1;
                    const float xa = odd ? r0 : r1, xb = odd ? r2 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0xB1, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0xB1, 0xF, 0xF, true));
                    if (odd) { r0 = ya; r2 = yb; } else { r1 = ya; r3 = yb; }
                }
                {   // stage B: exchange with lane^2
                    const bool hi2 = lane & 2;
                    const float xa = hi2 ? r0 : r2, xb = hi2 ? r1 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0x4E, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0x4E, 0xF, 0xF, true));
                    if (hi2) { r0 = ya; r1 = yb; } else { r2 = ya; r3 = yb; }
                }
                const int64_t row = m0 + wv * 32 + 8 * gq + 4 * khalf + jq;
                if (cok && row < p.M) {
                    float4 v = make_float4(r0, r1, r2, r3);
                    if (p.bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }        // wave-uniform: only the two head convs carry a bias
                    if (nvec) {
                        if (pAdd) add4(v, ld4(pAdd + row * p.N + colq));
                        // a row of this 32-column tile is 128 contiguous bytes in fp32 (whole lines: streaming stores) but 64 in bf16 storage: non-temporal
                        // PARTIAL lines cost DRAM efficiency (round 5: MobileNetV3 512x512 bs 64 bf16 14.05 -> 13.91 ms with plain stores, same box)
                        if (sizeof(T) == 2) st4(pC + row * p.N + colq, v); else st4_stream(pC + row * p.N + colq, v);
                    } else {
                        // ragged width (75-channel heads, 10-channel gate): rows are not 16-B aligned -> up to four element stores
                        // off ONE address (the transposed layout keeps this cheap: one row, consecutive columns)
                        T* dst = pC + row * p.N + colq;
                        const T* ad = pAdd ? pAdd + row * p.N + colq : nullptr;
                        st1(dst, v.x + (ad ? ld1(ad) : 0.f));
                        if (colq + 1 < p.N) st1(dst + 1, v.y + (ad ? ld1(ad + 1) : 0.f));
                        if (colq + 2 < p.N) st1(dst + 2, v.z + (ad ? ld1(ad + 2) : 0.f));
                        if (colq + 3 < p.N) st1(dst + 3, v.w + (ad ? ld1(ad + 3) : 0.f));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[u][gq * 4 + i] = 0.f;
                // keep the 4*TN store groups sequential: letting the scheduler hoist every addend load / transpose ahead of
                // the first store raised the kernel's peak VGPR demand by ~27 and made the TN >= 3 main loops spill
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // flat (tile, k-tile) walk; stage t lives in ring slot t % 3, two stages stay in flight behind the consumer
    int c_mt = mt_begin, c_kt = 0, c_slot = 0, c_slot_b = 0;           // stage being consumed
    auto consume = [&]() {
        compute(c_kt, c_slot, X6 != 3 ? c_slot : c_slot_b);
        if (++c_kt == nk) { epilogue(c_mt); c_kt = 0; ++c_mt; }
        if (++c_slot == S) c_slot = 0;
        c_slot_b ^= 1;
    };
    if constexpr (X6 != 3) {
        int i_mt = mt_begin, i_kt = 0, i_slot = 0;       // next stage to issue
        seat(mt_begin);
        auto issue_next = [&]() {
            issue_a(i_kt, i_slot); issue_b(i_kt, i_slot);
            if (++i_kt == nk) { i_kt = 0; ++i_mt; if (i_mt < mt_end) seat(i_mt); }
            if (++i_slot == S) i_slot = 0;
        };
        const int pre = total < S - 1 ? total : S - 1;
        for (int t = 0; t < pre; ++t) issue_next();
        const int steady = total - pre;                      // steps that still have a stage to issue
        for (int t = 0; t < steady; ++t) {
            wait_vmcnt<LPW*(S - 2)>();                       // my share of the oldest stage has landed
            __builtin_amdgcn_s_barrier();                    // ... for every wave; the slot refilled below is fully consumed
            issue_next();
            consume();
        }
        for (int t = 0; t < pre; ++t) {                      // drain
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            consume();
        }
    } else if constexpr (SWP) {
        // ---- planes mode, software-pipelined -------------------------------------------------------------------------------------
        // Measured on the one-stage-at-a-time form (profiles/r03_pmc_step.txt, ISA): per k-stage a wave issues ~110 vector-ALU
        // instructions (LDS addresses, BN-apply + activation of the A fragment, the three-way cut) BEFORE its 6 TN MFMAs, back to back —
        // ~450 cycles in which it feeds the matrix pipe nothing; SQ_VALU_MFMA_BUSY 40-44 % on the fat shapes.  Here iteration t cuts
        // stage t while the products of stage t-1 issue: the cut is split into 24 pieces of 2-4 instructions and one piece follows each
        // MFMA (an 8-pass MFMA occupies the pipe for 32 cycles, a piece issues in 8-16), LDS reads carry their offsets as immediates.
        // Same products in the same order per accumulator as the one-at-a-time form: bit-identical results.
        // Measured (tools/ab/w6_bench.py, same box): 512x512 at 22x22 0.499 -> 0.464 ms (130 -> 140 TF/s), K960 N160 -12 %, K576 N96 -9 %,
        // K1280 N512 -6 %.  What it did NOT fix: the kernel is not bound by any one thing — timing builds with a part removed give, for
        // 512x512: no plane DMA -18 %, no plane LDS reads -17 %, no A DMA -9 %, no cut -7 %, no barrier / vmcnt wait -5 %, all of them
        // together 0.27 ms (the MFMAs + epilogue alone, 57 % of the bf16 peak / 6).  Issuing the DMA pieces between the MFMAs, requesting
        // the next stage's first planes a stage ahead and the size of the accumulator group (2 / 4) all measured neutral.  The plane
        // traffic (12 KB of DMA per stage per workgroup, 12 KB of LDS reads per stage per WAVE) is the largest single share.  The 64-row
        // wave tile that halves it per FLOP (two row blocks share each plane read and DMA: 256 x 128 workgroup tile, 128 accumulators
        // pinned to accumulation registers, 84 KB of LDS, hence ONE wave per SIMD) was built on this scaffold and measured: 512x512
        // 0.47 -> 0.55 ms, K1280 N512 0.29 -> 0.34 — a lone wave per SIMD exposes every barrier, vmcnt and LDS round trip that the
        // second wave covers here; removed again.
        int a_mt = mt_begin, a_kt = 0, a_slot = 0, a_n = 0;   // next A stage to issue
        int b_mt = mt_begin, b_kt = 0, b_slot = 0, b_n = 0;   // next B stage to issue
        auto seat_part = [&](int mt, bool part_a) {
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                if ((i < NA_W) != part_a) continue;
                if (part_a) {
                    int m = mt * BM + d_row0[i] + drow;
                    if (m >= (int)p.M) m = (int)p.M - 1;
                    d_cur[i] = pA + (int64_t)m * p.K + dk;
                } else {
                    d_cur[i] = d_bptr[i];
                }
            }
        };
        seat_part(mt_begin, true); seat_part(mt_begin, false);
        auto next_a = [&]() {
            issue_a(a_kt, a_slot); ++a_n;
            if (++a_kt == nk) { a_kt = 0; ++a_mt; if (a_mt < mt_end) seat_part(a_mt, true); }
            if (++a_slot == S) a_slot = 0;
        };
        auto next_b = [&]() {
            issue_b(b_kt, b_slot); ++b_n;
            if (++b_kt == nk) { b_kt = 0; ++b_mt; if (b_mt < mt_end) seat_part(b_mt, false); }
            if (++b_slot == SB) b_slot = 0;
        };
        // loop-invariant LDS byte addresses of the lane
        const unsigned oA0 = lds_off(sA + (wv * 32 + lrow) * BKD + ((khalf ^ swz) << 2));
        const unsigned oA1 = lds_off(sA + (wv * 32 + lrow) * BKD + (((2 + khalf) ^ swz) << 2));
        const unsigned oB = lds_off(sB + lane * 4);
        const unsigned oS = lds_off(sScale + khalf * 4);
        const unsigned shift_off = (unsigned)Kpad * 4u;
        bf16x8_t qh, qm, ql;                                  // the cut A fragment whose products come next
        int p_kt = 0, p_slot = 0;                             // stage being cut
        int m_slot_b = 0;                                     // plane slot of the stage being multiplied (its tile / k-tile: c_mt, c_kt)
        auto step = [&](auto prep_c, auto mma_c) -> bool {
            constexpr bool PREP = decltype(prep_c)::value, MMA = decltype(mma_c)::value;
            constexpr int NPIECE = 24, NISSUE = MMA ? 6 * TN : 1;
            v4f_t a0, a1, sc0, sc1, sh0, sh1;
            float z[8];
            v2f r1[4], r2[4];
            v4u_t nhu, nmu, nlu;
            auto piece = [&](auto pc) {
                constexpr int P = decltype(pc)::value;
                if constexpr (PREP) {
                    if constexpr (P < 8) {                            // element P: BN-apply + activation of the producer
                        const float av = P < 4 ? a0[P & 3] : a1[P & 3];
                        if constexpr (XF != 0) {
                            const float zz = fmaf(av, P < 4 ? sc0[P & 3] : sc1[P & 3], P < 4 ? sh0[P & 3] : sh1[P & 3]);
                            z[P] = XF == 1 ? __builtin_amdgcn_fmed3f(zz, slope * zz, hi) : zz * __builtin_amdgcn_fmed3f(zz + 3.f, 0.f, 6.f) * (1.f / 6.f);
                        } else {
                            z[P] = av;
                        }
                        asm volatile("" : "+v"(z[P]));               // pins the piece HERE (values used only by the next iteration are otherwise sunk below the MFMAs)
                    } else if constexpr (P < 16) {                    // three-way cut of the pair (2i, 2i+1), in two halves (x6_split's arithmetic)
                        constexpr int i = (P - 8) >> 1;
                        if constexpr (((P - 8) & 1) == 0) {
                            const v2f a = v2f{z[2 * i], z[2 * i + 1]};
                            const v2f ah = v2f{__uint_as_float(__float_as_uint(a.x) & 0xffff0000u), __uint_as_float(__float_as_uint(a.y) & 0xffff0000u)};
                            r1[i] = a - ah;
                            asm volatile("" : "+v"(r1[i]));
                        } else {
                            const v2f b = r1[i];
                            const v2f bh = v2f{__uint_as_float(__float_as_uint(b.x) & 0xffff0000u), __uint_as_float(__float_as_uint(b.y) & 0xffff0000u)};
                            r2[i] = b - bh;
                            asm volatile("" : "+v"(r2[i]));
                        }
                    } else if constexpr (P < 22) {                    // pack the high halves: two dwords per piece
                        constexpr int q = P - 16, w = q >> 1, d = (q & 1) * 2;
                        auto pk = [](float lo, float hi_) { return __builtin_amdgcn_perm(__float_as_uint(hi_), __float_as_uint(lo), 0x07060302u); };
                        unsigned t0, t1;
                        if constexpr (w == 0) { t0 = pk(z[2 * d], z[2 * d + 1]); t1 = pk(z[2 * d + 2], z[2 * d + 3]); }
                        if constexpr (w == 1) { t0 = pk(r1[d].x, r1[d].y); t1 = pk(r1[d + 1].x, r1[d + 1].y); }
                        if constexpr (w == 2) { t0 = pk(r2[d].x, r2[d].y); t1 = pk(r2[d + 1].x, r2[d + 1].y); }
                        asm volatile("" : "+v"(t0), "+v"(t1));
                        if constexpr (w == 0) { nhu[d] = t0; nhu[d + 1] = t1; }
                        if constexpr (w == 1) { nmu[d] = t0; nmu[d + 1] = t1; }
                        if constexpr (w == 2) { nlu[d] = t0; nlu[d + 1] = t1; }
                    }
                }
            };
            auto pieces_of = [&](auto ic) {                           // the pieces that follow issue point I
                constexpr int I = decltype(ic)::value;
                static_for<0, NPIECE>([&](auto pc) {
                    constexpr int P = decltype(pc)::value;
                    if constexpr (P * NISSUE / NPIECE == I) piece(pc);
                });
            };
            if constexpr (!MMA) {                                     // first stage of the walk: nothing to multiply yet
                const unsigned bA = (unsigned)p_slot * (A_ST * 4);
                a0 = lds_read_f4_at<0>(oA0 + bA); a1 = lds_read_f4_at<0>(oA1 + bA);
                if constexpr (XF != 0) {
                    const unsigned bS = oS + (unsigned)p_kt * (BKD * 4), bH = bS + shift_off;
                    sc0 = lds_read_f4_at<0>(bS); sc1 = lds_read_f4_at<32>(bS); sh0 = lds_read_f4_at<0>(bH); sh1 = lds_read_f4_at<32>(bH);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1));
                }
                pieces_of(std::integral_constant<int, 0>{});
            } else {
                // Column blocks in GROUPS of up to four (MNY_W6_GROUP; TN = 5: 3 + 2): product k is issued for every block of the group before
                // product k+1, so an MFMA's predecessor on the same accumulator is `group size` issues back (the pairs form left 38 % of the
                // wave cycles in issue stalls with the pipe 58 % idle: dependent issues).  A group's planes are requested up front in the
                // order the products need them — high, (the cut's operands), low, mid — and waited for with counted lgkmcnt.
                const unsigned bB = oB + (unsigned)m_slot_b * (B_ST * 4);
                v4f_t PH[TN], PM[TN], PL[TN];
                constexpr int NPREP = PREP ? (XF != 0 ? 6 : 2) : 0;  // LDS reads of the cut
                constexpr int G0 = TN <= MNY_W6_GROUP ? TN : (TN + 1) / 2;     // size of the first group (TN = 5 -> 3 + 2)
                auto rdH = [&](auto uc) { constexpr int u = decltype(uc)::value; PH[u] = lds_read_f4_at<u * 1024>(bB); };
                auto rdM = [&](auto uc) { constexpr int u = decltype(uc)::value; PM[u] = lds_read_f4_at<(TN + u) * 1024>(bB); };
                auto rdL = [&](auto uc) { constexpr int u = decltype(uc)::value; PL[u] = lds_read_f4_at<(2 * TN + u) * 1024>(bB); };
                auto tie = [&](auto uc, v4f_t* P_) { constexpr int u = decltype(uc)::value; asm volatile("" : "+v"(P_[u])); };
                auto group = [&](auto u0c, auto gsc, auto firstc) {
                    constexpr int u0 = decltype(u0c)::value, gs = decltype(gsc)::value;
                    constexpr bool first = decltype(firstc)::value;
                    constexpr bool more = u0 + gs < TN;              // a second group follows (its planes are requested before this group's third product)
                    constexpr int I0 = 6 * u0;
                    if constexpr (first) {
                        static_for<u0, u0 + gs>(rdH);
                        if constexpr (PREP) {
                            const unsigned bA = (unsigned)p_slot * (A_ST * 4);
                            a0 = lds_read_f4_at<0>(oA0 + bA); a1 = lds_read_f4_at<0>(oA1 + bA);
                            if constexpr (XF != 0) {
                                const unsigned bS = oS + (unsigned)p_kt * (BKD * 4), bH = bS + shift_off;
                                sc0 = lds_read_f4_at<0>(bS); sc1 = lds_read_f4_at<32>(bS); sh0 = lds_read_f4_at<0>(bH); sh1 = lds_read_f4_at<32>(bH);
                            }
                        }
                        static_for<u0, u0 + gs>(rdL);
                        static_for<u0, u0 + gs>(rdM);
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NPREP + 2 * gs) : "memory");      // high planes landed
                        static_for<u0, u0 + gs>([&](auto uc) { tie(uc, PH); });
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // requested during the first group
                        static_for<u0, u0 + gs>([&](auto uc) { tie(uc, PH); tie(uc, PL); tie(uc, PM); });
                    }
                    auto prod = [&](auto kc, const bf16x8_t& A_, v4f_t* P_) {
                        constexpr int k = decltype(kc)::value;
                        static_for<0, gs>([&](auto jc) {
                            constexpr int j = decltype(jc)::value, u = u0 + j, IA = I0 + k * gs + j;
                            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, __builtin_bit_cast(bf16x8_t, P_[u]), acc[u], 0, 0, 0);
                            if constexpr (PREP && first && k == 0 && j == 0) {     // the cut's operands: queued behind the high planes
                                if constexpr (XF != 0) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a0), "+v"(a1), "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1) : "n"(2 * gs));
                                else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(2 * gs));
                            }
                            pieces_of(std::integral_constant<int, IA>{});
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    };
                    prod(std::integral_constant<int, 0>{}, ql, PH);                // small terms first (the order of the one-at-a-time form)
                    if constexpr (first) {
                        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(gs) : "memory");               // low planes landed
                        static_for<u0, u0 + gs>([&](auto uc) { tie(uc, PL); });
                    }
                    prod(std::integral_constant<int, 1>{}, qh, PL);
                    if constexpr (first) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          // mid planes landed
                        static_for<u0, u0 + gs>([&](auto uc) { tie(uc, PM); });
                        if constexpr (more) { static_for<u0 + gs, TN>(rdH); static_for<u0 + gs, TN>(rdL); static_for<u0 + gs, TN>(rdM); }
                    }
                    prod(std::integral_constant<int, 2>{}, qm, PM);
                    prod(std::integral_constant<int, 3>{}, qm, PH);
                    prod(std::integral_constant<int, 4>{}, qh, PM);
                    prod(std::integral_constant<int, 5>{}, qh, PH);
                };
                group(std::integral_constant<int, 0>{}, std::integral_constant<int, G0>{}, std::true_type{});
                if constexpr (G0 < TN) group(std::integral_constant<int, G0>{}, std::integral_constant<int, TN - G0>{}, std::false_type{});
            }
            bool fin = false;
            if constexpr (MMA) {
                if (++c_kt == nk) { c_kt = 0; fin = true; }
                if (++m_slot_b == SB) m_slot_b = 0;
            }
            if constexpr (PREP) {
                qh = __builtin_bit_cast(bf16x8_t, nhu); qm = __builtin_bit_cast(bf16x8_t, nmu); ql = __builtin_bit_cast(bf16x8_t, nlu);
                if (++p_kt == nk) p_kt = 0;
                if (++p_slot == S) p_slot = 0;
            }
            return fin;
        };
        if (total > 0) { next_a(); next_b(); }               // A(0), B(0), then A(1): the order the counted wait below assumes
        if (total > 1) next_a();
        if (total > 0) {                                     // (three separate code paths, not one loop with a three-way branch: the accumulators
            wait_vmcnt<NA_W>();                              //  then merge only at the loop header and are not copied per iteration)
            if (total == 1) wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                    // A(0), B(0) landed
            if (b_n < total) next_b();                       // B(1)
            if (a_n < total) next_a();                       // A(2)
            step(std::true_type{}, std::false_type{});
            for (int t = 1; t < total; ++t) {
                if (t + 1 < total) wait_vmcnt<NA_W>(); else wait_vmcnt<0>();     // A(t), B(t) landed; A(t+1) may still be in flight
                __builtin_amdgcn_s_barrier();                // every wave has cut stage t-1 and multiplied stage t-2: A slot (t+2) % 3, B slot (t+1) % 3 are free
                if (b_n < total) next_b();                   // B(t+1)
                if (a_n < total) next_a();                   // A(t+2)
                if (step(std::true_type{}, std::true_type{})) { epilogue(c_mt); ++c_mt; }
            }
            step(std::false_type{}, std::true_type{});
            epilogue(c_mt);
        }
    } else {
        // planes mode: A two stages ahead (3 slots), B one stage ahead (2 slots).  The A and the B sources walk the same (tile, k-tile)
        // sequence with their own cursors; issue order inside an iteration is B(t+1) THEN A(t+2), so that at the top of iteration t+1 the
        // in-order counter may leave exactly A(t+2)'s NA_W instructions outstanding: everything older — A(t+1), B(t+1) — has landed.
        int a_mt = mt_begin, a_kt = 0, a_slot = 0, a_n = 0;   // next A stage to issue
        int b_mt = mt_begin, b_kt = 0, b_slot = 0, b_n = 0;   // next B stage to issue
        auto seat_part = [&](int mt, bool part_a) {
#pragma unroll
            for (int i = 0; i < LPW; ++i) {
                if ((i < NA_W) != part_a) continue;
                if (part_a) {
                    int m = mt * BM + d_row0[i] + drow;
                    if (m >= (int)p.M) m = (int)p.M - 1;
                    d_cur[i] = pA + (int64_t)m * p.K + dk;
                } else {
                    d_cur[i] = d_bptr[i];
                }
            }
        };
        seat_part(mt_begin, true); seat_part(mt_begin, false);
        auto next_a = [&]() {
            issue_a(a_kt, a_slot); ++a_n;
            if (++a_kt == nk) { a_kt = 0; ++a_mt; if (a_mt < mt_end) seat_part(a_mt, true); }
            if (++a_slot == S) a_slot = 0;
        };
        auto next_b = [&]() {
            issue_b(b_kt, b_slot); ++b_n;
            if (++b_kt == nk) { b_kt = 0; ++b_mt; if (b_mt < mt_end) seat_part(b_mt, false); }
            b_slot ^= 1;
        };
        if (total > 0) { next_a(); next_b(); }               // A(0), B(0), then A(1): the order the counted wait below assumes
        if (total > 1) next_a();
        for (int t = 0; t < total; ++t) {
            if (t + 1 < total) wait_vmcnt<NA_W>(); else wait_vmcnt<0>();     // A(t), B(t) landed; A(t+1) may still be in flight
            __builtin_amdgcn_s_barrier();                    // every wave is past stage t-1: B slot (t+1) % 2 and A slot (t+2) % 3 are free
            if (b_n < total) next_b();                       // B(t+1)
            if (a_n < total) next_a();                       // A(t+2)
            consume();
        }
    }

    if (p.stats) {
        wait_vmcnt<0>();
        __syncthreads();
        float* red = smem;   // [4][BN][2]
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const float a = s1[u] + __shfl_xor(s1[u], 32);
            const float b = s2[u] + __shfl_xor(s2[u], 32);
            if (khalf == 0) {
                red[(wv * BN + u * 32 + lrow) * 2 + 0] = a;
                red[(wv * BN + u * 32 + lrow) * 2 + 1] = b;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * BN + tid) * 2]; b += red[(w * BN + tid) * 2 + 1]; }
            p.stats[(int64_t)x * 2 * p.N + n0 + tid] = a;
            p.stats[(int64_t)x * 2 * p.N + p.N + n0 + tid] = b;
        }
    }
}

typedef void (*Nt2Kernel)(Gemm2Args);
static Nt2Kernel nt2_kernel(int TN, int XF, int BF = 0, int X6 = 0) {
    if (X6 == 3 && !BF) {                           // pre-cut weight planes (mny_pw_fwd_w6)
#define MNY_K6(T) (XF == 0 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 0, 0, 3> : XF == 1 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 0, 0, 3> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 2, 0, 0, 3>)
        switch (TN) { case 1: return MNY_K6(1); case 2: return MNY_K6(2); case 3: return MNY_K6(3); case 4: return MNY_K6(4); default: return MNY_K6(5); }
#undef MNY_K6
    }
    if (X6 && !BF) {
#define MNY_K6(T) (XF == 0 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 0, 0, 1> : XF == 1 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 0, 0, 1> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 2, 0, 0, 1>)
        switch (TN) { case 1: return MNY_K6(1); case 2: return MNY_K6(2); case 3: return MNY_K6(3); case 4: return MNY_K6(4); default: return MNY_K6(5); }
#undef MNY_K6
    }
#define MNY_K(T) (BF ? (XF == 0 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 1> : XF == 1 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 1> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 2, 1>) \
                     : (XF == 0 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 0> : XF == 1 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 0> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 2, 0>))
    switch (TN) { case 1: return MNY_K(1); case 2: return MNY_K(2); case 3: return MNY_K(3); case 4: return MNY_K(4); default: return MNY_K(5); }
#undef MNY_K
}

// fp32 GEMMs whose matrix-core time exceeds their HBM time take the six-product bf16 form (MNY_X6=0: never, =1: always, for A/B runs)
static int nt_x6(int64_t M, int K, int N) {
    static const int env = getenv("MNY_X6") ? atoi(getenv("MNY_X6")) : -1;
    if (env >= 0) return env != 0;
    const double ai = 2.0 * K * N / (4.0 * (K + N));            // FLOP per byte of the A and C rows
    return ai >= 20.0;                                          // 157 TFLOP/s / 8 TB/s
}

struct Nt2Plan { int TN, n_tiles, m_tiles, gx, tiles_per_block, grid; size_t lds; };

static size_t nt2_lds(int TN, int K, bool xf, int BF = 0) {
    const int Kpad = (int)cdiv(K, BF ? 32 : 16) * (BF ? 32 : 16);
    size_t ring = (size_t)3 * (BM * 16 + 32 * TN * 16) * sizeof(float);
    size_t red = (size_t)4 * 32 * TN * 2 * sizeof(float);
    return (ring > red ? ring : red) + (xf ? 2 * Kpad * sizeof(float) : 0);
}

// kernels whose dynamic LDS exceeds the 64 KB default need the opt-in once (the pre-cut planes mode at TN >= 4 with a deep scale cache)
static void nt2_allow_lds(Nt2Kernel k, size_t lds) {
    if (lds <= 64 * 1024) return;
    (void)allow_lds((const void*)k, lds);
}

// ---- pre-cut weight planes for the six-product form --------------------------------------------------------------------------
// src [R][C] fp32 (a conv weight [Cout][Cin] or its transpose) -> dst: three bf16 planes [R][nk][2][8], nk = ceil(C / 16); chunk
// (row, kt, h) holds k = 16 kt + {4h, 4h+1, 4h+2, 4h+3, 8+4h, ..., 8+4h+3} — the eight values lane-half h of the NT kernel multiplies
// in stage kt — cut by truncation exactly like x6_split (hi + mid + lo == the fp32 value), zeros past C.  One thread per chunk.
struct Cut3Job { const float* src; void* dst; int32_t R, C, block0, pad; };
__global__ __launch_bounds__(256) void cut3_batch_kernel(const Cut3Job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    const Cut3Job jb = jobs[block_job[blockIdx.x]];
    const int nk = (jb.C + 15) / 16;
    const int64_t idx = (int64_t)(blockIdx.x - jb.block0) * 256 + threadIdx.x;       // chunk index: (row * nk + kt) * 2 + h
    if (idx >= (int64_t)jb.R * nk * 2) return;
    const int h = (int)(idx & 1);
    const int64_t rk = idx >> 1;
    const int kt = (int)(rk % nk);
    const int64_t row = rk / nk;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kt * 16 + (e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4));
        x[e] = k < jb.C ? jb.src[row * jb.C + k] : 0.f;
    }
    bf16x8_t hh, mm, ll;
    x6_split(v4f_t{x[0], x[1], x[2], x[3]}, v4f_t{x[4], x[5], x[6], x[7]}, hh, mm, ll);
    const int64_t plane = (int64_t)jb.R * nk * 32;                                     // bytes
    char* d = reinterpret_cast<char*>(jb.dst) + idx * 16;
    *reinterpret_cast<v4f_t*>(d) = __builtin_bit_cast(v4f_t, hh);
    *reinterpret_cast<v4f_t*>(d + plane) = __builtin_bit_cast(v4f_t, mm);
    *reinterpret_cast<v4f_t*>(d + 2 * plane) = __builtin_bit_cast(v4f_t, ll);
}

// resident workgroups per CU for (TN, XF) at a given dynamic-LDS size (queried once per distinct size)
static int nt2_blocks_per_cu(int TN, int XF, size_t lds, int BF = 0) {
    static std::map<uint64_t, int> cache;
    static std::mutex mu;
    const uint64_t key = ((uint64_t)BF << 48) | ((uint64_t)TN << 40) | ((uint64_t)XF << 32) | (uint64_t)lds;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)nt2_kernel(TN, XF, BF), 256, lds) != hipSuccess || nb < 1) nb = 1;
    if (nb > 8) nb = 8;
    cache[key] = nb;
    return nb;
}

static Nt2Plan nt2_plan(int64_t M, int K, int N, bool xf, int BF = 0, int tn_cap = 5) {
    Nt2Plan pl;
    int best = 1, best_pad = 1 << 30;
    static const int max_tn = getenv("MNY_NT_MAXTN") ? atoi(getenv("MNY_NT_MAXTN")) : 5;
    for (int tn = max_tn < tn_cap ? max_tn : tn_cap; tn >= 1; --tn) {        // minimise the padded width; ties -> the wider tile (A read fewer times,
        int pad = (int)cdiv(N, 32 * tn) * 32 * tn;   // longer contiguous output rows)
        if (pad < best_pad) { best_pad = pad; best = tn; }
    }
    // grid depends only on (M,K,N): occupancy is taken for the transform variant so stat_parts() and fwd agree
    const int blocks_per_cu = nt2_blocks_per_cu(best, 1, nt2_lds(best, K, true, BF), BF);
    pl.TN = best;
    pl.n_tiles = (int)cdiv(N, 32 * best);
    pl.m_tiles = (int)cdiv(M, BM);
    int want = 256 * blocks_per_cu / pl.n_tiles;      // one full wave of resident workgroups
    if (want < 8) want = 8;
    if (want > kMaxParts) want = kMaxParts;
    int gx = pl.m_tiles < want ? pl.m_tiles : want;
    pl.tiles_per_block = (int)cdiv(pl.m_tiles, gx);
    // Balance the CUs, not the slots: all blocks are resident at once and share their CU's matrix pipes, so a launch takes
    // (blocks on the busiest CU) x (tiles per block).  Filling every slot (3/CU) with 5.04 tiles' worth of work each means 6
    // rounds on CUs holding 3 blocks = 18 tile-times, where 2 blocks/CU x 8 tiles = 16.  Search tiles-per-block upwards from
    // the one-wave minimum; ties -> more blocks (more waves to hide latency).
    static const bool balance = getenv("MNY_NT_BALANCE") == nullptr || atoi(getenv("MNY_NT_BALANCE")) != 0;
    static const double min_ai = getenv("MNY_NT_BALANCE_AI") ? atof(getenv("MNY_NT_BALANCE_AI")) : 100.0;
    // ... for the matrix-pipe-bound shapes only (FLOP per byte of A + C rows): an HBM-bound layer wants every slot filled
    // (memory-level parallelism), measured 5 % slower with fewer, longer blocks
    const double ai = 2.0 * K * N / ((BF ? 2.0 : 4.0) * (K + N));
    if (balance && ai >= min_ai) {
        const int t_min = pl.tiles_per_block;
        long best_cost = -1;
        int best_t = t_min, best_b = 0;
        for (int t = t_min; t <= 4 * t_min + 4 && t <= pl.m_tiles; ++t) {
            const int g = (int)cdiv(pl.m_tiles, t);
            const int nb = g * pl.n_tiles;                                     // real blocks (the grid's padding blocks exit at once)
            if (nb < 256) break;                                               // never leave CUs without a block
            const long cost = (long)cdiv(nb, 256) * t;
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && nb > best_b)) { best_cost = cost; best_t = t; best_b = nb; }
        }
        pl.tiles_per_block = best_t;
    }
    pl.gx = (int)cdiv(pl.m_tiles, pl.tiles_per_block);
    pl.grid = (int)cdiv(pl.gx, 8) * 8 * pl.n_tiles;
    static const bool plan_debug = getenv("MNY_PLAN_DEBUG") != nullptr;
    if (plan_debug) fprintf(stderr, "nt2_plan M=%lld K=%d N=%d TN=%d bpc=%d m_tiles=%d n_tiles=%d t=%d gx=%d grid=%d\n", (long long)M, K, N, best, blocks_per_cu, pl.m_tiles, pl.n_tiles, pl.tiles_per_block, pl.gx, pl.grid);
    pl.lds = nt2_lds(best, K, xf, BF);
    return pl;
}

struct NtPlan { int TN; int BK; int n_tiles; int m_tiles; int gx; int tiles_per_block; size_t lds; };

static NtPlan nt_plan(int64_t M, int K, int N, bool xf = true) {
    NtPlan pl;
    int best = 1; int best_pad = 1 << 30;
    for (int tn = 4; tn >= 1; --tn) {   // minimise padded N; ties -> wider tile (fewer re-reads of A)
        int pad = (int)cdiv(N, 32 * tn) * 32 * tn;
        if (pad < best_pad) { best_pad = pad; best = tn; }
    }
    pl.TN = best;
    pl.n_tiles = (int)cdiv(N, 32 * best);
    pl.m_tiles = (int)cdiv(M, BM);
    int max_gx = kMaxParts;
    int gx = pl.m_tiles < max_gx ? pl.m_tiles : max_gx;
    pl.tiles_per_block = (int)cdiv(pl.m_tiles, gx);
    pl.gx = (int)cdiv(pl.m_tiles, pl.tiles_per_block);
    // deep K: BK = 32 halves the barriers per FLOP — as long as two blocks still fit the CU's 160 KB of LDS
    auto lds_for = [&](int bk) {
        const int Kpad = (int)cdiv(K, bk) * bk;
        return (size_t)(2 * BM * (bk + 4) + 2 * 32 * best * (bk + 4) + (xf ? 2 * Kpad : 0)) * sizeof(float);
    };
    pl.BK = (K >= 64 && K % 32 == 0 && lds_for(32) <= 80 * 1024) ? 32 : 16;
    pl.lds = lds_for(pl.BK);
    return pl;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[N][K] = sum_m dY[m][n] * act_in(X)[m][k]      (reduction over M, output tiny)
//
// Both operands are M-major, which is exactly the MFMA operand layout (lane = channel, k = row).
// A block owns a BI x BJ tile of dW for a slice of M; it streams 16/32-row chunks of dY and X through a
// double-buffered LDS stage (coalesced float4 loads along the channel axis, BN-apply + activation applied
// to X on the way in) and feeds v_mfma_f32_32x32x2_f32 from conflict-free ds_read_b32 (lanes = consecutive
// channels).  Two decompositions:
//   MODE 0 (fat dW):  2x2 waves, each TIxTJ tiles of 32x32 -> BI = 64*TI, BJ = 64*TJ, chunk = 16 rows.
//   MODE 1 (thin dW): all 4 waves own the SAME TIxTJ tiles (BI = 32*TI, BJ = 32*TJ = whole dW) and split the
//                     32-row chunk four ways; a fixed-order LDS reduction combines them at the end.
// Partials [split][N][K] are then summed in a fixed order by reduce_parts_kernel (deterministic).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {   // X / dY are T* of the kernel instantiation
    const void* X; const float* in_scale; const float* in_shift; int in_act;
    const void* dY; float* partial;
    int64_t M; int K; int N;
    int64_t rows_per_block;
    int gx, gy, splits;          // XCD-aware launches of the LDS-DMA kernels (1-D grid): output tiles gx x gy, M splits; gx == 0: plain 3-D grid
};

// XCD-aware block order of the LDS-DMA weight-gradient kernels (round 3).  The gx*gy output tiles of one M split read the SAME rows of X
// and dY; launched as a (gx, gy, splits) grid their workgroups have consecutive linear ids and the dispatcher deals those round-robin
// over the 8 XCDs — each with its own L2 — so every tile fetched its operands from HBM again: counter traffic 1.67x the algorithmic bytes
// (profiles/r02_traffic_mny_pw_wgrad.json; K64 N384 = 6 tiles: (384 + 6*64) / (384 + 64) = 1.71).  A 1-D grid of 8*ceil(splits/8)*gx*gy
// blocks, block L -> XCD L & 7, puts all tiles of a split on one XCD, next to each other in its dispatch order.
struct WgBlock { int bx, by, bz; bool live; };
__device__ __forceinline__ WgBlock wg_block(const WgradArgs& p) {
    WgBlock b;
    if (p.gx == 0) { b.bx = blockIdx.x; b.by = blockIdx.y; b.bz = blockIdx.z; b.live = true; return b; }
    const int T = p.gx * p.gy, L = blockIdx.x, idx = L >> 3, grp = idx / T, t = idx - grp * T;
    b.bz = grp * 8 + (L & 7);
    b.bx = t % p.gx; b.by = t / p.gx;
    b.live = b.bz < p.splits;
    return b;
}

template <typename T, int MODE, int TI, int TJ>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgradArgs p) {
    constexpr int KC = MODE == 0 ? 16 : 32;
    constexpr int BI = (MODE == 0 ? 64 : 32) * TI;
    constexpr int BJ = (MODE == 0 ? 64 : 32) * TJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                      // [2][KC][BI]  dY
    float* sB = smem + 2 * KC * BI;        // [2][KC][BJ]  act(X)
    float* sScale = sB + 2 * KC * BJ;      // [BJ]
    float* sShift = sScale + BJ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int co0 = blockIdx.x * BI, ci0 = blockIdx.y * BJ;
    const int64_t m_begin = (int64_t)blockIdx.z * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const bool has_xf = p.in_scale != nullptr;
    const bool do_xf = has_xf || p.in_act != MNY_ACT_NONE;
    const bool vecA = (p.N & 3) == 0, vecB = (p.K & 3) == 0;

    for (int j = tid; j < BJ; j += 256) {
        const int ci = ci0 + j;
        sScale[j] = (has_xf && ci < p.K) ? p.in_scale[ci] : 1.f;
        sShift[j] = (has_xf && ci < p.K) ? p.in_shift[ci] : 0.f;
    }
    __syncthreads();

    const int ioff = MODE == 0 ? (wave >> 1) * 32 * TI : 0;
    const int joff = MODE == 0 ? (wave & 1) * 32 * TJ : 0;
    const int krow0 = MODE == 0 ? 0 : wave * (KC / 4);
    constexpr int KROWS = MODE == 0 ? KC : KC / 4;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int A_F4 = KC * BI / 4, B_F4 = KC * BJ / 4;
    constexpr int A_PER = (A_F4 + 255) / 256, B_PER = (B_F4 + 255) / 256;
    float4 ra[A_PER], rb[B_PER];

    auto gload = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int idx = tid + i * 256;
            float4 v = f4zero();
            if (idx < A_F4) {
                const int row = idx / (BI / 4), c = (idx % (BI / 4)) * 4;
                const int64_t m = m0 + row;
                const int co = co0 + c;
                if (m < m_end && co < p.N) {
                    const T* src = (const T*)p.dY + m * p.N + co;
                    if (vecA) v = ld4(src);
                    else { v.x = ld1(src); if (co + 1 < p.N) v.y = ld1(src + 1); if (co + 2 < p.N) v.z = ld1(src + 2); if (co + 3 < p.N) v.w = ld1(src + 3); }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = tid + i * 256;
            float4 v = f4zero();
            if (idx < B_F4) {
                const int row = idx / (BJ / 4), c = (idx % (BJ / 4)) * 4;
                const int64_t m = m0 + row;
                const int ci = ci0 + c;
                if (m < m_end && ci < p.K) {
                    const T* src = (const T*)p.X + m * p.K + ci;
                    if (vecB) v = ld4(src);
                    else { v.x = ld1(src); if (ci + 1 < p.K) v.y = ld1(src + 1); if (ci + 2 < p.K) v.z = ld1(src + 2); if (ci + 3 < p.K) v.w = ld1(src + 3); }
                    if (do_xf) {
                        v = xform4(v, ld4(sScale + c), ld4(sShift + c), p.in_act);
                        if (!vecB) { if (ci + 1 >= p.K) v.y = 0.f; if (ci + 2 >= p.K) v.z = 0.f; if (ci + 3 >= p.K) v.w = 0.f; }
                    }
                }
            }
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int idx = tid + i * 256;
            if (idx < A_F4) st4(sA + buf * KC * BI + idx * 4, ra[i]);       // [row][c] is exactly idx*4
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = tid + i * 256;
            if (idx < B_F4) st4(sB + buf * KC * BJ + idx * 4, rb[i]);
        }
    };

    const int64_t nchunks = (m_end - m_begin + KC - 1) / KC;
    if (nchunks > 0) {
        gload(m_begin);
        lstore(0);
    }
    __syncthreads();
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int buf = (int)(ch & 1);
        if (ch + 1 < nchunks) gload(m_begin + (ch + 1) * KC);
        const float* a_base = sA + buf * KC * BI + (krow0 + kk) * BI + ioff + li;
        const float* b_base = sB + buf * KC * BJ + (krow0 + kk) * BJ + joff + li;
#pragma unroll
        for (int kp = 0; kp < KROWS / 2; ++kp) {
            float af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = a_base[kp * 2 * BI + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = b_base[kp * 2 * BJ + j * 32];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (ch + 1 < nchunks) lstore(buf ^ 1);
        __syncthreads();
    }

    float* dst = p.partial + (int64_t)blockIdx.z * p.N * p.K;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int ci = ci0 + joff + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + ioff + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = acc[i][j][r];
                }
            }
    } else {
        float* red = smem;                 // [3][16][64] (12 KB <= staging area)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                __syncthreads();
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[i][j][r];
                }
                __syncthreads();
                if (wave == 0) {
                    const int ci = ci0 + j * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = ((acc[i][j][r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                        const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                        if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = v;
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient v2: the same decomposition fed by LDS-DMA (3-stage ring of 16-row chunks, counted
// vmcnt, raw barriers).  The chunk image [row][channel] is linear, i.e. exactly the lane-linear order a
// DMA instruction writes, and fragments are read along the channel axis (conflict-free ds_read_b32), so no
// swizzle is needed.  Rows past the block's M-slice take dY from a 16-B zero buffer (product = 0 * finite),
// X rows are clamped to a valid row; BN-apply + activation is applied to X when its fragment is read.
// Requires N % 4 == 0 and K % 4 == 0 (16-B aligned rows) and a non-hswish view.
// ------------------------------------------------------------------------------------------------
template <int MODE, int TI, int TJ, int X6 = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TI * TJ >= 8 ? 2 : 1, 8))) void pw_wgrad_dma_kernel(WgradArgs p) {
    constexpr int KC = 16, S = 3;
    constexpr int BI = (MODE == 0 ? 64 : 32) * TI;
    constexpr int BJ = (MODE == 0 ? 64 : 32) * TJ;
    constexpr int A_ST = KC * BI, B_ST = KC * BJ, STAGE = A_ST + B_ST;      // floats
    constexpr int NA = A_ST / 256, NB = B_ST / 256, NL = NA + NB;           // 1-KiB DMA instructions per stage
    constexpr int LPW = (NL + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const WgBlock blk = wg_block(p);
    if (!blk.live) return;
    const int co0 = blk.bx * BI, ci0 = blk.by * BJ;
    const int64_t m_begin = (int64_t)blk.bz * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const bool has_xf = p.in_scale != nullptr;
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);

    const int ioff = MODE == 0 ? (wave >> 1) * 32 * TI : 0;
    const int joff = MODE == 0 ? (wave & 1) * 32 * TJ : 0;
    const int krow0 = MODE == 0 ? 0 : wave * (KC / 4);
    constexpr int KROWS = MODE == 0 ? KC : KC / 4;

    float sc[TJ], sh[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int ci = ci0 + joff + j * 32 + li;
        sc[j] = (has_xf && ci < p.K) ? p.in_scale[ci] : 1.f;
        sh[j] = (has_xf && ci < p.K) ? p.in_shift[ci] : 0.f;
    }

    // per-lane constants of this wave's DMA instructions
    const float* zero_src = reinterpret_cast<const float*>(&mny_zero16);
    // running sources (round 3, as in the NT GEMM): one pointer per slot, bumped by 16 rows per stage; rows past the block's M slice
    // select a fixed filler (dY: zeros, X: the slice's last row) — no per-instruction branch, no 64-bit multiply in the loop
    int d_row[LPW], d_lds[LPW], d_step[LPW];
    const float* d_cur[LPW];
    const float* d_past[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wave + 4 * i;
        if (j >= NL) j = NL - 1;
        const bool isA = j < NA;
        const int q = (isA ? j : j - NA) * 64 + lane;                // float4 index inside the tile
        const int W4 = (isA ? BI : BJ) / 4;
        d_row[i] = q / W4;
        const int c = (q % W4) * 4;
        d_lds[i] = isA ? j * 256 : A_ST + (j - NA) * 256;
        const bool ok = isA ? co0 + c < p.N : ci0 + c < p.K;
        const float* base = isA ? (const float*)p.dY : (const float*)p.X;
        const int stride = isA ? p.N : p.K, off = isA ? co0 + c : ci0 + c;
        d_cur[i] = ok ? base + (m_begin + d_row[i]) * stride + off : zero_src;
        d_past[i] = (ok && !isA) ? base + (m_end - 1) * stride + off : zero_src;
        d_step[i] = ok ? KC * stride : 0;
    }

    auto issue = [&](int64_t m0, int slot) {
        float* stage = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const float* src = (m0 + d_row[i] < m_end) ? d_cur[i] : d_past[i];
            d_cur[i] += d_step[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int slot) {
        const float* a_base = smem + slot * STAGE + (krow0 + kk) * BI + ioff + li;
        const float* b_base = smem + slot * STAGE + A_ST + (krow0 + kk) * BJ + joff + li;
        if constexpr (X6 != 0 && MODE == 0) {
            // six-product bf16 form (see x6_split): the lane's eight rows of the chunk are rows 2*kp + kk, the ones the fp32 path
            // feeds one MFMA step at a time; a whole 16-row chunk is one v_mfma_f32_32x32x16_bf16 per partial product
            bf16x8_t ah[TI], am[TI], al[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const v4f_t x0 = {a_base[0 * BI + i * 32], a_base[2 * BI + i * 32], a_base[4 * BI + i * 32], a_base[6 * BI + i * 32]};
                const v4f_t x1 = {a_base[8 * BI + i * 32], a_base[10 * BI + i * 32], a_base[12 * BI + i * 32], a_base[14 * BI + i * 32]};
                x6_split(x0, x1, ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float z[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(b_base[2 * e * BJ + j * 32], sc[j], sh[j]);
                    z[e] = fminf(fmaxf(t, slope * t), hi);
                }
                bf16x8_t bh, bm, bl;
                x6_split(v4f_t{z[0], z[1], z[2], z[3]}, v4f_t{z[4], z[5], z[6], z[7]}, bh, bm, bl);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int kp = 0; kp < KROWS / 2; ++kp) {
            float af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = a_base[kp * 2 * BI + i * 32];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const float z = fmaf(b_base[kp * 2 * BJ + j * 32], sc[j], sh[j]);
                bf[j] = fminf(fmaxf(z, slope * z), hi);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const int total = (int)((m_end - m_begin + KC - 1) / KC);
    int i_t = 0, i_slot = 0, c_slot = 0;
    auto issue_next = [&]() {
        issue(m_begin + (int64_t)i_t * KC, i_slot);
        ++i_t;
        if (++i_slot == S) i_slot = 0;
    };
    const int pre = total < S - 1 ? total : S - 1;
    for (int t = 0; t < pre; ++t) issue_next();
    const int steady = total - pre;
    for (int t = 0; t < steady; ++t) {
        wait_vmcnt<LPW*(S - 2)>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        compute(c_slot);
        if (++c_slot == S) c_slot = 0;
    }
    for (int t = 0; t < pre; ++t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(c_slot);
        if (++c_slot == S) c_slot = 0;
    }

    float* dst = p.partial + (int64_t)blk.bz * p.N * p.K;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int ci = ci0 + joff + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + ioff + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = acc[i][j][r];
                }
            }
    } else {
        float* red = smem;                 // [3][16][64] (12 KB <= ring)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                __syncthreads();
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[i][j][r];
                }
                __syncthreads();
                if (wave == 0) {
                    const int ci = ci0 + j * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = ((acc[i][j][r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                        const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                        if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = v;
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16 operands (dY, X in bf16): same tile decomposition and LDS-DMA ring, on
// v_mfma_f32_32x32x16_bf16.  The MFMA wants, per lane, 8 consecutive REDUCTION indices (rows m) of one
// channel, but the chunk image is [row][channel] with channels contiguous — exactly the case gfx950's
// transposing LDS read exists for: ds_read_b64_tr_b16 hands lane c of a 16-lane group the 4 rows of column c of a
// [4 rows][16 channels] block (each lane supplies the address of 4 contiguous channels of row i/4), so two such
// reads build one operand with no shuffles.  BN-apply + activation of X runs on the transposed fragment, where
// a lane holds ONE channel (scalar scale/shift per lane).  MODE 0: 16-row chunks, 2x2 waves over the tile;
// MODE 1: 64-row chunks, each wave a 16-row quarter of the whole (thin) dW, combined through LDS at the end.
// Requires N % 8 == 0 and K % 8 == 0 (16-B chunks of 8 bf16); anything else takes the register-staged kernel.
// ------------------------------------------------------------------------------------------------
typedef short v4s_t __attribute__((ext_vector_type(4)));

template <int MODE, int TI, int TJ, int XF>
__global__ __launch_bounds__(256) void pw_wgrad_bf16_kernel(WgradArgs p) {
    constexpr int KC = MODE == 0 ? 16 : 64, S = 3;
    constexpr int BI = (MODE == 0 ? 64 : 32) * TI;
    constexpr int BJ = (MODE == 0 ? 64 : 32) * TJ;
    constexpr int A_ST = KC * BI, B_ST = KC * BJ, STAGE = A_ST + B_ST;      // bf16 elements
    constexpr int NA = A_ST / 512, NB = B_ST / 512, NL = NA + NB;           // 1-KiB DMA instructions per stage
    constexpr int LPW = (NL + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_f);
    const bf16_t* pX = (const bf16_t*)p.X;
    const bf16_t* pdY = (const bf16_t*)p.dY;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const WgBlock blk = wg_block(p);
    if (!blk.live) return;
    const int co0 = blk.bx * BI, ci0 = blk.by * BJ;
    const int64_t m_begin = (int64_t)blk.bz * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const bool has_xf = p.in_scale != nullptr;
    const float slope = act_slope(p.in_act), hi = act_hi(p.in_act);

    const int ioff = MODE == 0 ? (wave >> 1) * 32 * TI : 0;
    const int joff = MODE == 0 ? (wave & 1) * 32 * TJ : 0;
    const int krow0 = MODE == 0 ? 0 : wave * 16;

    float sc[TJ], sh[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int ci = ci0 + joff + j * 32 + li;
        sc[j] = (has_xf && ci < p.K) ? p.in_scale[ci] : 1.f;
        sh[j] = (has_xf && ci < p.K) ? p.in_shift[ci] : 0.f;
    }

    const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(&mny_zero16);
    // running sources (round 3): see pw_wgrad_dma_kernel
    int d_row[LPW], d_lds[LPW], d_step[LPW];
    const bf16_t* d_cur[LPW];
    const bf16_t* d_past[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wave + 4 * i;
        if (j >= NL) j = NL - 1;
        const bool isA = j < NA;
        const int q = (isA ? j : j - NA) * 64 + lane;                // 16-B chunk index inside the tile
        const int W8 = (isA ? BI : BJ) / 8;
        d_row[i] = q / W8;
        const int c = (q % W8) * 8;
        d_lds[i] = isA ? j * 512 : A_ST + (j - NA) * 512;            // bf16 elements
        const bool ok = isA ? co0 + c < p.N : ci0 + c < p.K;
        const bf16_t* base = isA ? pdY : pX;
        const int stride = isA ? p.N : p.K, off = isA ? co0 + c : ci0 + c;
        d_cur[i] = ok ? base + (m_begin + d_row[i]) * stride + off : zero_src;
        d_past[i] = (ok && !isA) ? base + (m_end - 1) * stride + off : zero_src;
        d_step[i] = ok ? KC * stride : 0;
    }

    auto issue = [&](int64_t m0, int slot) {
        bf16_t* stage = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const bf16_t* src = (m0 + d_row[i] < m_end) ? d_cur[i] : d_past[i];
            d_cur[i] += d_step[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing read: 16-lane group g = (lane>>4)&1 covers channels 16g..16g+15 of the lane's 32-channel tile; lane i of
    // the group addresses row i/4, channels 4(i%4)..+3 of the [4][16] block and RECEIVES rows 0..3 of channel i.
    const int tr_row = (lane & 15) >> 2;
    const int tr_col = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto frag = [&](const bf16_t* tile, int pitch, int col0) -> uint4 {
        const bf16_t* base = tile + (krow0 + 8 * kk + tr_row) * pitch + col0 + tr_col;
        const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(base));
        const v4s_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(base + 4 * pitch));
        const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi4);
        return make_uint4(a.x, a.y, b.x, b.y);                    // k = 8*kk + 0..7 of this lane's channel
    };

    auto compute = [&](int slot) {
        const bf16_t* tA = smem + slot * STAGE;
        const bf16_t* tB = tA + A_ST;
        uint4 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = frag(tA, BI, ioff + i * 32);
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            uint4 u = frag(tB, BJ, joff + j * 32);
            if (XF != 0) {
                float z[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                              __uint_as_float(u.y & 0xffff0000u), __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                              __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float zz = fmaf(z[e], sc[j], sh[j]);
                    z[e] = XF == 1 ? __builtin_amdgcn_fmed3f(zz, slope * zz, hi) : zz * __builtin_amdgcn_fmed3f(zz + 3.f, 0.f, 6.f) * (1.f / 6.f);
                }
                u = make_uint4(pack_bf16x2(z[0], z[1]), pack_bf16x2(z[2], z[3]), pack_bf16x2(z[4], z[5]), pack_bf16x2(z[6], z[7]));
            }
            bf[j] = u;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bf[j]),
                                                                    acc[i][j], 0, 0, 0);
    };

    const int total = (int)((m_end - m_begin + KC - 1) / KC);
    int i_t = 0, i_slot = 0, c_slot = 0;
    auto issue_next = [&]() {
        issue(m_begin + (int64_t)i_t * KC, i_slot);
        ++i_t;
        if (++i_slot == S) i_slot = 0;
    };
    const int pre = total < S - 1 ? total : S - 1;
    for (int t = 0; t < pre; ++t) issue_next();
    const int steady = total - pre;
    for (int t = 0; t < steady; ++t) {
        wait_vmcnt<LPW*(S - 2)>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        compute(c_slot);
        if (++c_slot == S) c_slot = 0;
    }
    for (int t = 0; t < pre; ++t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        compute(c_slot);
        if (++c_slot == S) c_slot = 0;
    }

    float* dst = p.partial + (int64_t)blk.bz * p.N * p.K;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int ci = ci0 + joff + j * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + ioff + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = acc[i][j][r];
                }
            }
    } else {
        float* red = smem_f;               // [3][16][64] floats (12 KB <= ring)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                __syncthreads();
                if (wave > 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[i][j][r];
                }
                __syncthreads();
                if (wave == 0) {
                    const int ci = ci0 + j * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = ((acc[i][j][r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                        const int co = co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                        if (co < p.N && ci < p.K) dst[(int64_t)co * p.K + ci] = v;
                    }
                }
            }
    }
}

typedef void (*WgKernel)(WgradArgs);
static WgKernel wg_bf16_kernel(int mode, int TI, int TJ, int XF) {
    switch (mode * 100 + TI * 10 + TJ) {
#define MNY_W(MD, I, J) case MD * 100 + I * 10 + J: return XF == 0 ? (WgKernel)pw_wgrad_bf16_kernel<MD, I, J, 0> : XF == 1 ? (WgKernel)pw_wgrad_bf16_kernel<MD, I, J, 1> : (WgKernel)pw_wgrad_bf16_kernel<MD, I, J, 2>;
        MNY_W(0, 1, 1) MNY_W(0, 1, 2) MNY_W(0, 2, 1) MNY_W(0, 2, 2)
        MNY_W(1, 1, 1) MNY_W(1, 1, 2) MNY_W(1, 1, 3) MNY_W(1, 1, 4) MNY_W(1, 1, 5) MNY_W(1, 1, 6) MNY_W(1, 2, 1) MNY_W(1, 2, 2) MNY_W(1, 2, 3)
        MNY_W(1, 3, 1) MNY_W(1, 3, 2) MNY_W(1, 4, 1) MNY_W(1, 5, 1) MNY_W(1, 6, 1)
#undef MNY_W
        default: return nullptr;
    }
}

static WgKernel wg_dma_kernel(int mode, int TI, int TJ, int x6 = 0) {
    if (x6 && mode == 0) switch (TI * 10 + TJ) {
        case 11: return (WgKernel)pw_wgrad_dma_kernel<0, 1, 1, 1>; case 12: return (WgKernel)pw_wgrad_dma_kernel<0, 1, 2, 1>;
        case 21: return (WgKernel)pw_wgrad_dma_kernel<0, 2, 1, 1>; case 22: return (WgKernel)pw_wgrad_dma_kernel<0, 2, 2, 1>;
        case 24: return (WgKernel)pw_wgrad_dma_kernel<0, 2, 4, 1>;
    }
    switch (mode * 100 + TI * 10 + TJ) {
#define MNY_W(MD, I, J) case MD * 100 + I * 10 + J: return (WgKernel)pw_wgrad_dma_kernel<MD, I, J>;
        MNY_W(0, 1, 1) MNY_W(0, 1, 2) MNY_W(0, 2, 1) MNY_W(0, 2, 2)
        MNY_W(1, 1, 1) MNY_W(1, 1, 2) MNY_W(1, 1, 3) MNY_W(1, 1, 4) MNY_W(1, 1, 5) MNY_W(1, 1, 6) MNY_W(1, 2, 1) MNY_W(1, 2, 2) MNY_W(1, 2, 3)
        MNY_W(1, 3, 1) MNY_W(1, 3, 2) MNY_W(1, 4, 1) MNY_W(1, 5, 1) MNY_W(1, 6, 1)
#undef MNY_W
        default: return nullptr;
    }
}

struct WgPlan { int mode, TI, TJ, gx, gy, splits; int64_t rows_per_block; size_t lds, lds_dma; };

static int pick_block(int c) {      // 64 or 128: minimise the padded extent, ties -> 128
    const int p64 = (int)cdiv(c, 64) * 64, p128 = (int)cdiv(c, 128) * 128;
    return p128 <= p64 ? 128 : 64;
}

// wide (round 6): the six-product fp32 form with a 128 x 256 tile per workgroup (a wave = 2 x 4 blocks of 32: three operand cuts per 24 MFMAs instead
// of four — both operands of a weight gradient are activations, cut in the kernel —, and a 512 x 512 problem re-reads dY twice and X four times
// instead of four + four); 254 VGPRs at two waves per SIMD, 72 KB of LDS ring (two workgroups per CU).  The split count is the 128 x 128 plan's
// either way, so the partial-row count does not depend on which of the two a launch takes.  Measured (bracketed steps, same box): 512 x 512 @
// M 123 904 0.423 -> 0.399 ms, 1280 -> 512 @ 30 976 0.262 -> 0.248, the 33 weight gradients of the headline step 4.30 -> 4.20 ms
static bool wg_wide_shape(int64_t M, int K, int N) {
    static const int env = getenv("MNY_WG_TJ4") ? atoi(getenv("MNY_WG_TJ4")) : 1;     // (=0: the 128 x 128 tile, for A/B runs)
    return env != 0 && nt_x6(M, K, N) && K % 256 == 0 && N % 128 == 0;
}

static WgPlan wg_plan(int64_t M, int K, int N, bool bf16 = false, bool wide_ok = false) {
    WgPlan pl;
    const int nco = (int)cdiv(N, 32), nci = (int)cdiv(K, 32);
    int BI, BJ, KC;
    static const bool slice96 = getenv("MNY_WG_NOSLICE96") == nullptr;
    if (nco * nci <= 6) {
        pl.mode = 1; pl.TI = nco; pl.TJ = nci; BI = 32 * nco; BJ = 32 * nci; KC = bf16 ? 64 : 32;
        pl.gx = pl.gy = 1;
    } else if (slice96 && !bf16 && ((K == 96 && N >= 192) || (N == 96 && K >= 192))) {
        // a 96-wide side is 1.5 of the 64-wide units of the 2x2-wave tiles (25-33 % of the MFMAs multiply padding): slice the
        // other side instead and give every workgroup exact 96 x 64 tiles in the all-waves-share-the-tile mode
        pl.mode = 1; KC = 32;
        if (K == 96) { pl.TJ = 3; pl.TI = 2; } else { pl.TI = 3; pl.TJ = 2; }
        BI = 32 * pl.TI; BJ = 32 * pl.TJ;
        pl.gx = (int)cdiv(N, BI); pl.gy = (int)cdiv(K, BJ);
    } else {
        pl.mode = 0; BI = pick_block(N); BJ = pick_block(K); KC = 16;
        pl.TI = BI / 64; pl.TJ = BJ / 64;
        pl.gx = (int)cdiv(N, BI); pl.gy = (int)cdiv(K, BJ);
    }
    static const int wg_blocks = getenv("MNY_WG_BLOCKS") ? atoi(getenv("MNY_WG_BLOCKS")) : 1024;     // (round 6, next to the 128 x 256 tile: 1536 -> 1024: weight gradients 4.00 -> 3.99 ms, their combines 0.34 -> 0.31 ms, 0.4 GB fewer partial rows per step; 768: 4.32 ms)
    int64_t splits = wg_blocks / ((int64_t)pl.gx * pl.gy);
    if (splits < 1) splits = 1;
    const int64_t max_splits = cdiv(M, 4 * KC);
    if (splits > max_splits) splits = max_splits;
    if (splits > 768) splits = 768;
    const int64_t rpb = cdiv(cdiv(M, splits), KC) * KC;
    pl.rows_per_block = rpb;
    pl.splits = (int)cdiv(M, rpb);
    if (wide_ok && !bf16 && pl.mode == 0 && BI == 128 && wg_wide_shape(M, K, N)) { BJ = 256; pl.TJ = 4; pl.gy = K / 256; }
    size_t stage = (size_t)2 * (bf16 ? 32 : KC) * (BI + BJ) * sizeof(float);    // register-staged kernel: fp32 image, its own KC
    if (pl.mode == 0) stage = (size_t)2 * 16 * (BI + BJ) * sizeof(float);
    if (stage < 3 * 16 * 64 * sizeof(float)) stage = 3 * 16 * 64 * sizeof(float);
    pl.lds = stage + 2 * BJ * sizeof(float);
    pl.lds_dma = bf16 ? (size_t)3 * KC * (BI + BJ) * 2 : (size_t)3 * 16 * (BI + BJ) * sizeof(float);
    if (pl.lds_dma < 3 * 16 * 64 * sizeof(float)) pl.lds_dma = 3 * 16 * 64 * sizeof(float);
    return pl;
}

// partial rows [parts][n] -> out[n]: 32 outputs x 8 part-slices per block, fp64, fixed order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ parts, int nparts, int64_t n, float* __restrict__ out) {
    __shared__ double red[8][32];
    const int ol = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + ol;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int pidx = slice;                              // four partial rows in flight per thread: the loop is pure load latency otherwise
        for (; pidx + 24 < nparts; pidx += 32) {
            s0 += (double)parts[(int64_t)pidx * n + i];        s1 += (double)parts[(int64_t)(pidx + 8) * n + i];
            s2 += (double)parts[(int64_t)(pidx + 16) * n + i]; s3 += (double)parts[(int64_t)(pidx + 24) * n + i];
        }
        for (; pidx < nparts; pidx += 8) s0 += (double)parts[(int64_t)pidx * n + i];
    }
    red[slice][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && i < n) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][ol];
        out[i] = (float)s;
    }
}

// column sums of a [M][C] matrix -> partial rows; used for the head convs' bias gradient
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ y, float* __restrict__ parts, int64_t M, int C,
                                                     int64_t rows_per_block) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int slot = threadIdx.x >> 6;
    __shared__ float red[4][64];
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = min(m0 + rows_per_block, M);
    float s = 0.f;
    if (c < C)
        for (int64_t m = m0 + slot; m < m1; m += 4) s += ld1(y + m * C + c);
    red[slot][threadIdx.x & 63] = s;
    __syncthreads();
    if (slot == 0 && c < C) parts[(int64_t)blockIdx.x * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// dst[c][r] = src[r][c], destination row pitch Rp >= R, pad columns zeroed
template <typename T>
__global__ void transpose_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int Cc, int Rp) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < Rp && c < Cc) st1(dst + (int64_t)c * Rp + r, r < R ? tile[threadIdx.x][j] : 0.f);
    }
}

static int colsum_parts(int64_t M) { int64_t p = cdiv(M, 256); return (int)(p < 512 ? p : 512); }


// ================================================================================================
// Fused BatchNorm-backward for HBM-bound "expand" pointwise units (Cin <= 32, one 32-wide tile).
//
// Unit: Y = X W^T, A = act(sc*Y + sh) with batch statistics; G = dL/dA.  With dz = G*act'(sc*Y+sh) and the
// per-channel BN-backward coefficients (ca, cb, cc):  dY = ca*dz + cb*Y + cc.  Hence
//     dW = ca o (dz^T X) + cb o (W X^T X) + cc (x) colsum(X)          (Y^T X = W X^T X)
//     dX = dz (ca o W) + Y (cb o W) + (cc . W)
// so dY is never materialised and the unit's big output tensor is read 4 times instead of 7:
//   stage 1 (one pass over G, Y, X): sum dz, sum dz*yhat, P1 = dz^T X, Gram = X^T X, colsum(X)
//   finalize (tiny, fp64)          : dgamma, dbeta, ca/cb/cc, dW, B1 = ca o W^T, B2 = cb o W^T, bias = cc . W
//   stage 2 (one pass over G, Y)   : dX = dz B1^T + Y B2^T + bias (+ addend)
// Both streaming stages are LDS-DMA pipelines like the kernels above.
// ================================================================================================
struct BnwArgs {
    const float* G; const float* Y; const float* sc; const float* sh; int act; const float* mean; const float* invstd;
    const float* X; const float* xsc; const float* xsh; int xact;
    float* partial; unsigned long long* mask;   // mask[i*npairs + m/2]: bit (m&1)*32 + c%32 = [act'(z) is the "on" value]
    int64_t npairs; int64_t M; int K; int N; int64_t rows_per_block;
    int tiles_total;    // ceil(N/32); blockIdx.y selects a slice of TI column tiles (wide units run as two slices: occupancy)
};

// partial layout per split: P1[N*K] | Gram[K*K] | s1[N] | s2[N] | s3[K]   (bnw_stride: common.h — exdw.hip writes the same rows)

template <int TI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void pw_bnbwd_stage1_kernel(BnwArgs p) {
    constexpr int KC = 16, S = 3, BI = 32 * TI, BJ = 32;
    constexpr int G_ST = KC * BI, X_ST = KC * BJ, STAGE = 2 * G_ST + X_ST;   // floats: G | Y | X
    constexpr int NG = G_ST / 256, NX = X_ST / 256, NL = 2 * NG + NX;
    constexpr int LPW = (NL + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int64_t m_begin = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const bool has_xf = p.xsc != nullptr;
    const float xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    const float aslope = act_slope(p.act), ahi = act_hi(p.act);
    const int krow0 = wave * (KC / 4);
    const int tile_off = blockIdx.y * TI, n_off = tile_off * 32;      // this block's column slice
    const bool slice0 = blockIdx.y == 0;                              // Gram / colsum(X) are taken once

    float sc[TI], sh[TI], mu[TI], is[TI], s1[TI], s2[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int co = n_off + i * 32 + li;
        const bool ok = co < p.N;
        sc[i] = ok ? p.sc[co] : 0.f; sh[i] = ok ? p.sh[co] : 0.f; mu[i] = ok ? p.mean[co] : 0.f; is[i] = ok ? p.invstd[co] : 0.f;
        s1[i] = 0.f; s2[i] = 0.f;
    }
    const float xs = (has_xf && li < p.K) ? p.xsc[li] : 1.f, xh = (has_xf && li < p.K) ? p.xsh[li] : 0.f;
    float s3 = 0.f;

    const float* zero_src = reinterpret_cast<const float*>(&mny_zero16);
    int d_row[LPW], d_lds[LPW], d_kind[LPW];
    bool d_ok[LPW];
    int d_off[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wave + 4 * i;
        if (j >= NL) j = NL - 1;
        d_kind[i] = j < NG ? 0 : (j < 2 * NG ? 1 : 2);                    // 0: G, 1: Y, 2: X
        const int jj = d_kind[i] == 0 ? j : (d_kind[i] == 1 ? j - NG : j - 2 * NG);
        const int q = jj * 64 + lane;
        const int W4 = (d_kind[i] == 2 ? BJ : BI) / 4;
        d_row[i] = q / W4;
        const int c = (q % W4) * 4;
        d_lds[i] = (d_kind[i] == 0 ? 0 : (d_kind[i] == 1 ? G_ST : 2 * G_ST)) + jj * 256;
        d_ok[i] = d_kind[i] == 2 ? c < p.K : n_off + c < p.N;
        d_off[i] = d_kind[i] == 2 ? c : n_off + c;
    }
    auto issue = [&](int64_t m0, int slot) {
        float* stage = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int64_t m = m0 + d_row[i];
            const int64_t mc = m < m_end ? m : m_end - 1;
            const float* src = zero_src;
            if (d_ok[i]) {
                if (d_kind[i] == 0) src = m < m_end ? p.G + m * p.N + d_off[i] : zero_src;   // dz = 0 past the slice
                else if (d_kind[i] == 1) src = p.Y + mc * p.N + d_off[i];
                else src = p.X + mc * p.K + d_off[i];
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };

    f32x16 acc[TI], gram;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) gram[r] = 0.f;

    auto compute = [&](int64_t m0, int slot) {
        const float* gb = smem + slot * STAGE + (krow0 + kk) * BI + li;
        const float* yb = gb + G_ST;
        const float* xb = smem + slot * STAGE + 2 * G_ST + (krow0 + kk) * BJ + li;
#pragma unroll
        for (int kp = 0; kp < KC / 8; ++kp) {
            const bool rok = m0 + krow0 + kp * 2 + kk < m_end;
            const float xz = fmaf(xb[kp * 2 * BJ], xs, xh);
            const float b = rok ? fminf(fmaxf(xz, xslope * xz), xhi) : 0.f;     // rows past the slice contribute nothing
            s3 += b;
            float af[TI];
            const int64_t pair = (m0 + krow0 + kp * 2) >> 1;                     // rows (2*pair, 2*pair+1) <-> kk
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const float g = gb[kp * 2 * BI + i * 32], y = yb[kp * 2 * BI + i * 32];
                const float z = fmaf(y, sc[i], sh[i]);
                const bool on = z > 0.f && z < ahi;                                // act' = on ? 1 : slope
                const unsigned long long bal = __ballot(on);
                if (lane == 0 && m0 + krow0 + kp * 2 < m_end && tile_off + i < p.tiles_total) p.mask[(int64_t)(tile_off + i) * p.npairs + pair] = bal;
                const float dz = on ? g : g * aslope;
                s1[i] += dz;
                s2[i] = fmaf(dz, (y - mu[i]) * is[i], s2[i]);
                af[i] = dz;
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], b, acc[i], 0, 0, 0);
            if (slice0) gram = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, gram, 0, 0, 0);
        }
    };

    const int total = (int)((m_end - m_begin + KC - 1) / KC);
    int i_t = 0, i_slot = 0, c_slot = 0, c_t = 0;
    auto issue_next = [&]() { issue(m_begin + (int64_t)i_t * KC, i_slot); ++i_t; if (++i_slot == S) i_slot = 0; };
    auto consume = [&]() { compute(m_begin + (int64_t)c_t * KC, c_slot); ++c_t; if (++c_slot == S) c_slot = 0; };
    const int pre = total < S - 1 ? total : S - 1;
    for (int t = 0; t < pre; ++t) issue_next();
    for (int t = 0; t < total - pre; ++t) {
        wait_vmcnt<LPW*(S - 2)>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        consume();
    }
    for (int t = 0; t < pre; ++t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        consume();
    }

    // ---- block reduction over the 4 waves (fixed order) and the two k-halves, then one partial row
    float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(p.N, p.K);
    float* red = smem;                         // [3][16][64]
    auto reduce_tile = [&](f32x16& t, float* out, int rows, int cols, int ld, int row0) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = t[r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = ((t[r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < rows && li < cols) out[(int64_t)row * ld + li] = v;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < TI; ++i) reduce_tile(acc[i], dst, p.N, p.K, p.K, n_off + i * 32);
    if (slice0) reduce_tile(gram, dst + (int64_t)p.N * p.K, p.K, p.K, p.K, 0);
    // vectors: lanes l and l^32 hold the same channel
    __syncthreads();
    float* vred = smem;                        // [4][2*TI+1][32]
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const float a = s1[i] + __shfl_xor(s1[i], 32), b2 = s2[i] + __shfl_xor(s2[i], 32);
        if (kk == 0) { vred[(wave * (2 * TI + 1) + i) * 32 + li] = a; vred[(wave * (2 * TI + 1) + TI + i) * 32 + li] = b2; }
    }
    {
        const float c3 = s3 + __shfl_xor(s3, 32);
        if (kk == 0) vred[(wave * (2 * TI + 1) + 2 * TI) * 32 + li] = c3;
    }
    __syncthreads();
    float* vdst = dst + (int64_t)p.N * p.K + (int64_t)p.K * p.K;
    for (int e = tid; e < (2 * TI + 1) * 32; e += 256) {
        const int v = e / 32, l = e % 32;
        float a = 0.f;
        for (int w = 0; w < 4; ++w) a += vred[(w * (2 * TI + 1) + v) * 32 + l];
        if (v < TI) { const int co = n_off + v * 32 + l; if (co < p.N) vdst[co] = a; }
        else if (v < 2 * TI) { const int co = n_off + (v - TI) * 32 + l; if (co < p.N) vdst[p.N + co] = a; }
        else if (l < p.K && slice0) vdst[2 * p.N + l] = a;
    }
}

// Stage 1, second generation (round 3).  Same tiling, LDS-DMA ring, lane mapping, partial-row layout and arithmetic (order
// included: bit-identical results) as pw_bnbwd_stage1_kernel; what changed is the instruction stream.  The counters of the first
// generation (profiles/r03_pmc_pw_bnbwd.txt, 16 -> 96 @ 176x176) showed 15 scalar and 20 vector instructions per MFMA, 44 % of the
// wave cycles stalled at issue and the matrix pipe 30 % busy at 4.45 TB/s: every LDS read sat in its own basic block behind a
// `lane == 0` mask store (s_cbranch_execz + lgkmcnt(0) per read), the DMA issue went through per-instruction kind / range branches.
// Here the row loop body is ONE basic block: all 2 + 4 TI fragment reads of a stage are issued together, the ballots of a stage are
// kept in scalar registers and leave through one exec-masked group of 16-byte stores, the DMA sources are running pointers
// (select, not branch, for the rows past the slice), the Gram MFMA is unconditional (its column-slice twin is simply not stored).
// T: storage type of G, Y, X (float, or bf16_t for bf16-storage plans: the LDS image holds bf16, a 16-byte DMA chunk is 8 elements, the
// arithmetic is the same fp32; K % 8 == 0 and N % 8 == 0 there)
// S: ring depth.  A stage is 16 rows whatever the storage type, i.e. HALF the bytes with bf16 storage (2 x 5 KB per workgroup in flight at
// three stages; the bf16 instantiation runs 2.9 TB/s where the fp32 one runs 4.3-5.2).  Six stages (MNY_BNW_RING=6) were measured in round 5
// and changed nothing: bytes in flight are not what holds this kernel back.
template <int TI, typename T = float, int S = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 8))) void pw_bnbwd_stage1b_kernel(BnwArgs p) {
    constexpr int KC = 16, BI = 32 * TI, BJ = 32;
    constexpr int EPC = 16 / (int)sizeof(T), EPI = 64 * EPC;                  // elements per 16-byte chunk / per 1-KiB DMA instruction
    constexpr int G_ST = KC * BI, X_ST = KC * BJ, STAGE = 2 * G_ST + X_ST;   // elements of T: G | Y | X
    constexpr int NG = (G_ST + EPI - 1) / EPI, NX = (X_ST + EPI - 1) / EPI, NL = 2 * NG + NX;
    static_assert(G_ST % EPI == 0 && X_ST % EPI == 0, "whole DMA instructions per tile");
    constexpr int LPW = (NL + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    T* const smem = reinterpret_cast<T*>(smem_f);
    const T* const pG = reinterpret_cast<const T*>(p.G);
    const T* const pY = reinterpret_cast<const T*>(p.Y);
    const T* const pX = reinterpret_cast<const T*>(p.X);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int64_t m_begin = (int64_t)blockIdx.x * p.rows_per_block;
    const int64_t m_end = min(m_begin + p.rows_per_block, p.M);
    const bool has_xf = p.xsc != nullptr;
    const float xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    const float aslope = act_slope(p.act), ahi = act_hi(p.act);
    const int krow0 = wave * (KC / 4);
    const int tile_off = blockIdx.y * TI, n_off = tile_off * 32;
    const bool slice0 = blockIdx.y == 0;

    float sc[TI], sh[TI], mu[TI], is[TI], s1[TI], s2[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int co = n_off + i * 32 + li;
        const bool ok = co < p.N;
        sc[i] = ok ? p.sc[co] : 0.f; sh[i] = ok ? p.sh[co] : 0.f; mu[i] = ok ? p.mean[co] : 0.f; is[i] = ok ? p.invstd[co] : 0.f;
        s1[i] = 0.f; s2[i] = 0.f;
    }
    const float xs = (has_xf && li < p.K) ? p.xsc[li] : 1.f, xh = (has_xf && li < p.K) ? p.xsh[li] : 0.f;
    float s3 = 0.f;

    // DMA instruction slots of this wave: a running source pointer per slot (row m_begin + d_row of its tensor), the clamp target
    // for rows past the slice (Y, X: the slice's last row — finite filler, its products are annihilated by dz = 0 / b = 0; G: zeros)
    const T* zero_src = reinterpret_cast<const T*>(&mny_zero16);
    const T* d_cur[LPW];
    const T* d_past[LPW];
    int d_row[LPW], d_lds[LPW], d_step[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wave + 4 * i;
        if (j >= NL) j = NL - 1;
        const int kind = j < NG ? 0 : (j < 2 * NG ? 1 : 2);              // 0: G, 1: Y, 2: X
        const int jj = kind == 0 ? j : (kind == 1 ? j - NG : j - 2 * NG);
        const int q = jj * 64 + lane;                                     // 16-byte chunk index inside the tile
        const int W4 = (kind == 2 ? BJ : BI) / EPC;
        d_row[i] = q / W4;
        const int c = (q % W4) * EPC;
        d_lds[i] = (kind == 0 ? 0 : (kind == 1 ? G_ST : 2 * G_ST)) + jj * EPI;
        const bool ok = kind == 2 ? c < p.K : n_off + c < p.N;
        const T* base = kind == 0 ? pG : (kind == 1 ? pY : pX);
        const int stride = kind == 2 ? p.K : p.N;
        const int off = kind == 2 ? c : n_off + c;
        d_cur[i] = ok ? base + (m_begin + d_row[i]) * stride + off : zero_src;
        d_past[i] = (ok && kind != 0) ? base + (m_end - 1) * stride + off : zero_src;
        d_step[i] = ok ? KC * stride : 0;
    }
    auto issue = [&](int64_t m0, int slot) {
        T* stage = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const T* src = (m0 + d_row[i] < m_end) ? d_cur[i] : d_past[i];
            d_cur[i] += d_step[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };

    f32x16 acc[TI], gram;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) gram[r] = 0.f;

    auto compute = [&](int64_t m0, int slot) {
        const T* gb = smem + slot * STAGE + (krow0 + kk) * BI + li;
        const T* yb = gb + G_ST;
        const T* xb = smem + slot * STAGE + 2 * G_ST + (krow0 + kk) * BJ + li;
        float xr[KC / 8], gr[KC / 8][TI], yr[KC / 8][TI];
#pragma unroll
        for (int kp = 0; kp < KC / 8; ++kp) {                             // every fragment read of the stage, one wait
            xr[kp] = ld1(xb + kp * 2 * BJ);
#pragma unroll
            for (int i = 0; i < TI; ++i) { gr[kp][i] = ld1(gb + kp * 2 * BI + i * 32); yr[kp][i] = ld1(yb + kp * 2 * BI + i * 32); }
        }
        unsigned long long bal[KC / 8][TI];
#pragma unroll
        for (int kp = 0; kp < KC / 8; ++kp) {
            const bool rok = m0 + krow0 + kp * 2 + kk < m_end;
            const float xz = fmaf(xr[kp], xs, xh);
            const float b = rok ? fminf(fmaxf(xz, xslope * xz), xhi) : 0.f;     // rows past the slice contribute nothing
            s3 += b;
            float af[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const float g = gr[kp][i], y = yr[kp][i];
                const float z = fmaf(y, sc[i], sh[i]);
                const bool on = z > 0.f && z < ahi;                                // act' = on ? 1 : slope
                bal[kp][i] = __ballot(on);
                const float dz = on ? g : g * aslope;
                s1[i] += dz;
                s2[i] = fmaf(dz, (y - mu[i]) * is[i], s2[i]);
                af[i] = dz;
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], b, acc[i], 0, 0, 0);
            gram = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, gram, 0, 0, 0);
        }
        // the stage's mask words (rows 2 pair, 2 pair + 1 <-> the two lane halves; kp = 0, 1 are consecutive pairs, pair0 is even):
        // one 16-byte store per column tile from lane 0
        if (lane == 0) {
            const int64_t pair0 = (m0 + krow0) >> 1;
            const bool full = m0 + krow0 + 2 < m_end;                              // both pairs of this wave start inside the slice
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                if (tile_off + i < p.tiles_total) {
                    unsigned long long* dstw = p.mask + (int64_t)(tile_off + i) * p.npairs + pair0;
                    if (full) *reinterpret_cast<ulonglong2*>(dstw) = make_ulonglong2(bal[0][i], bal[1][i]);
                    else if (m0 + krow0 < m_end) dstw[0] = bal[0][i];
                }
            }
        }
    };

    const int total = (int)((m_end - m_begin + KC - 1) / KC);
    int i_t = 0, i_slot = 0, c_slot = 0, c_t = 0;
    auto issue_next = [&]() { issue(m_begin + (int64_t)i_t * KC, i_slot); ++i_t; if (++i_slot == S) i_slot = 0; };
    auto consume = [&]() { compute(m_begin + (int64_t)c_t * KC, c_slot); ++c_t; if (++c_slot == S) c_slot = 0; };
    const int pre = total < S - 1 ? total : S - 1;
    for (int t = 0; t < pre; ++t) issue_next();
    for (int t = 0; t < total - pre; ++t) {
        wait_vmcnt<LPW*(S - 2)>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        consume();
    }
    for (int t = 0; t < pre; ++t) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        consume();
    }

    // ---- block reduction over the 4 waves (fixed order) and the two k-halves, then one partial row (as the first generation)
    float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(p.N, p.K);
    float* red = smem_f;                       // [3][16][64]
    auto reduce_tile = [&](f32x16& t, float* out, int rows, int cols, int ld, int row0) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = t[r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = ((t[r] + red[(0 * 16 + r) * 64 + lane]) + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane];
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < rows && li < cols) out[(int64_t)row * ld + li] = v;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < TI; ++i) reduce_tile(acc[i], dst, p.N, p.K, p.K, n_off + i * 32);
    if (slice0) reduce_tile(gram, dst + (int64_t)p.N * p.K, p.K, p.K, p.K, 0);
    __syncthreads();
    float* vred = smem_f;                      // [4][2*TI+1][32]
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const float a = s1[i] + __shfl_xor(s1[i], 32), b2 = s2[i] + __shfl_xor(s2[i], 32);
        if (kk == 0) { vred[(wave * (2 * TI + 1) + i) * 32 + li] = a; vred[(wave * (2 * TI + 1) + TI + i) * 32 + li] = b2; }
    }
    {
        const float c3 = s3 + __shfl_xor(s3, 32);
        if (kk == 0) vred[(wave * (2 * TI + 1) + 2 * TI) * 32 + li] = c3;
    }
    __syncthreads();
    float* vdst = dst + (int64_t)p.N * p.K + (int64_t)p.K * p.K;
    for (int e = tid; e < (2 * TI + 1) * 32; e += 256) {
        const int v = e / 32, l = e % 32;
        float a = 0.f;
        for (int w = 0; w < 4; ++w) a += vred[(w * (2 * TI + 1) + v) * 32 + l];
        if (v < TI) { const int co = n_off + v * 32 + l; if (co < p.N) vdst[co] = a; }
        else if (v < 2 * TI) { const int co = n_off + (v - TI) * 32 + l; if (co < p.N) vdst[p.N + co] = a; }
        else if (l < p.K && slice0) vdst[2 * p.N + l] = a;
    }
}

static inline size_t bnw_finalize_lds(int N, int K) { return (size_t)3 * N * sizeof(double) + ((size_t)N * K + (size_t)K * K) * sizeof(float); }   // <= 33 KB (N <= 192, K <= 32)

// finalize.  red = [P1 | Gram | s1 | s2 | s3] summed over splits.  Every block recomputes the N coefficient triples (cheap) and
// takes a grid-stride share of the fp64 element loops (one block took 80 us per unit).
__global__ __launch_bounds__(256) void pw_bnbwd_finalize_kernel(const float* __restrict__ red, const float* __restrict__ W,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, double count, int N, int K,
                                                                float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ B1, float* __restrict__ Q, float* __restrict__ bias) {
    extern __shared__ double sd[];             // ca[N] cb[N] cc[N] | W[N][K] | Gram[K][K] (floats): the element loops below are serial
    double* ca = sd; double* cb = sd + N; double* cc = sd + 2 * N;          // fp64 chains over W and the Gram matrix; out of L2 they took 25 us per launch
    float* sW = reinterpret_cast<float*>(sd + 3 * N); float* sG = sW + N * K;
    for (int e = threadIdx.x; e < N * K; e += blockDim.x) sW[e] = W[e];
    for (int e = threadIdx.x; e < K * K; e += blockDim.x) sG[e] = red[(int64_t)N * K + e];
    const float* P1 = red; const float* Gm = red + (int64_t)N * K; const float* s1 = Gm + (int64_t)K * K;
    const float* s2 = s1 + N; const float* s3 = s2 + N;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const double a = (double)gamma[n] * (double)invstd[n];
        const double b = -a * (double)invstd[n] * (double)s2[n] / count;
        ca[n] = a; cb[n] = b; cc[n] = -a * (double)s1[n] / count - b * (double)mean[n];
        if (blockIdx.x == 0) { dbeta[n] = s1[n]; dgamma[n] = s2[n]; }
    }
    __syncthreads();
    for (int e = gtid; e < N * K; e += gsz) {
        const int n = e / K, k = e % K;
        double wg = 0.0;
        for (int j = 0; j < K; ++j) wg += (double)sW[n * K + j] * (double)sG[j * K + k];
        dW[e] = (float)(ca[n] * (double)P1[e] + cb[n] * wg + cc[n] * (double)s3[k]);
        B1[(int64_t)k * N + n] = (float)(ca[n] * (double)sW[e]);     // [K][N]: rows = dX columns, contraction over n
    }
    for (int e = gtid; e < K * K; e += gsz) {                         // Y (cb o W) = X (W^T diag(cb) W): Q[kc][k], symmetric
        const int kc = e / K, k = e % K;
        double a = 0.0;
        for (int n = 0; n < N; ++n) a += cb[n] * (double)sW[n * K + kc] * (double)sW[n * K + k];
        Q[e] = (float)a;
    }
    for (int k = gtid; k < K; k += gsz) {
        double a = 0.0;
        for (int n = 0; n < N; ++n) a += cc[n] * (double)sW[n * K + k];
        bias[k] = (float)a;
    }
}

// stage 2: dX[M,Kc] = dz[M,N] B1[Kc,N]^T + act(X)[M,Kc] Q[Kc,Kc]^T + bias (+addend), dz = G * (mask ? 1 : slope).
// Streams ONLY G (plus the 1-bit mask and the thin X).  Tile: 64 rows x 32-wide k-steps (full 128-B lines); waves
// 2 (row halves) x 2 (k-chunk parity), summed once per tile; the last k-step of every tile is the (X, Q) product.
struct BndArgs {
    const float* G; const unsigned long long* mask; int act;
    const float* X; const float* xsc; const float* xsh; int xact;
    const float* B1; const float* Q; const float* bias; const float* addend; float* C;
    int64_t npairs; int64_t M; int N; int Kc; int TI; int m_tiles, tiles_per_block;
    // RED (pw_bnbwd_dgrad2_kernel<.., true>, round 6): C is the complete output gradient of the conv+BN+act unit whose raw output is rY: its
    // BN-backward sums leave as one partial row per workgroup, red[gridDim.x][2][Kc]
    const float* rY; const float* r_scale; const float* r_shift; const float* r_mean; const float* r_invstd; int r_act; float* red;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void pw_bnbwd_dgrad_kernel(BndArgs p) {
    constexpr int BMS = 64, BKD = 32, S = 3;
    constexpr int A_ST = BMS * BKD, B_ST = 32 * BKD, STAGE = A_ST + B_ST;            // G(or X) | B1(or Q)   (12 KB)
    constexpr int NA = A_ST / 256, NB = B_ST / 256, NL = NA + NB;                    // 8 + 4
    constexpr int LPW = NL / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nkg = (p.N + BKD - 1) / BKD, nk = nkg + 1;                             // + the (X, Q) step
    float* sXs = smem + S * STAGE;                                                   // [32] input-view scale / shift
    float* sXh = sXs + 32;
    float* xred = sXh + 32;                                                          // [2][16][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 1, wk = wv & 1;
    const int lrow = lane & 31, khalf = lane >> 5;
    if (tid < 32) {
        sXs[tid] = (p.xsc && tid < p.Kc) ? p.xsc[tid] : (tid < p.Kc ? 1.f : 0.f);
        sXh[tid] = (p.xsc && tid < p.Kc) ? p.xsh[tid] : 0.f;
    }
    const int mt_begin = blockIdx.x * p.tiles_per_block;
    const int mt_end = min(mt_begin + p.tiles_per_block, p.m_tiles);
    const int total = (mt_end - mt_begin) * nk;

    const float* zero_src = reinterpret_cast<const float*>(&mny_zero16);
    const int drow = lane >> 3, dpos = lane & 7;
    int d_row[LPW], d_k[LPW], d_lds[LPW], d_kind[LPW];           // kind 0: streamed operand, 1: weights, 2: mask words
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        int j = wv + 4 * i;
        if (j >= NL) j = NL - 1;
        d_kind[i] = j < NA ? 0 : (j < NA + NB ? 1 : 2);
        const int jj = d_kind[i] == 0 ? j : j - NA;
        const int row = jj * 8 + drow;
        d_row[i] = row;
        d_k[i] = (dpos ^ (row & 7)) * 4;
        d_lds[i] = d_kind[i] == 2 ? A_ST + B_ST : (d_kind[i] == 0 ? 0 : A_ST) + jj * 256;
    }
    auto issue = [&](int mt, int kt, int slot) {
        float* stage = smem + slot * STAGE;
        const bool xstep = kt == nkg;
        const int width = xstep ? p.Kc : p.N;                 // row length of the streamed operand / contraction length
        const int k0 = xstep ? 0 : kt * BKD;
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int k = k0 + d_k[i];
            const bool kok = k < width;
            const float* src;
            if (d_kind[i] == 0) {
                int m = mt * BMS + d_row[i];
                if (m >= (int)p.M) m = (int)p.M - 1;
                src = (xstep ? p.X : p.G) + (int64_t)m * width + (kok ? k : 0);
            } else if (d_kind[i] == 1) {
                const int n = d_row[i];
                src = (n < p.Kc && kok) ? (xstep ? p.Q : p.B1) + (int64_t)n * width + k : zero_src;
            } else {
                src = zero_src;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + d_lds[i]), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int swz = lrow & 7;
    const float aslope = act_slope(p.act);
    const float xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    // The act' mask word of a step is an ordinary global load.  Loaded where it is used, the compiler has to drain the whole
    // VM queue (vmcnt(0)) in front of it — including the DMA stages just issued — so every step exposed a full memory latency
    // (2.5 TB/s).  It is therefore fetched ONE STEP AHEAD, before that step's DMA batch is issued: by the time it is consumed
    // the counted wait at the top of the next iteration has already covered it and the ring stays two stages deep.
    // (Staging the words through the DMA ring instead was measured slower: 2.13 vs 1.74 ms on the 16->96 unit.)
    auto compute = [&](int mt, int kt, int slot, unsigned long long mw) {
        const float* st = smem + slot * STAGE;
        const float* a_row = st + (wm * 32 + lrow) * BKD;
        const float* b_row = st + A_ST + lrow * BKD;
        const bool xstep = kt == nkg;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const int chunk = (kc * 2 + wk) * 2 + khalf;
            const int o = (chunk ^ swz) << 2;
            v4f_t a = lds_read_f4(a_row + o);                      // raw LDS reads: see "LDS reads the compiler cannot see"
            v4f_t b = lds_read_f4(b_row + o);
            v4f_t xs = lds_read_f4(sXs + chunk * 4), xh = lds_read_f4(sXh + chunk * 4);
            MNY_LGKM_WAIT(a);
            MNY_LGKM_DEP(b); MNY_LGKM_DEP(xs); MNY_LGKM_DEP(xh);
            if (xstep) {
                float z;
                z = fmaf(a.x, xs.x, xh.x); a.x = fminf(fmaxf(z, xslope * z), xhi);
                z = fmaf(a.y, xs.y, xh.y); a.y = fminf(fmaxf(z, xslope * z), xhi);
                z = fmaf(a.z, xs.z, xh.z); a.z = fminf(fmaxf(z, xslope * z), xhi);
                z = fmaf(a.w, xs.w, xh.w); a.w = fminf(fmaxf(z, xslope * z), xhi);
            } else {
                const unsigned bits = (unsigned)(mw >> (chunk * 4)) & 15u;
                a.x = (bits & 1u) ? a.x : a.x * aslope; a.y = (bits & 2u) ? a.y : a.y * aslope;
                a.z = (bits & 4u) ? a.z : a.z * aslope; a.w = (bits & 8u) ? a.w : a.w * aslope;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
    };
    // Epilogue: the two k-parity waves of a row half are summed through LDS; the owner then transposes each group of 4
    // accumulator registers across its lane quad (DPP butterfly, as in the NT kernel) so a lane holds 4 consecutive output
    // columns of one row: 4 float4 addend loads + 4 float4 stores per tile instead of 16 dependent scalar load -> wait ->
    // store chains.  Nothing here may make hipcc drain the VM queue (the next tiles' DMA stages are already in flight):
    //   * the LDS exchange uses raw ds_write_b128 / ds_read_b128 + explicit lgkmcnt + raw s_barrier (a compiler-visible LDS
    //     access, or __syncthreads, waits vmcnt(0));
    //   * the tile's addend rows are fetched by raw global_load_dwordx4 when the tile's FIRST k-step is consumed, i.e. before
    //     the following stages' DMA batches — VM loads retire in order, so the counted waits of the next steps cover them;
    //   * the bias vector is loaded once, before the pipeline starts.
    const int quad = lrow >> 2, jq = lane & 3;
    const int colq = quad * 4;
    const bool cok = colq < p.Kc;                                  // Kc % 4 == 0: all four columns or none
    const int colc = cok ? colq : 0;
    const float4 bv = ld4(p.bias + colc);
    float* xr_w = xred + ((wm * 4) * 64 + lane) * 4;               // [wm][q][lane][4]: lane-contiguous 16-B slots
    v4f_t ad0, ad1, ad2, ad3;
    auto addend_fetch = [&](int mt) {                              // raw loads: see above
        const int64_t m0 = (int64_t)mt * BMS + wm * 32 + 4 * khalf + jq;
        const float* a0 = p.addend + min(m0, p.M - 1) * p.Kc + colc;
        const float* a1 = p.addend + min(m0 + 8, p.M - 1) * p.Kc + colc;
        const float* a2 = p.addend + min(m0 + 16, p.M - 1) * p.Kc + colc;
        const float* a3 = p.addend + min(m0 + 24, p.M - 1) * p.Kc + colc;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ad0) : "v"(a0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ad1) : "v"(a1) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ad2) : "v"(a2) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ad3) : "v"(a3) : "memory");
    };
    auto epilogue = [&](int mt) {
        if (wk == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f_t v = {acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
                asm volatile("ds_write_b128 %0, %1" : : "v"(lds_off(xr_w + q * 256)), "v"(v) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wk == 0) {
            v4f_t xq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xq[q] = lds_read_f4(xr_w + q * 256);
            MNY_LGKM_WAIT(xq[0]);
            MNY_LGKM_DEP(xq[1]); MNY_LGKM_DEP(xq[2]); MNY_LGKM_DEP(xq[3]);
            if (p.addend) { MNY_LGKM_DEP(ad0); MNY_LGKM_DEP(ad1); MNY_LGKM_DEP(ad2); MNY_LGKM_DEP(ad3); }
            const int64_t m0 = (int64_t)mt * BMS;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float r0 = acc[gq * 4 + 0] + xq[gq].x, r1 = acc[gq * 4 + 1] + xq[gq].y;
                float r2 = acc[gq * 4 + 2] + xq[gq].z, r3 = acc[gq * 4 + 3] + xq[gq].w;
                {   // stage A: exchange with lane^1 inside the quad
                    const bool odd = lane & 1;
                    const float xa = odd ? r0 : r1, xb = odd ? r2 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0xB1, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0xB1, 0xF, 0xF, true));
                    if (odd) { r0 = ya; r2 = yb; } else { r1 = ya; r3 = yb; }
                }
                {   // stage B: exchange with lane^2
                    const bool hi2 = lane & 2;
                    const float xa = hi2 ? r0 : r2, xb = hi2 ? r1 : r3;
                    const float ya = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xa), 0x4E, 0xF, 0xF, true));
                    const float yb = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, xb), 0x4E, 0xF, 0xF, true));
                    if (hi2) { r0 = ya; r1 = yb; } else { r2 = ya; r3 = yb; }
                }
                const v4f_t ad = gq == 0 ? ad0 : (gq == 1 ? ad1 : (gq == 2 ? ad2 : ad3));
                float4 o = make_float4(r0 + bv.x, r1 + bv.y, r2 + bv.z, r3 + bv.w);
                if (p.addend) { o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w; }
                const int64_t row = m0 + wm * 32 + 8 * gq + 4 * khalf + jq;
                if (cok && row < p.M) st4_stream(p.C + row * p.Kc + colq, o);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // the exchange buffer may be rewritten by the next tile
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };
    __syncthreads();
    int i_mt = mt_begin, i_kt = 0, i_slot = 0, c_mt = mt_begin, c_kt = 0, c_slot = 0;
    auto issue_next = [&]() { issue(i_mt, i_kt, i_slot); if (++i_kt == nk) { i_kt = 0; ++i_mt; } if (++i_slot == S) i_slot = 0; };
    // Software pipeline.  The mask word of the NEXT step is fetched with a raw `global_load_dwordx2` (inline asm) placed before
    // that step's DMA batch; the counted wait at the END of the iteration (all but the newest LPW VM operations retired — VM
    // loads retire in issue order) covers it, and only then is it moved into the register the next iteration consumes.  Load,
    // wait and move are volatile asm statements, so their order is fixed and the compiler — which would otherwise put a
    // vmcnt(0) in front of any use of an ordinary load's result and drain the DMA ring every step — never sees a pending load.
    unsigned long long m_next = 0ull, m_cur = 0ull;
    auto mask_fetch = [&](int mt, int kt) {
        const int64_t row = min((int64_t)mt * BMS + wm * 32 + lrow, p.M - 1);
        const int kc = min(kt, nkg - 1);
        const unsigned long long* ptr = p.mask + (int64_t)kc * p.npairs + (row >> 1);
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(m_next) : "v"(ptr) : "memory");
    };
    auto mask_commit = [&]() { asm volatile("v_mov_b64 %0, %1" : "=v"(m_cur) : "v"(m_next)); };
    auto mask_word = [&](int mt) -> unsigned long long {                    // the row's 32 bits of the committed pair word
        const int64_t row = min((int64_t)mt * BMS + wm * 32 + lrow, p.M - 1);
        return m_cur >> ((row & 1) * 32);
    };
    auto consume = [&]() {
        compute(c_mt, c_kt, c_slot, mask_word(c_mt));
        if (++c_kt == nk) { epilogue(c_mt); c_kt = 0; ++c_mt; }
        if (++c_slot == S) c_slot = 0;
    };
    // addend rows of tile `mt` are requested when the loop is about to consume the tile's first step (before that
    // iteration's DMA batch): nk >= 2 steps and their counted waits lie between the request and the epilogue
    auto maybe_fetch_addend = [&]() { if (p.addend && c_kt == 0 && wk == 0) addend_fetch(c_mt); };
    auto fetch_next = [&](bool more) {                                        // mask of the step AFTER the one about to be consumed
        int n_kt = c_kt + 1, n_mt = c_mt;
        if (n_kt == nk) { n_kt = 0; ++n_mt; }
        if (!more) { n_mt = c_mt; n_kt = c_kt; }                              // past the end: re-read a valid word (unused)
        mask_fetch(n_mt, n_kt);
    };
    if (total <= 0) return;
    mask_fetch(c_mt, c_kt);                                                   // oldest entry of the VM queue
    const int pre = total < S - 1 ? total : S - 1;
    for (int t = 0; t < pre; ++t) issue_next();
    const int steady = total - pre;
    if (steady > 0) wait_vmcnt<LPW*(S - 2)>(); else wait_vmcnt<0>();
    mask_commit();
    for (int t = 0; t < steady; ++t) {
        __builtin_amdgcn_s_barrier();                    // every wave's share of the stage to consume has landed
        fetch_next(t + 1 < total);
        maybe_fetch_addend();
        issue_next();
        consume();
        if (t + 1 < steady) wait_vmcnt<LPW*(S - 2)>(); else wait_vmcnt<0>();
        mask_commit();
    }
    for (int t = 0; t < pre; ++t) {
        __builtin_amdgcn_s_barrier();
        fetch_next(steady + t + 1 < total);
        maybe_fetch_addend();
        wait_vmcnt<0>();                                 // drain phase: nothing newer is worth keeping in flight
        consume();
        mask_commit();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// stage 2, second generation: no LDS ring, no barrier in the loop.
// The DMA-ring kernel above streams G at 3.9 TB/s: a 64-row step is 12 KB per workgroup behind a barrier, two of them in flight,
// and the matrix pipe idles on memory latency (replacing its MFMAs by the six-product bf16 form changed nothing).  A wave can
// instead load the A operand of v_mfma_f32_32x32x16_bf16 STRAIGHT from the row-major tensor — lane (row i, half h) owns, per
// 16-column stage, the two 16-B chunks 16s + 4h.. and 16s + 8 + 4h.. of row i — and that access pattern streams at the copy rate
// (tools/probe/rowfrag_probe.hip: 5.3-5.5 TB/s for N = 96 / 144 / 192, as fast as a row-linear float4 stream).  So: one wave =
// one 32-row tile at a time, all of its G fragments (2 * N/16 float4 per lane) in registers, each pair re-requested for the wave's
// NEXT tile the moment it has been cut (rolling prefetch: a full tile of loads always in flight, one buffer); the small operands
// (B1 = ca o W^T [Kc][N], Q [Kc][Kc]) are cut into bf16 planes ONCE per workgroup into LDS in MFMA-operand order; products in the
// six-product form (x6_split), two accumulators (even / odd stages); epilogue = the DPP quad transpose of the first generation.
// N % 16 == 0, N <= 192, Kc <= 32.
// ---------------------------------------------------------------------------------------------------------------------
// T: storage type of G, X, the addend and dX (bf16_t for bf16-storage plans: 8-byte loads widened to fp32, the same cuts and products —
// the bf16-exact G has empty mid / lo pieces, which costs matrix-pipe cycles this HBM-bound kernel has to spare — one RNE on store).
// HALF: N = 16 NS - 8 (the last stage's upper eight columns do not exist: N = 72 of MobileNetV3).
template <int NS, typename T = float, bool HALF = false, bool RED = false>
__global__ __launch_bounds__(256) void pw_bnbwd_dgrad2_kernel(BndArgs p) {
    constexpr int N = NS * 16 - (HALF ? 8 : 0), NW = (N + 31) / 32;               // mask words (32 columns each) per row
    const T* const pG = reinterpret_cast<const T*>(p.G);
    const T* const pX = reinterpret_cast<const T*>(p.X);
    const T* const pAd = reinterpret_cast<const T*>(p.addend);
    T* const pC = reinterpret_cast<T*>(p.C);
    __shared__ __attribute__((aligned(16))) float sB[(NS + 2) * 3 * 256];     // [stage][piece][lane] 16-B slots; stages NS, NS+1 = Q
    __shared__ __attribute__((aligned(16))) float sXs[32];
    __shared__ __attribute__((aligned(16))) float sXh[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, khalf = lane >> 5;
    const int Kc = p.Kc;
    const int XS = (Kc + 15) / 16;                               // stages of the (X, Q) product
    // ---- cut the small operands once per workgroup ----
    for (int c = tid; c < (NS + 2) * 64; c += 256) {
        const int sidx = c >> 6, l = c & 63, j = l & 31, h = l >> 5;
        const bool isq = sidx >= NS;
        const int st = isq ? sidx - NS : sidx;
        const int width = isq ? Kc : N;
        const float* src = (isq ? p.Q : p.B1) + (int64_t)j * width;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = st * 16 + (e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4));
            v[e] = (j < Kc && k < width) ? src[k] : 0.f;
        }
        bf16x8_t hh, mm, ll;
        x6_split(v4f_t{v[0], v[1], v[2], v[3]}, v4f_t{v[4], v[5], v[6], v[7]}, hh, mm, ll);
        float* dst = sB + (sidx * 3) * 256 + l * 4;
        *reinterpret_cast<v4f_t*>(dst) = __builtin_bit_cast(v4f_t, hh);
        *reinterpret_cast<v4f_t*>(dst + 256) = __builtin_bit_cast(v4f_t, mm);
        *reinterpret_cast<v4f_t*>(dst + 512) = __builtin_bit_cast(v4f_t, ll);
    }
    if (tid < 32) {
        sXs[tid] = (p.xsc && tid < Kc) ? p.xsc[tid] : (tid < Kc ? 1.f : 0.f);
        sXh[tid] = (p.xsc && tid < Kc) ? p.xsh[tid] : 0.f;
    }
    __syncthreads();
    const float aslope = act_slope(p.act), xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    const int quad = lrow >> 2, jq = lane & 3;
    const int colq = quad * 4;
    const bool cok = colq < Kc;
    const int colc = cok ? colq : 0;
    const float4 bv = ld4(p.bias + colc);
    float4 rsc = f4one(), rsh = f4zero(), rmu = f4zero(), ris = f4zero(), rs1 = f4zero(), rs2 = f4zero();
    if constexpr (RED) { rsc = ld4(p.r_scale + colc); rsh = ld4(p.r_shift + colc); rmu = ld4(p.r_mean + colc); ris = ld4(p.r_invstd + colc); }
    const float rslope = act_slope(p.r_act), rhi = act_hi(p.r_act);
    const unsigned* mask32 = reinterpret_cast<const unsigned*>(p.mask);
    const int64_t mstride = 2 * p.npairs;                        // 32-bit mask words per 32-column tile

    const int64_t ntiles = (p.M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    float4 g[2 * NS], xr[4], ad[4];
    unsigned mk[NW];
    auto row_of = [&](int64_t tile) { const int64_t r = tile * 32 + lrow; return r < p.M ? r : p.M - 1; };
    auto load_g = [&](int64_t tile, int s) {
        const T* row = pG + row_of(tile) * N + 16 * s + 4 * khalf;
        g[2 * s] = ld4(row);
        if (HALF && s == NS - 1) g[2 * s + 1] = f4zero(); else g[2 * s + 1] = ld4(row + 8);
    };
    auto load_mx = [&](int64_t tile) {                           // mask words and the thin X row of the lane
        const int64_t r = row_of(tile);
#pragma unroll
        for (int w = 0; w < NW; ++w) mk[w] = mask32[(int64_t)w * mstride + r];
        const T* xrow = pX + r * Kc + 4 * khalf;
        xr[0] = ld4(xrow); xr[1] = (8 + 4 * khalf < Kc) ? ld4(xrow + 8) : f4zero();
        xr[2] = (16 + 4 * khalf < Kc) ? ld4(xrow + 16) : f4zero(); xr[3] = (24 + 4 * khalf < Kc) ? ld4(xrow + 24) : f4zero();
    };
    auto load_ad = [&](int64_t tile) {
        if (p.addend) {
            const int64_t m0 = tile * 32 + 4 * khalf + jq;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) { const int64_t r2 = m0 + 8 * gq; ad[gq] = ld4(pAd + (r2 < p.M ? r2 : p.M - 1) * Kc + colc); }
        }
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntiles) {
#pragma unroll
        for (int s = 0; s < NS; ++s) load_g(tile, s);
        load_mx(tile);
        load_ad(tile);
    }
    for (; tile < ntiles; tile += stride) {
        const int64_t next = tile + stride < ntiles ? tile + stride : tile;      // past the end: harmless re-reads
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a0 = g[2 * s], a1 = g[2 * s + 1];
            const unsigned word = mk[s >> 1] >> (16 * (s & 1) + 4 * khalf);
            float z[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[e] = ((word >> e) & 1u) ? z[e] : z[e] * aslope;
                z[4 + e] = ((word >> (8 + e)) & 1u) ? z[4 + e] : z[4 + e] * aslope;
            }
            bf16x8_t ah, am, al;
            x6_split(v4f_t{z[0], z[1], z[2], z[3]}, v4f_t{z[4], z[5], z[6], z[7]}, ah, am, al);
            // this stage's registers are free: request them for the next tile — HERE, not earlier: without the fence the compiler hoists every
            // load of the next tile to the top of the loop (SSA renames the registers) and the kernel needs two tiles of them (435 VGPRs)
            asm volatile("" : "+v"(ah), "+v"(am), "+v"(al) :: "memory");
            load_g(next, s);
            const float* bsrc = sB + (s * 3) * 256 + lane * 4;
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc));
            const bf16x8_t bm = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 256));
            const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 512));
            f32x16& acc = (s & 1) ? acc1 : acc0;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                   // stages in order: the cuts of later stages would otherwise all be formed first
        }
        // the (act(X), Q) product
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st < XS) {
                const float4 a0 = xr[2 * st], a1 = xr[2 * st + 1];
                float z[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float4 s0 = *reinterpret_cast<const float4*>(sXs + st * 16 + 4 * khalf), s1 = *reinterpret_cast<const float4*>(sXs + st * 16 + 8 + 4 * khalf);
                const float4 h0 = *reinterpret_cast<const float4*>(sXh + st * 16 + 4 * khalf), h1 = *reinterpret_cast<const float4*>(sXh + st * 16 + 8 + 4 * khalf);
                const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float t = fmaf(z[e], sv[e], hv[e]); z[e] = fminf(fmaxf(t, xslope * t), xhi); }
                bf16x8_t ah, am, al;
                x6_split(v4f_t{z[0], z[1], z[2], z[3]}, v4f_t{z[4], z[5], z[6], z[7]}, ah, am, al);
                const float* bsrc = sB + ((NS + st) * 3) * 256 + lane * 4;
                const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc));
                const bf16x8_t bm = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 256));
                const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const v4f_t*>(bsrc + 512));
                f32x16& acc = st ? acc1 : acc0;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
        asm volatile("" : "+v"(acc0), "+v"(acc1) :: "memory");     // (the next tile's small loads stay behind the last use of this tile's)
        load_mx(next);
        // epilogue: lane = output column lrow, registers = rows; DPP quad transpose -> lane jq of a quad holds row 8 gq + 4 khalf + jq, 4 columns
        const int64_t m0 = tile * 32;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float r0 = acc0[gq * 4 + 0] + acc1[gq * 4 + 0], r1 = acc0[gq * 4 + 1] + acc1[gq * 4 + 1];
            float r2 = acc0[gq * 4 + 2] + acc1[gq * 4 + 2], r3 = acc0[gq * 4 + 3] + acc1[gq * 4 + 3];
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
            float4 o = make_float4(r0 + bv.x, r1 + bv.y, r2 + bv.z, r3 + bv.w);
            if (p.addend) {
                // scalar adds, spelled out: left to the compiler the bf16 instantiations get v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]
                // (the addend's widened halves sit swapped in their register pair) for row groups 2 and 3, and on MI355X exactly those
                // sums came out WITHOUT the addend in lanes 48-63 of a few waves per launch, differently from launch to launch
                // (tools/ab/stress_bnbwd_bf16.py; the registers themselves were intact — DESIGN.md, round 3)
                asm("v_add_f32 %0, %0, %1" : "+v"(o.x) : "v"(ad[gq].x)); asm("v_add_f32 %0, %0, %1" : "+v"(o.y) : "v"(ad[gq].y));
                asm("v_add_f32 %0, %0, %1" : "+v"(o.z) : "v"(ad[gq].z)); asm("v_add_f32 %0, %0, %1" : "+v"(o.w) : "v"(ad[gq].w));
            }
            const int64_t row = m0 + 8 * gq + 4 * khalf + jq;
            if (cok && row < p.M) st4_stream(pC + row * Kc + colq, o);
            if constexpr (RED) {                                     // sums over the STORED gradient of the unit in front (its raw output row: a thin 16-B load)
                if (cok && row < p.M) {
                    const float4 yv = ld4(reinterpret_cast<const float*>(p.rY) + row * Kc + colq);
                    const float ov[4] = {o.x, o.y, o.z, o.w}, yy[4] = {yv.x, yv.y, yv.z, yv.w};
                    const float sc4[4] = {rsc.x, rsc.y, rsc.z, rsc.w}, sh4[4] = {rsh.x, rsh.y, rsh.z, rsh.w};
                    const float mu4[4] = {rmu.x, rmu.y, rmu.z, rmu.w}, is4[4] = {ris.x, ris.y, ris.z, ris.w};
                    float a1[4] = {rs1.x, rs1.y, rs1.z, rs1.w}, a2[4] = {rs2.x, rs2.y, rs2.z, rs2.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = fmaf(yy[e], sc4[e], sh4[e]);
                        const float dz = ov[e] * ((z > 0.f ? 1.f : rslope) * (z < rhi ? 1.f : 0.f));
                        a1[e] += dz;
                        a2[e] = fmaf(dz, (yy[e] - mu4[e]) * is4[e], a2[e]);
                    }
                    rs1 = make_float4(a1[0], a1[1], a1[2], a1[3]); rs2 = make_float4(a2[0], a2[1], a2[2], a2[3]);
                }
            }
        }
        asm volatile("" ::: "memory");
        load_ad(next);
    }
    if constexpr (RED) {                                             // one partial row per workgroup: the eight lanes of a column quad, then the four waves, fixed order
        __syncthreads();
        float4* sr = reinterpret_cast<float4*>(sB);                  // [2][256]
        sr[tid] = rs1; sr[256 + tid] = rs2;
        __syncthreads();
        if (tid < 16 && tid * 4 < Kc) {                              // thread = column quad
            float4 a = f4zero(), b = f4zero();
            for (int w = 0; w < 4; ++w)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 4; ++j) {
                        const int l = w * 64 + h * 32 + tid * 4 + j;
                        add4(a, sr[l]); add4(b, sr[256 + l]);
                    }
            float* dst = p.red + (int64_t)blockIdx.x * 2 * Kc;
            st4(dst + tid * 4, a);
            st4(dst + Kc + tid * 4, b);
        }
    }
}

struct BnwPlan { int TI, splits; int64_t rows_per_block; size_t lds1, lds2; int gx2, tiles_per_block, m_tiles; int TIs, nsl; };

static bool bnw_supported(int64_t M, int K, int N) {
    return M >= 4096 && K <= 32 && K % 4 == 0 && N % 4 == 0 && N <= 192 && N > K;
}

static BnwPlan bnw_plan(int64_t M, int K, int N) {
    BnwPlan pl;
    pl.TI = (int)cdiv(N, 32);
    // whole waves of resident workgroups: TI <= 3 -> 168 VGPRs, 3 per CU (768); TI 4,5 -> 2 per CU (512); TI 6 -> 1 per CU
    // stage 1 of a wide unit (5 or 6 column tiles: 225+ VGPRs, 1-2 workgroups per CU) runs as two column slices of <= 3 tiles
    // (168 VGPRs, 3 per CU): each slice streams its half of G and Y and all of the thin X
    static const bool slicing = getenv("MNY_BNW_NOSLICE") == nullptr;
    pl.nsl = (slicing && pl.TI >= 5) ? 2 : 1;
    pl.TIs = (int)cdiv(pl.TI, pl.nsl);
    int64_t splits = (pl.TIs <= 3 ? 768 : 1024) / pl.nsl;
    const int64_t max_splits = cdiv(M, 64);
    if (splits > max_splits) splits = max_splits;
    pl.rows_per_block = cdiv(cdiv(M, splits), 16) * 16;
    pl.splits = (int)cdiv(M, pl.rows_per_block);
    pl.lds1 = (size_t)3 * 16 * (2 * 32 * pl.TIs + 32) * sizeof(float);
    const size_t need = (size_t)4 * (2 * pl.TIs + 1) * 32 * sizeof(float);
    if (pl.lds1 < need) pl.lds1 = need;
    if (pl.lds1 < 3 * 16 * 64 * sizeof(float)) pl.lds1 = 3 * 16 * 64 * sizeof(float);
    pl.lds2 = (size_t)3 * (64 * 32 + 32 * 32) * sizeof(float) + 64 * sizeof(float) + 2 * 16 * 64 * sizeof(float);
    pl.m_tiles = (int)cdiv(M, 64);
    int gx = pl.m_tiles < 768 ? pl.m_tiles : 768;
    pl.tiles_per_block = (int)cdiv(pl.m_tiles, gx);
    pl.gx2 = (int)cdiv(pl.m_tiles, pl.tiles_per_block);
    return pl;
}

// combine + finalize of the fused expand-unit backward for a producer outside this file (exdw.hip): partial rows in the layout above
int pw_bnbwd_finalize_launch(const float* partials, int nparts, float* red, const float* w, const float* gamma, const float* mean,
                             const float* invstd, int64_t M, int Nc, int K, float* dw, float* dgamma, float* dbeta, float* B1, float* Q,
                             float* bias, hipStream_t st) {
    const int64_t stride = bnw_stride(Nc, K);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(stride, 32)), dim3(256), 0, st, partials, nparts, stride, red);
    hipLaunchKernelGGL(pw_bnbwd_finalize_kernel, dim3((unsigned)cdiv((int64_t)Nc * K, 256)), dim3(256), bnw_finalize_lds(Nc, K), st, red, w, gamma,
                       mean, invstd, (double)M, Nc, K, dw, dgamma, dbeta, B1, Q, bias);
    return check_launch("pw_bnbwd_finalize_kernel");
}

}  // namespace mny

using namespace mny;
static thread_local int g_pw_route = -1;        // kernel family of this thread's last pointwise-conv call (mny_pw_last_route below)

extern "C" int mny_pw_stat_parts(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return MNY_EINVAL;
    if (pw_thin_ok(0, 0, M, K, Nc)) return pw_thin_parts(M, K, Nc, 0);
    if (pw_wide_ok(M, K, Nc, false)) return pw_wide_parts(M, K, Nc, false) + ((M & 31) ? nt2_plan(M & 31, K, Nc, true).gx : 0);      // + the last M % 32 rows
    if ((K & 3) == 0 && getenv("MNY_GEMM_V1") == nullptr) return nt2_plan(M, K, Nc, true).gx;
    return nt_plan(M, K, Nc).gx;
}
extern "C" int mny_pw_stat_parts_bf16(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return MNY_EINVAL;
    if (pwt_ok(M, K, Nc, MNY_ACT_NONE, false)) return pwt_parts(M);
    if (pw_thin_ok(1, 0, M, K, Nc)) return pw_thin_parts(M, K, Nc, 0);
    if ((K & 7) == 0 && getenv("MNY_GEMM_V1") == nullptr) return nt2_plan(M, K, Nc, true, 1).gx;
    return nt_plan(M, K, Nc).gx;
}

// register-staged NT kernel (any K, any storage type)
template <typename T>
static int pw_fwd_v1(const T* x, const float* in_scale, const float* in_shift, int in_act, const T* w, const float* bias,
                     const T* addend, T* y, float* stats, int64_t M, int K, int Nc, hipStream_t st) {
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    NtPlan pl = nt_plan(M, K, Nc, xf);
    MNY_REQUIRE(pl.lds <= 160 * 1024, "pw_fwd: K=%d too large for the LDS scale cache", K);
    GemmArgs a{x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, pl.m_tiles, pl.tiles_per_block};
    dim3 grid(pl.gx, pl.n_tiles), block(256);
    {                             // > 64 KB of dynamic LDS needs an explicit opt-in per kernel
        const void* ks[] = {(const void*)pw_gemm_nt_kernel<T, 1, 16>, (const void*)pw_gemm_nt_kernel<T, 2, 16>, (const void*)pw_gemm_nt_kernel<T, 3, 16>,
                            (const void*)pw_gemm_nt_kernel<T, 4, 16>, (const void*)pw_gemm_nt_kernel<T, 1, 32>, (const void*)pw_gemm_nt_kernel<T, 2, 32>,
                            (const void*)pw_gemm_nt_kernel<T, 3, 32>, (const void*)pw_gemm_nt_kernel<T, 4, 32>};
        for (const void* k : ks)
            if (!allow_lds(k, 160 * 1024)) {
                set_error("pw_fwd: hipFuncSetAttribute failed"); return MNY_EHIP;
            }
    }
#define MNY_NT(TN_, B) hipLaunchKernelGGL((pw_gemm_nt_kernel<T, TN_, B>), grid, block, pl.lds, st, a)
    switch (pl.TN * 100 + pl.BK) {
        case 116: MNY_NT(1, 16); break; case 216: MNY_NT(2, 16); break; case 316: MNY_NT(3, 16); break; case 416: MNY_NT(4, 16); break;
        case 132: MNY_NT(1, 32); break; case 232: MNY_NT(2, 32); break; case 332: MNY_NT(3, 32); break; default: MNY_NT(4, 32); break;
    }
#undef MNY_NT
    return check_launch("pw_gemm_nt_kernel");
}

extern "C" int mny_pw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                          const float* bias, const float* addend, float* y, float* stats, int64_t M, int K, int Nc,
                          void* stream) {
    MNY_REQUIRE(x && w && y, "pw_fwd: null pointer");
    MNY_REQUIRE(M > 0 && K > 0 && Nc > 0, "pw_fwd: empty problem");
    MNY_REQUIRE(!(stats && bias), "pw_fwd: stats and bias are mutually exclusive");
    hipStream_t st = (hipStream_t)stream;
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    static const bool force_v1 = getenv("MNY_GEMM_V1") != nullptr;   // A/B switch for profiling
    if (pw_thin_ok(0, 0, M, K, Nc)) {                      // short reduction: vector-ALU stream kernel (pwthin.hip)
        g_pw_route = MNY_ROUTE_THIN;
        return pw_thin_launch(0, x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, st);
    }
    bool took_wide = false;
    if (pw_wide_ok(M, K, Nc, false)) {              // short reduction, wide output: barrier-free matrix-core kernel (pwwide.hip)
        if (!bias && !addend && in_act != MNY_ACT_HSIGMOID) {
            const int64_t Mf = M & ~(int64_t)31;    // it takes whole 32-row tiles; the last M % 32 rows follow below (one more partial row)
            g_pw_route = MNY_ROUTE_WIDE;
            took_wide = true;
            const int rc = pw_wide_launch(x, in_scale, in_shift, in_act, w, y, stats, Mf, K, Nc, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, st);
            if (rc != 0 || Mf == M) return rc;
            if (stats) stats += (int64_t)pw_wide_parts(M, K, Nc, false) * 2 * Nc;
            x += Mf * K; y += Mf * Nc; M -= Mf;
        } else {
            MNY_REQUIRE(!stats, "pw_fwd: statistics with a bias / addend / h-sigmoid input on a wide-kernel shape");   // (its partial rows follow mny_pw_stat_parts)
        }
    }
    if ((K & 3) == 0 && !force_v1) {                // LDS-DMA pipeline (v2): 16-B aligned input rows
        Nt2Plan p2 = nt2_plan(M, K, Nc, xf);
        MNY_REQUIRE(p2.lds <= 64 * 1024, "pw_fwd: K=%d too large for the LDS scale cache", K);
        Gemm2Args g{x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                    nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        dim3 grid2(p2.grid), block2(256);
        const int XF = !xf ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1);
        if (!took_wide) g_pw_route = nt_x6(M, K, Nc) ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32;     // (the last M % 32 rows of a wide-kernel call stay "wide")
        hipLaunchKernelGGL(nt2_kernel(p2.TN, XF, 0, nt_x6(M, K, Nc)), grid2, block2, p2.lds, st, g);
        return check_launch("pw_gemm_nt_dma_kernel");
    }
    if (!took_wide) g_pw_route = MNY_ROUTE_TILE_V1;
    return pw_fwd_v1<float>(x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, st);
}

// data-gradient GEMM of a pointwise conv whose INPUT is the output of a conv+BN+act unit: dx = dy * W (W^T given, [K][Nc] rows = [Nc][K]
// transposed -> NT), and in the same epilogue that unit's BN-backward reduction over (dx, its raw output y): the separate
// mny_bn_bwd_reduce pass (one more read of dx and of y) disappears — only y is read, by the tile that just produced dx.
// the reduction epilogue holds four more per-column constants and a group of loads: the 160-column tile (TN = 5) spills with it
static const int kRedMaxTn = getenv("MNY_RED_TN") ? atoi(getenv("MNY_RED_TN")) : 4;      // (A/B: 3 = no 128-column reduction tiles)
static bool dgrad_bnred_ok(int64_t M, int K, int Nc, int act) {
    static const bool red512 = getenv("MNY_RED512") != nullptr && atoi(getenv("MNY_RED512")) != 0;
    if (K >= 512 && Nc >= 512 && !red512) return false;      // matrix-pipe-bound: the reduce pass it would save is cheaper than the longer epilogue (MNY_RED512=1: A/B)
    if ((pw_thin_ok(0, 1, M, K, Nc) || pw_thin_ok(1, 1, M, K, Nc)) && act == MNY_ACT_HSIGMOID) return false;
    if (pw_wide_ok(M, K, Nc, true) && act >= MNY_ACT_HSWISH) return false;      // the wide-output kernel's reduction form knows the clamp family (its h-swish builds spill)      // the short-reduction kernel knows the clamp family and h-swish (what units use)
    return M > 0 && K > 0 && Nc > 0 && (K & 3) == 0 && act >= MNY_ACT_NONE && act <= MNY_ACT_HSIGMOID && getenv("MNY_GEMM_V1") == nullptr;
}
extern "C" int mny_pw_dgrad_bnred_supported(int64_t M, int K, int Nc, int act) { return dgrad_bnred_ok(M, K, Nc, act) ? 1 : 0; }
extern "C" int mny_pw_dgrad_bnred_parts(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0 || (K & 3)) return MNY_EINVAL;
    if (pw_thin_ok(0, 1, M, K, Nc)) return pw_thin_parts(M, K, Nc, 1);
    if (pw_wide_ok(M, K, Nc, true)) return pw_wide_parts(M, K, Nc, true) + ((M & 31) ? nt2_plan(M & 31, K, Nc, false, 0, kRedMaxTn).gx : 0);
    return nt2_plan(M, K, Nc, false, 0, kRedMaxTn).gx;
}
template <int BF>
static int pw_dgrad_bnred_impl(const void* dy, const void* wT, void* dx, const void* y, const float* scale, const float* shift, int act,
                               const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream, const void* addend = nullptr) {
    MNY_REQUIRE(dy && wT && dx && y && scale && shift && mean && invstd && red, "pw_dgrad_bnred: null pointer");
    MNY_REQUIRE(dgrad_bnred_ok(M, K, Nc, act) && (!BF || (K & 7) == 0), "pw_dgrad_bnred: unsupported problem M=%lld K=%d N=%d act=%d (see mny_pw_dgrad_bnred_supported)",
                (long long)M, K, Nc, act);
    if (BF && pwt_ok(M, K, Nc, MNY_ACT_NONE, false)) {       // K <= 48 at a large pixel count: a wave per 16 pixels (gate.hip)
        g_pw_route = MNY_ROUTE_WAVE16;
        return pwt_launch(dy, nullptr, nullptr, MNY_ACT_NONE, wT, addend, dx, red, M, K, Nc, (hipStream_t)stream, y, scale, shift, mean, invstd, act);
    }
    if (pw_thin_ok(BF, 1, M, K, Nc)) g_pw_route = MNY_ROUTE_THIN;
    if (pw_thin_ok(BF, 1, M, K, Nc))
        return pw_thin_launch(BF, dy, nullptr, nullptr, MNY_ACT_NONE, wT, nullptr, addend, dx, red, M, K, Nc, addend ? 2 : 1, y, scale, shift, mean, invstd, act,
                              (hipStream_t)stream);
    bool took_wide = false;
    if (!BF && pw_wide_ok(M, K, Nc, true)) {        // whole 32-row tiles on the barrier-free kernel, the last M % 32 rows below (one more partial row)
        g_pw_route = MNY_ROUTE_WIDE;
        took_wide = true;
        const int64_t Mf = M & ~(int64_t)31;
        const int rc = pw_wide_launch((const float*)dy, nullptr, nullptr, MNY_ACT_NONE, (const float*)wT, (float*)dx, red, Mf, K, Nc, (const float*)y, scale, shift,
                                      mean, invstd, act, (const float*)addend, (hipStream_t)stream);
        if (rc != 0 || Mf == M) return rc;
        red += (int64_t)pw_wide_parts(M, K, Nc, true) * 2 * Nc;
        dy = (const float*)dy + Mf * K; dx = (float*)dx + Mf * Nc; y = (const float*)y + Mf * Nc;
        if (addend) addend = (const float*)addend + Mf * Nc;
        M -= Mf;
    }
    Nt2Plan p2 = nt2_plan(M, K, Nc, false, BF, kRedMaxTn);
    MNY_REQUIRE(p2.lds <= 64 * 1024, "pw_dgrad_bnred: K=%d too large", K);
    Gemm2Args g{dy, nullptr, nullptr, MNY_ACT_NONE, wT, nullptr, addend, dx, red, M, K, Nc, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                y, scale, shift, mean, invstd, act};
    Nt2Kernel k;
    constexpr int X = BF ? 0 : 1;                   // the six-product bf16 form exists for fp32 operands only
    const bool x6 = !BF && nt_x6(M, K, Nc);
    if (!took_wide) g_pw_route = x6 ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32;
    if (addend) switch (p2.TN) {                    // RED = 2: with an addend (own instantiations: the addend loads cost the TN = 4 variant registers)
        case 1: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, BF, 2, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, BF, 2>; break;
        case 2: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, BF, 2, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, BF, 2>; break;
        case 3: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, BF, 2, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, BF, 2>; break;
        default: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<4, 0, BF, 2, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<4, 0, BF, 2>; break;
    }
    else switch (p2.TN) {
        case 1: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, BF, 1, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, BF, 1>; break;
        case 2: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, BF, 1, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, BF, 1>; break;
        case 3: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, BF, 1, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, BF, 1>; break;
        default: k = x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<4, 0, BF, 1, X> : (Nt2Kernel)pw_gemm_nt_dma_kernel<4, 0, BF, 1>; break;
    }
    hipLaunchKernelGGL(k, dim3(p2.grid), dim3(256), p2.lds, (hipStream_t)stream, g);
    return check_launch("pw_gemm_nt_dma_kernel<RED>");
}
extern "C" int mny_pw_dgrad_bnred(const float* dy, const float* wT, float* dx, const float* y, const float* scale, const float* shift, int act,
                                  const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream) {
    return pw_dgrad_bnred_impl<0>(dy, wT, dx, y, scale, shift, act, mean, invstd, red, M, K, Nc, stream);
}
// the addend variant exists for column tiles of <= 96 (TN <= 3: the 128-column tile has no registers left for the addend loads)
extern "C" int mny_pw_dgrad_bnred_add_supported(int64_t M, int K, int Nc, int act) {
    return dgrad_bnred_ok(M, K, Nc, act) && (pw_thin_ok(0, 1, M, K, Nc) || pw_wide_ok(M, K, Nc, true) || nt2_plan(M, K, Nc, false, 0, kRedMaxTn).TN <= 3) ? 1 : 0;
}
extern "C" int mny_pw_dgrad_bnred_add_supported_bf16(int64_t M, int K, int Nc, int act) {
    return dgrad_bnred_ok(M, K, Nc, act) && (K & 7) == 0 && (pwt_ok(M, K, Nc, MNY_ACT_NONE, false) || pw_thin_ok(1, 1, M, K, Nc) || nt2_plan(M, K, Nc, false, 1, kRedMaxTn).TN <= 3) ? 1 : 0;
}
extern "C" int mny_pw_dgrad_bnred_add(const float* dy, const float* wT, const float* addend, float* dx, const float* y, const float* scale, const float* shift,
                                      int act, const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(addend && mny_pw_dgrad_bnred_add_supported(M, K, Nc, act), "pw_dgrad_bnred_add: null addend or unsupported shape (mny_pw_dgrad_bnred_add_supported)");
    return pw_dgrad_bnred_impl<0>(dy, wT, dx, y, scale, shift, act, mean, invstd, red, M, K, Nc, stream, addend);
}
extern "C" int mny_pw_dgrad_bnred_add_bf16(const void* dy, const void* wT, const void* addend, void* dx, const void* y, const float* scale, const float* shift,
                                           int act, const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(addend && mny_pw_dgrad_bnred_add_supported_bf16(M, K, Nc, act), "pw_dgrad_bnred_add: null addend or unsupported shape");
    return pw_dgrad_bnred_impl<1>(dy, wT, dx, y, scale, shift, act, mean, invstd, red, M, K, Nc, stream, addend);
}
// bf16 storage: dy, wT, dx, y are bf16; the sums are taken over the ROUNDED dx (what a separate reduce pass would read back)
extern "C" int mny_pw_dgrad_bnred_supported_bf16(int64_t M, int K, int Nc, int act) { return dgrad_bnred_ok(M, K, Nc, act) && (K & 7) == 0 ? 1 : 0; }
extern "C" int mny_pw_dgrad_bnred_parts_bf16(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0 || (K & 7)) return MNY_EINVAL;
    if (pwt_ok(M, K, Nc, MNY_ACT_NONE, false)) return pwt_parts(M);
    if (pw_thin_ok(1, 1, M, K, Nc)) return pw_thin_parts(M, K, Nc, 1);
    return nt2_plan(M, K, Nc, false, 1, kRedMaxTn).gx;
}
extern "C" int mny_pw_dgrad_bnred_bf16(const void* dy, const void* wT, void* dx, const void* y, const float* scale, const float* shift, int act,
                                       const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream) {
    return pw_dgrad_bnred_impl<1>(dy, wT, dx, y, scale, shift, act, mean, invstd, red, M, K, Nc, stream);
}

// ---- low-rank BatchNorm-backward correction of a wide expand unit's data gradient (csrc/lrbwd.hip) ----------------------------------
// dx = view(x) Q + r + addend  (x the K-wide input of the unit behind its LINEAR view in_scale / in_shift — applied where the A fragment is read,
// as in every forward GEMM —, Q [K][K] symmetric and r [K] from mny_lr_prep, addend = the main term dzc W, usually dx itself); with `red` the
// BN-backward sums of the unit whose complete output gradient dx now is (its raw output ry, as mny_pw_dgrad_bnred_add).  Always the LDS-DMA
// kernel, column tiles of <= 96 (the addend form's limit).
static bool lr_fix_wide(int64_t M, int K, int r_act) {       // which kernel a call WITH a reduction target takes — the one rule for the launch and the row count
    static const bool no_wide_fix = getenv("MNY_LR_FIX_DMA") != nullptr;          // (A/B: always the LDS-DMA kernel)
    return !no_wide_fix && pw_wide_fix_ok(M, K) && r_act < MNY_ACT_HSWISH;
}
static int pw_lr_fix_impl(const float* x, const float* in_scale, const float* in_shift, const float* q, const float* r, const float* addend, float* dx,
                          const float* ry, const float* r_scale, const float* r_shift, int r_act, const float* r_mean, const float* r_invstd, float* red,
                          int64_t M, int K, void* stream) {
    MNY_REQUIRE(x && q && r && addend && dx && M > 0 && K > 0 && (K & 3) == 0 && (!in_scale) == (!in_shift), "pw_lr_fix: bad arguments (K=%d)", K);
    MNY_REQUIRE(!red || (ry && r_scale && r_shift && r_mean && r_invstd && r_act >= MNY_ACT_NONE && r_act <= MNY_ACT_HSIGMOID), "pw_lr_fix: incomplete reduction target");
    const bool xf = in_scale != nullptr;
    if (red && lr_fix_wide(M, K, r_act)) {            // K = 64 / 96 at whole 32-row tiles: the barrier-free kernel of pwwide.hip
        g_pw_route = MNY_ROUTE_WIDE;
        return pw_wide_fix_launch(x, in_scale, in_shift, q, r, addend, dx, red, M, K, ry, r_scale, r_shift, r_mean, r_invstd, r_act, (hipStream_t)stream);
    }
    Nt2Plan p2 = nt2_plan(M, K, K, xf, 0, 3);
    MNY_REQUIRE(p2.lds <= 64 * 1024, "pw_lr_fix: K=%d too large", K);
    Gemm2Args g{x, in_scale, in_shift, MNY_ACT_NONE, q, r, addend, dx, red, M, K, K, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                ry, r_scale, r_shift, r_mean, r_invstd, r_act};
    const bool x6 = nt_x6(M, K, K);
    Nt2Kernel k;
#define MNY_FX(T) (xf ? (x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 0, 2, 1, 1> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 1, 0, 2, 0, 1>) \
                      : (x6 ? (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 0, 2, 1, 1> : (Nt2Kernel)pw_gemm_nt_dma_kernel<T, 0, 0, 2, 0, 1>))
    if (red) switch (p2.TN) { case 1: k = MNY_FX(1); break; case 2: k = MNY_FX(2); break; default: k = MNY_FX(3); break; }
    else k = nt2_kernel(p2.TN, xf ? 1 : 0, 0, x6 ? 1 : 0);
#undef MNY_FX
    g_pw_route = x6 ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32;
    hipLaunchKernelGGL(k, dim3(p2.grid), dim3(256), p2.lds, (hipStream_t)stream, g);
    return check_launch("pw_gemm_nt_dma_kernel<lr_fix>");
}
extern "C" int mny_pw_lr_fix_parts(int64_t M, int K, int r_act) {
    if (M <= 0 || K <= 0 || (K & 3)) return MNY_EINVAL;
    return lr_fix_wide(M, K, r_act) ? pw_wide_fix_parts(M, K) : nt2_plan(M, K, K, true, 0, 3).gx;
}
extern "C" int mny_pw_lr_fix(const float* x, const float* in_scale, const float* in_shift, const float* q, const float* r, const float* addend, float* dx,
                             const float* ry, const float* r_scale, const float* r_shift, int r_act, const float* r_mean, const float* r_invstd, float* red,
                             int64_t M, int K, void* stream) {
    return pw_lr_fix_impl(x, in_scale, in_shift, q, r, addend, dx, ry, r_scale, r_shift, r_act, r_mean, r_invstd, red, M, K, stream);
}

// ---- forward / data-gradient GEMMs on pre-cut weight planes (fp32 plans, six-product form) ---------------------------------------
// Planes pay where the matrix pipe is the bound.  Round 2: a stage of B grew from 64 to 96 bytes per row, which cost the mid-size shapes
// their third resident workgroup (7-18 % faster from ~60 FLOP per byte up, 10-30 % slower below ~40).  Round 3: the plane ring is two
// slots deep (the A ring stays at three), the LDS footprint equals the plain mode's and the threshold moved down to 30 FLOP per byte.
static bool w6_ok(int64_t M, int K, int Nc) {
    static const bool off = getenv("MNY_NO_W6") != nullptr;
    // (round 3: with the two-slot plane ring the planes no longer cost a resident workgroup; same-box A/B per shape: K576 N96 -11 %,
    // K384 N96 -10 %, the 75-channel heads -9..-11 %, K384 N64 (27 FLOP per byte) +0..+7 % -> threshold 30, was 50)
    static const double min_ai = getenv("MNY_W6_AI") ? atof(getenv("MNY_W6_AI")) : 30.0;
    const double ai = 2.0 * K * Nc / (4.0 * (K + Nc));
    return !off && M > 0 && K > 0 && Nc > 0 && (K & 3) == 0 && !pw_thin_ok(0, 0, M, K, Nc) && !pw_wide_ok(M, K, Nc, false) && !pw_wide_ok(M, K, Nc, true) && nt_x6(M, K, Nc) != 0 && ai >= min_ai &&
           getenv("MNY_GEMM_V1") == nullptr;
}
extern "C" int mny_pw_w6_supported(int64_t M, int K, int Nc) { return w6_ok(M, K, Nc) ? 1 : 0; }
extern "C" size_t mny_pw_w6_bytes(int K, int Nc) { return (K <= 0 || Nc <= 0) ? 0 : (size_t)3 * Nc * ((K + 15) / 16) * 32; }
extern "C" int mny_cut3_batch(const mny_cut3_job* jobs, const int32_t* block_job, int nblocks, void* stream) {
    MNY_REQUIRE(jobs && block_job && nblocks > 0, "cut3_batch: bad arguments");
    static_assert(sizeof(mny_cut3_job) == sizeof(Cut3Job), "job layout");
    hipLaunchKernelGGL(cut3_batch_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const Cut3Job*>(jobs), block_job);
    return check_launch("cut3_batch_kernel");
}
extern "C" int mny_pw_fwd_w6(const float* x, const float* in_scale, const float* in_shift, int in_act, const void* w6, const float* bias,
                             const float* addend, float* y, float* stats, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(x && w6 && y, "pw_fwd_w6: null pointer");
    MNY_REQUIRE(w6_ok(M, K, Nc), "pw_fwd_w6: unsupported problem M=%lld K=%d N=%d (mny_pw_w6_supported)", (long long)M, K, Nc);
    MNY_REQUIRE(!(stats && bias), "pw_fwd_w6: stats and bias are mutually exclusive");
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    Nt2Plan p2 = nt2_plan(M, K, Nc, xf);                          // same tiling and partial rows as mny_pw_fwd
#if MNY_W6_SWP
    // software-pipelined form: three-slot A ring + THREE-slot plane ring (24 + 9 TN KB) + the scale / shift cache
    const int Kpad16 = (int)cdiv(K, 16) * 16;
    const size_t lds = (size_t)(3 * BM * 16 + 3 * 32 * p2.TN * 24) * sizeof(float) + (xf ? 2 * Kpad16 * sizeof(float) : 0);
#else
    const size_t lds = p2.lds;                                    // three-slot A ring + two-slot plane ring = the plain mode's three-slot (A + fp32 B) ring, to the byte
#endif
    MNY_REQUIRE(lds <= 96 * 1024, "pw_fwd_w6: K=%d too large for the LDS scale cache", K);
    Gemm2Args g{x, in_scale, in_shift, in_act, w6, bias, addend, y, stats, M, K, Nc, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    const int XF = !xf ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1);
    const Nt2Kernel k = nt2_kernel(p2.TN, XF, 0, 3);
    g_pw_route = MNY_ROUTE_DMA_X6;
    nt2_allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(p2.grid), dim3(256), lds, (hipStream_t)stream, g);
    return check_launch("pw_gemm_nt_dma_kernel<w6>");
}
// data gradient + BN-backward sums of the fed unit (mny_pw_dgrad_bnred / _add semantics; addend may be NULL) on pre-cut W^T planes
extern "C" int mny_pw_dgrad_bnred_w6(const float* dy, const void* wT6, const float* addend, float* dx, const float* y, const float* scale,
                                     const float* shift, int act, const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc,
                                     void* stream) {
    MNY_REQUIRE(dy && wT6 && dx && y && scale && shift && mean && invstd && red, "pw_dgrad_bnred_w6: null pointer");
    MNY_REQUIRE(w6_ok(M, K, Nc) && dgrad_bnred_ok(M, K, Nc, act), "pw_dgrad_bnred_w6: unsupported problem M=%lld K=%d N=%d", (long long)M, K, Nc);
    Nt2Plan p2 = nt2_plan(M, K, Nc, false, 0, kRedMaxTn);
    MNY_REQUIRE(!addend || p2.TN <= 3, "pw_dgrad_bnred_w6: the addend form needs a column tile of <= 96 (mny_pw_dgrad_bnred_add_supported)");
    const size_t lds = p2.lds;
    MNY_REQUIRE(lds <= 96 * 1024, "pw_dgrad_bnred_w6: K=%d too large", K);
    Gemm2Args g{dy, nullptr, nullptr, MNY_ACT_NONE, wT6, nullptr, addend, dx, red, M, K, Nc, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                y, scale, shift, mean, invstd, act};
    Nt2Kernel k;
    if (addend) switch (p2.TN) {
        case 1: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, 0, 2, 3>; break; case 2: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, 0, 2, 3>; break;
        default: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, 0, 2, 3>; break;
    }
    else switch (p2.TN) {
        case 1: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<1, 0, 0, 1, 3>; break; case 2: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<2, 0, 0, 1, 3>; break;
        case 3: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<3, 0, 0, 1, 3>; break; default: k = (Nt2Kernel)pw_gemm_nt_dma_kernel<4, 0, 0, 1, 3>; break;
    }
    g_pw_route = MNY_ROUTE_DMA_X6;
    nt2_allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(p2.grid), dim3(256), lds, (hipStream_t)stream, g);
    return check_launch("pw_gemm_nt_dma_kernel<RED, w6>");
}

extern "C" int mny_pw_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* w,
                               const float* bias, const void* addend, void* y, float* stats, int64_t M, int K, int Nc,
                               void* stream) {
    MNY_REQUIRE(x && w && y, "pw_fwd: null pointer");
    MNY_REQUIRE(M > 0 && K > 0 && Nc > 0, "pw_fwd: empty problem");
    MNY_REQUIRE(!(stats && bias), "pw_fwd: stats and bias are mutually exclusive");
    hipStream_t st = (hipStream_t)stream;
    const bool xf = in_scale != nullptr || in_act != MNY_ACT_NONE;
    static const bool force_v1 = getenv("MNY_GEMM_V1") != nullptr;
    // mny_pw_stat_parts_bf16 sizes the caller's partial rows from (M, K, Nc) alone: a view / bias that moves the call to another kernel family
    // than the query assumed would write a different number of rows than mny_bn_finalize sums — refuse it instead
    MNY_REQUIRE(!stats || pwt_ok(M, K, Nc, in_act, bias != nullptr) == pwt_ok(M, K, Nc, MNY_ACT_NONE, false),
                "pw_fwd_bf16: statistics with in_act=%d at M=%lld K=%d N=%d take another kernel family than mny_pw_stat_parts_bf16 counted rows for", in_act, (long long)M, K, Nc);
    if (pwt_ok(M, K, Nc, in_act, bias != nullptr))         // K <= 48 at a large pixel count: a wave per 16 pixels on the bf16 matrix cores (gate.hip)
        return (g_pw_route = MNY_ROUTE_WAVE16, pwt_launch(x, in_scale, in_shift, in_act, w, addend, y, stats, M, K, Nc, st));
    if (pw_thin_ok(1, 0, M, K, Nc))
        return (g_pw_route = MNY_ROUTE_THIN, pw_thin_launch(1, x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, st));
    if ((K & 7) == 0 && !force_v1) {                // LDS-DMA pipeline on the bf16 matrix cores (16-B aligned bf16 rows)
        Nt2Plan p2 = nt2_plan(M, K, Nc, xf, 1);
        MNY_REQUIRE(p2.lds <= 64 * 1024, "pw_fwd: K=%d too large for the LDS scale cache", K);
        Gemm2Args g{x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, p2.m_tiles, p2.tiles_per_block, p2.gx, p2.n_tiles,
                    nullptr, nullptr, nullptr, nullptr, nullptr, 0};
        const int XF = !xf ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1);
        g_pw_route = MNY_ROUTE_DMA_F32;
        hipLaunchKernelGGL(nt2_kernel(p2.TN, XF, 1), dim3(p2.grid), dim3(256), p2.lds, st, g);
        return check_launch("pw_gemm_nt_dma_kernel<bf16>");
    }
    g_pw_route = MNY_ROUTE_TILE_V1;
    return pw_fwd_v1<bf16_t>((const bf16_t*)x, in_scale, in_shift, in_act, (const bf16_t*)w, bias, (const bf16_t*)addend, (bf16_t*)y, stats,
                             M, K, Nc, st);
}

// which kernel family the LAST pointwise-conv call of this thread took — set by the dispatchers themselves where they decide, so it cannot
// drift from them (the predictor below ignores h-swish views / bias gradients).  engine.py records it for every call of a plan's
// first replay; NetPlan.kernel_routes() reports those.
extern "C" int mny_pw_last_route(void) { return g_pw_route; }

// PREDICTED kernel family of a call with a plain view (no h-swish input, no bias gradient): the predicates of the three dispatchers in
// their order.  For shapes no plan at hand launches (tests ask it about the bs-256 shapes while running a bs-64 plan).
extern "C" int mny_pw_route(int op, int bf16, int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0 || op < 0 || op > 2) return MNY_EINVAL;
    static const bool gemm_v1 = getenv("MNY_GEMM_V1") != nullptr, wgrad_v1 = getenv("MNY_WGRAD_V1") != nullptr;
    const int al = bf16 ? 7 : 3;
    if (op == 2) {
        if (!bf16 && pw_wgs_ok(M, K, Nc)) return MNY_ROUTE_WGRAD_STREAM;
        if ((Nc & al) || (K & al) || wgrad_v1) return MNY_ROUTE_TILE_V1;
        return (!bf16 && nt_x6(M, K, Nc)) ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32;
    }
    if (bf16 && op == 0 && pwt_ok(M, K, Nc, MNY_ACT_NONE, false)) return MNY_ROUTE_WAVE16;
    if (pw_thin_ok(bf16 ? 1 : 0, op, M, K, Nc)) return MNY_ROUTE_THIN;
    if (!bf16 && pw_wide_ok(M, K, Nc, op == 1)) return MNY_ROUTE_WIDE;
    if ((K & al) || gemm_v1) return MNY_ROUTE_TILE_V1;
    return (!bf16 && nt_x6(M, K, Nc)) ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32;
}

extern "C" size_t mny_pw_wgrad_ws_floats(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return 0;
    WgPlan pl = wg_plan(M, K, Nc);
    size_t splits = (size_t)pl.splits;
    if (pw_wgs_ok(M, K, Nc) && (size_t)pw_wgs_splits(M, K, Nc) > splits) splits = (size_t)pw_wgs_splits(M, K, Nc);     // whichever kernel the call takes
    return splits * Nc * K + (size_t)colsum_parts(M) * Nc;
}

// (without a bias gradient the narrow-sided shapes run the stream kernel of pwwgs.hip: its partial-row count)
extern "C" int mny_pw_wgrad_splits(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return MNY_EINVAL;
    return pw_wgs_ok(M, K, Nc) ? pw_wgs_splits(M, K, Nc) : wg_plan(M, K, Nc, false).splits;
}
extern "C" int mny_pw_wgrad_splits_bf16(int64_t M, int K, int Nc) { return (M <= 0 || K <= 0 || Nc <= 0) ? MNY_EINVAL : wg_plan(M, K, Nc, true).splits; }

template <typename T>
static int pw_wgrad_impl(const T* x, const float* in_scale, const float* in_shift, int in_act, const T* dy,
                         float* dw, float* dbias, float* ws, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(x && dy && ws, "pw_wgrad: null pointer");
    MNY_REQUIRE(dw || !dbias, "pw_wgrad: a deferred combine (dw == NULL) cannot carry a bias gradient");
    MNY_REQUIRE(M > 0 && K > 0 && Nc > 0, "pw_wgrad: empty problem");
    constexpr bool is_f32 = sizeof(T) == 4;
    if (is_f32 && pw_wgs_ok(M, K, Nc) && !dbias) {              // narrow-sided shape: barrier-free stream kernel (pwwgs.hip), partial rows [pw_wgs_splits][Nc][K]
        g_pw_route = MNY_ROUTE_WGRAD_STREAM;
        int rc = pw_wgs_launch((const float*)x, in_scale, in_shift, in_act, (const float*)dy, ws, M, K, Nc, (hipStream_t)stream);
        if (rc || !dw) return rc;
        const int64_t n = (int64_t)Nc * K;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(n, 32)), dim3(256), 0, (hipStream_t)stream, ws, pw_wgs_splits(M, K, Nc), n, dw);
        return check_launch("reduce_parts_kernel");
    }
    static const bool force_v1 = getenv("MNY_WGRAD_V1") != nullptr;
    const bool dma_f32 = is_f32 && (Nc & 3) == 0 && (K & 3) == 0 && in_act != MNY_ACT_HSWISH && !force_v1;
    WgPlan pl = wg_plan(M, K, Nc, !is_f32, dma_f32);
    WgradArgs a{x, in_scale, in_shift, in_act, dy, ws, M, K, Nc, pl.rows_per_block, 0, pl.gy, pl.splits};
    dim3 grid(pl.gx, pl.gy, pl.splits), block(256);
    static const bool xcd_order = getenv("MNY_WGRAD_NO_XCD") == nullptr;      // (same-box A/B switch)
    WgradArgs a_dma = a;
    // (only with >= 64 splits, i.e. >= 8 per XCD: K1280 N512 — 40 tiles, 38 splits — lost 11 % to XCD imbalance)
    const bool use_xcd = xcd_order && pl.gx * pl.gy > 1 && pl.splits >= 64;
    if (use_xcd) a_dma.gx = pl.gx;
    const dim3 grid_dma = use_xcd ? dim3((unsigned)(cdiv(pl.splits, 8) * 8 * pl.gx * pl.gy)) : grid;
    hipStream_t st = (hipStream_t)stream;
    WgKernel dk = nullptr;                       // LDS-DMA kernels read raw 16-B chunks: aligned rows only
    if (is_f32) {
        if (dma_f32) dk = wg_dma_kernel(pl.mode, pl.TI, pl.TJ, nt_x6(M, K, Nc));
        if (dk && pl.lds_dma > 64 * 1024 && !allow_lds((const void*)dk, 160 * 1024)) { set_error("pw_wgrad: hipFuncSetAttribute failed"); return MNY_EHIP; }
    } else if ((Nc & 7) == 0 && (K & 7) == 0 && !force_v1) {
        const int XF = (in_scale == nullptr && in_act == MNY_ACT_NONE) ? 0 : (in_act == MNY_ACT_HSWISH ? 2 : 1);
        dk = wg_bf16_kernel(pl.mode, pl.TI, pl.TJ, XF);
        if (dk && pl.lds_dma > 64 * 1024) {      // > 64 KB of dynamic LDS needs a per-kernel opt-in (once)
            if (!allow_lds((const void*)dk, 160 * 1024)) {
                set_error("pw_wgrad: hipFuncSetAttribute failed"); return MNY_EHIP;
            }
        }
    }
    const int key = dk ? -1 : pl.mode * 100 + pl.TI * 10 + pl.TJ;
    g_pw_route = !dk ? MNY_ROUTE_TILE_V1 : ((is_f32 && nt_x6(M, K, Nc)) ? MNY_ROUTE_DMA_X6 : MNY_ROUTE_DMA_F32);
    if (dk) hipLaunchKernelGGL(dk, grid_dma, block, pl.lds_dma, st, a_dma);
#define MNY_WG(MD, I, J) hipLaunchKernelGGL((pw_wgrad_kernel<T, MD, I, J>), grid, block, pl.lds, st, a)
    switch (key) {
        case -1: break;
        case 11: MNY_WG(0, 1, 1); break; case 12: MNY_WG(0, 1, 2); break; case 21: MNY_WG(0, 2, 1); break; case 22: MNY_WG(0, 2, 2); break;
        case 111: MNY_WG(1, 1, 1); break; case 112: MNY_WG(1, 1, 2); break; case 113: MNY_WG(1, 1, 3); break; case 114: MNY_WG(1, 1, 4); break;
        case 115: MNY_WG(1, 1, 5); break; case 116: MNY_WG(1, 1, 6); break; case 121: MNY_WG(1, 2, 1); break; case 122: MNY_WG(1, 2, 2); break;
        case 123: MNY_WG(1, 2, 3); break; case 131: MNY_WG(1, 3, 1); break; case 132: MNY_WG(1, 3, 2); break; case 141: MNY_WG(1, 4, 1); break;
        case 151: MNY_WG(1, 5, 1); break; case 161: MNY_WG(1, 6, 1); break;
        default: set_error("pw_wgrad: no kernel for mode %d tile %dx%d", pl.mode, pl.TI, pl.TJ); return MNY_EUNSUPPORTED;
    }
#undef MNY_WG
    int rc = check_launch("pw_wgrad_kernel");
    if (rc) return rc;
    if (!dw) return MNY_OK;                         // partials only: the caller combines them (mny_reduce_batch)
    const int64_t n = (int64_t)Nc * K;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(n, 32)), dim3(256), 0, st, ws, pl.splits, n, dw);
    rc = check_launch("reduce_parts_kernel");
    if (rc) return rc;
    if (dbias) {
        const int parts = colsum_parts(M);
        float* cs = ws + (size_t)pl.splits * Nc * K;
        const int64_t rpb = cdiv(M, parts);
        hipLaunchKernelGGL((colsum_kernel<T>), dim3(parts, (unsigned)cdiv(Nc, 64)), dim3(256), 0, st, dy, cs, M, Nc, rpb);
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(Nc, 32)), dim3(256), 0, st, cs, parts, (int64_t)Nc, dbias);
        rc = check_launch("colsum");
    }
    return rc;
}
extern "C" int mny_pw_wgrad(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                            float* dw, float* dbias, float* ws, int64_t M, int K, int Nc, void* stream) {
    return pw_wgrad_impl<float>(x, in_scale, in_shift, in_act, dy, dw, dbias, ws, M, K, Nc, stream);
}
extern "C" int mny_pw_wgrad_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* dy,
                                 float* dw, float* dbias, float* ws, int64_t M, int K, int Nc, void* stream) {
    return pw_wgrad_impl<bf16_t>((const bf16_t*)x, in_scale, in_shift, in_act, (const bf16_t*)dy, dw, dbias, ws, M, K, Nc, stream);
}

// every W^T of a backward pass in ONE launch: block b serves 32x32 tile (b - job.block0) of job block_job[b]
template <typename T>
__global__ void transpose_batch_kernel(const mny_transpose_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    __shared__ float tile[32][33];
    const mny_transpose_job jb = jobs[block_job[blockIdx.x]];
    const int local = (int)blockIdx.x - jb.block0, tx = (jb.Cc + 31) / 32;
    const int bx = (local % tx) * 32, by = (local / tx) * 32;
    const float* src = jb.src;
    T* dst = (T*)jb.dst;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < jb.R && c < jb.Cc) ? src[(int64_t)r * jb.Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < jb.Rp && c < jb.Cc) st1(dst + (int64_t)c * jb.Rp + r, r < jb.R ? tile[threadIdx.x][j] : 0.f);
    }
}
extern "C" int mny_transpose_batch(const mny_transpose_job* jobs, const int32_t* block_job, int nblocks, void* stream) {
    MNY_REQUIRE(jobs && block_job && nblocks > 0, "transpose_batch: bad arguments");
    hipLaunchKernelGGL((transpose_batch_kernel<float>), dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream, jobs, block_job);
    return check_launch("transpose_batch_kernel");
}
extern "C" int mny_transpose_batch_bf16(const mny_transpose_job* jobs, const int32_t* block_job, int nblocks, void* stream) {
    MNY_REQUIRE(jobs && block_job && nblocks > 0, "transpose_batch: bad arguments");
    hipLaunchKernelGGL((transpose_batch_kernel<bf16_t>), dim3((unsigned)nblocks), dim3(32, 8), 0, (hipStream_t)stream, jobs, block_job);
    return check_launch("transpose_batch_kernel<bf16>");
}

extern "C" int mny_transpose(const float* src, float* dst, int R, int Cc, void* stream) {
    MNY_REQUIRE(src && dst && R > 0 && Cc > 0, "transpose: bad arguments");
    hipLaunchKernelGGL((transpose_kernel<float>), dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)stream, src, dst, R, Cc, R);
    return check_launch("transpose_kernel");
}
extern "C" int mny_transpose_pad(const float* src, float* dst, int R, int Cc, int Rp, void* stream) {
    MNY_REQUIRE(src && dst && R > 0 && Cc > 0 && Rp >= R, "transpose_pad: bad arguments");
    hipLaunchKernelGGL((transpose_kernel<float>), dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(Rp, 32)), dim3(32, 8), 0, (hipStream_t)stream, src, dst, R, Cc, Rp);
    return check_launch("transpose_kernel<pad>");
}
extern "C" int mny_transpose_pad_bf16(const float* src, void* dst, int R, int Cc, int Rp, void* stream) {
    MNY_REQUIRE(src && dst && R > 0 && Cc > 0 && Rp >= R, "transpose_pad: bad arguments");
    hipLaunchKernelGGL((transpose_kernel<bf16_t>), dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(Rp, 32)), dim3(32, 8), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, R, Cc, Rp);
    return check_launch("transpose_kernel<pad,bf16>");
}
extern "C" int mny_transpose_bf16(const float* src, void* dst, int R, int Cc, void* stream) {
    MNY_REQUIRE(src && dst && R > 0 && Cc > 0, "transpose: bad arguments");
    hipLaunchKernelGGL((transpose_kernel<bf16_t>), dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, R, Cc, R);
    return check_launch("transpose_kernel<bf16>");
}

// ---- fused BN-backward + wgrad + dgrad for thin "expand" units ------------------------------------------------
extern "C" int mny_pw_bnbwd_supported(int64_t M, int K, int Nc) { return bnw_supported(M, K, Nc) ? 1 : 0; }

extern "C" size_t mny_pw_bnbwd_ws_floats(int64_t M, int K, int Nc) {
    if (!bnw_supported(M, K, Nc)) return 0;
    BnwPlan pl = bnw_plan(M, K, Nc);
    // partials + reduced row + B1 + Q + bias + 1-bit activation mask (M/2 * TI 64-bit words); the wave form of bf16 storage (gate.hip) lays its own
    // buffers out in the same workspace: the larger of the two
    const size_t base = (size_t)(pl.splits + 1) * bnw_stride(Nc, K) + (size_t)Nc * K + (size_t)K * K + 64 + 64 + (size_t)((M / 2 + 66) / 2 * 2) * pl.TI * 2;
    const size_t wave = pwe_ok(M, K, Nc) ? pwe_ws_floats(M, K, Nc) : 0;
    return base > wave ? base : wave;
}

static bool bnw_red_ok(int64_t M, int K, int Nc) {      // the data-gradient stage that can carry the sums of the unit in front: the barrier-free second generation
    static const int v2 = getenv("MNY_BND_V2") ? atoi(getenv("MNY_BND_V2")) : 1;
    return bnw_supported(M, K, Nc) && v2 && (Nc == 96 || Nc == 144 || Nc == 192);
}
static int bnw_red_grid(int64_t M) {
    static const int v2_grid = getenv("MNY_BND_V2_GRID") ? atoi(getenv("MNY_BND_V2_GRID")) : 512;
    const int64_t tiles = cdiv(M, 32);
    return (int)(cdiv(tiles, 4) < v2_grid ? cdiv(tiles, 4) : v2_grid);
}
static int pw_bnbwd_impl(const float* g, const float* y, const float* scale, const float* shift, int act,
                         const float* mean, const float* invstd, const float* gamma,
                         const float* x, const float* in_scale, const float* in_shift, int in_act,
                         const float* w, const float* addend, float* dx /* may be NULL: no data gradient */,
                         float* dw, float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, void* stream,
                         const float* ry = nullptr, const float* r_scale = nullptr, const float* r_shift = nullptr, int r_act = 0,
                         const float* r_mean = nullptr, const float* r_invstd = nullptr, float* red_out = nullptr) {
    MNY_REQUIRE(g && y && scale && shift && mean && invstd && gamma && x && w && dw && dgamma && dbeta && ws, "pw_bnbwd: null pointer");
    MNY_REQUIRE(!red_out || (dx && ry && r_scale && r_shift && r_mean && r_invstd && r_act >= MNY_ACT_NONE && r_act < MNY_ACT_HSWISH && bnw_red_ok(M, K, Nc)),
                "pw_bnbwd_red: incomplete reduction target or unsupported shape (mny_pw_bnbwd_red_supported)");
    MNY_REQUIRE(bnw_supported(M, K, Nc), "pw_bnbwd: shape M=%lld K=%d N=%d not supported (need K<=32, N<=192, N>K)", (long long)M, K, Nc);
    MNY_REQUIRE(in_act != MNY_ACT_HSWISH && in_act != MNY_ACT_HSIGMOID && act != MNY_ACT_HSWISH && act != MNY_ACT_HSIGMOID,
                "pw_bnbwd: h-swish / h-sigmoid activations are not supported");
    BnwPlan pl = bnw_plan(M, K, Nc);
    hipStream_t st = (hipStream_t)stream;
    const int64_t stride = bnw_stride(Nc, K);
    float* red = ws + (size_t)pl.splits * stride;
    float* B1 = red + stride;
    float* Q = B1 + (size_t)Nc * K;
    float* bias = Q + (size_t)K * K;
    size_t moff = (size_t)(bias + 64 - ws);
    moff = (moff + 15) / 16 * 16;                                      // 64-byte aligned mask words
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + moff);
    const int64_t npairs = (M / 2 + 66) / 2 * 2;                      // even (16-B aligned rows of words) + a tile of slack
    BnwArgs a{g, y, scale, shift, act, mean, invstd, x, in_scale, in_shift, in_act, ws, mask, npairs, M, K, Nc, pl.rows_per_block, pl.TI};
    dim3 grid(pl.splits, pl.nsl), block(256);
    if (!allow_lds((const void*)pw_bnbwd_stage1_kernel<5>, 96 * 1024) || !allow_lds((const void*)pw_bnbwd_stage1_kernel<6>, 96 * 1024)) {     // TI >= 5 needs more than 64 KB of dynamic LDS
        set_error("pw_bnbwd: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    static const bool s1v2 = getenv("MNY_BNW_S1V2") == nullptr || atoi(getenv("MNY_BNW_S1V2")) != 0;     // (=0: first-generation stage 1, A/B)
    if (s1v2) {
        if (!allow_lds((const void*)pw_bnbwd_stage1b_kernel<5>, 96 * 1024) || !allow_lds((const void*)pw_bnbwd_stage1b_kernel<6>, 96 * 1024)) {
            set_error("pw_bnbwd: hipFuncSetAttribute failed"); return MNY_EHIP;
        }
        switch (pl.TIs) {
            case 1: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<1>), grid, block, pl.lds1, st, a); break;
            case 2: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<2>), grid, block, pl.lds1, st, a); break;
            case 3: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<3>), grid, block, pl.lds1, st, a); break;
            case 4: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<4>), grid, block, pl.lds1, st, a); break;
            case 5: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<5>), grid, block, pl.lds1, st, a); break;
            default: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<6>), grid, block, pl.lds1, st, a); break;
        }
    } else switch (pl.TIs) {
        case 1: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<1>), grid, block, pl.lds1, st, a); break;
        case 2: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<2>), grid, block, pl.lds1, st, a); break;
        case 3: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<3>), grid, block, pl.lds1, st, a); break;
        case 4: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<4>), grid, block, pl.lds1, st, a); break;
        case 5: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<5>), grid, block, pl.lds1, st, a); break;
        default: hipLaunchKernelGGL((pw_bnbwd_stage1_kernel<6>), grid, block, pl.lds1, st, a); break;
    }
    int rc = check_launch("pw_bnbwd_stage1_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(stride, 32)), dim3(256), 0, st, ws, pl.splits, stride, red);
    hipLaunchKernelGGL(pw_bnbwd_finalize_kernel, dim3((unsigned)cdiv((int64_t)Nc * K, 256)), dim3(256), bnw_finalize_lds(Nc, K), st, red, w, gamma, mean, invstd,
                       (double)M, Nc, K, dw, dgamma, dbeta, B1, Q, bias);
    rc = check_launch("pw_bnbwd_finalize_kernel");
    if (rc || !dx) return rc;
    if (!allow_lds((const void*)pw_bnbwd_dgrad_kernel, 96 * 1024)) {
        set_error("pw_bnbwd: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    BndArgs d{g, mask, act, x, in_scale, in_shift, in_act, B1, Q, bias, addend, dx, npairs, M, Nc, K, pl.TI, pl.m_tiles, pl.tiles_per_block,
              ry, r_scale, r_shift, r_mean, r_invstd, r_act, red_out};
    if (bnw_red_ok(M, K, Nc)) {                                    // second generation: barrier-free direct fragment loads (N % 16 == 0 instantiations)
        const int grid = bnw_red_grid(M);
        if (red_out) {
            if (Nc == 96) hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<6, float, false, true>), dim3(grid), dim3(256), 0, st, d);
            else if (Nc == 144) hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<9, float, false, true>), dim3(grid), dim3(256), 0, st, d);
            else hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<12, float, false, true>), dim3(grid), dim3(256), 0, st, d);
            return check_launch("pw_bnbwd_dgrad2_kernel<red>");
        }
        if (Nc == 96) hipLaunchKernelGGL(pw_bnbwd_dgrad2_kernel<6>, dim3(grid), dim3(256), 0, st, d);
        else if (Nc == 144) hipLaunchKernelGGL(pw_bnbwd_dgrad2_kernel<9>, dim3(grid), dim3(256), 0, st, d);
        else hipLaunchKernelGGL(pw_bnbwd_dgrad2_kernel<12>, dim3(grid), dim3(256), 0, st, d);
        return check_launch("pw_bnbwd_dgrad2_kernel");
    }
    hipLaunchKernelGGL(pw_bnbwd_dgrad_kernel, dim3(pl.gx2), dim3(256), pl.lds2, st, d);
    return check_launch("pw_bnbwd_dgrad_kernel");
}
extern "C" int mny_pw_bnbwd(const float* g, const float* y, const float* scale, const float* shift, int act,
                            const float* mean, const float* invstd, const float* gamma,
                            const float* x, const float* in_scale, const float* in_shift, int in_act,
                            const float* w, const float* addend, float* dx /* may be NULL: no data gradient */,
                            float* dw, float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, void* stream) {
    return pw_bnbwd_impl(g, y, scale, shift, act, mean, invstd, gamma, x, in_scale, in_shift, in_act, w, addend, dx, dw, dgamma, dbeta, ws, M, K, Nc, stream);
}
// ... whose data gradient completes the output gradient of the conv+BN+act unit in front (raw output ry [M][K], view r_scale / r_shift / r_act of the clamp
// family, statistics r_mean / r_invstd): that unit's BN-backward sums leave with it, red[mny_pw_bnbwd_red_parts(M, K, Nc)][2][K] (fp32 storage, N in {96, 144, 192})
extern "C" int mny_pw_bnbwd_red_supported(int64_t M, int K, int Nc) { return bnw_red_ok(M, K, Nc) ? 1 : 0; }
extern "C" int mny_pw_bnbwd_red_parts(int64_t M, int K, int Nc) { return bnw_red_ok(M, K, Nc) ? bnw_red_grid(M) : MNY_EINVAL; }
extern "C" int mny_pw_bnbwd_red(const float* g, const float* y, const float* scale, const float* shift, int act,
                                const float* mean, const float* invstd, const float* gamma,
                                const float* x, const float* in_scale, const float* in_shift, int in_act,
                                const float* w, const float* addend, float* dx, float* dw, float* dgamma, float* dbeta, float* ws,
                                const float* ry, const float* r_scale, const float* r_shift, int r_act, const float* r_mean, const float* r_invstd, float* red,
                                int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(red && dx, "pw_bnbwd_red: null pointer");
    return pw_bnbwd_impl(g, y, scale, shift, act, mean, invstd, gamma, x, in_scale, in_shift, in_act, w, addend, dx, dw, dgamma, dbeta, ws, M, K, Nc, stream,
                         ry, r_scale, r_shift, r_act, r_mean, r_invstd, red);
}

// ---- bf16-storage twin (round 3): G, Y, X, addend, dX are bf16; W, statistics, dW / dgamma / dbeta, the workspace stay fp32 --------------
// Shapes: K in {8, 16, 24, 32}, N in {64, 72, 96, 144, 192} (the expand units of MobileNetV3 / MobileNetV2), M >= 4096.  Same three steps:
// second-generation stage 1 on a bf16 LDS image, the fp32 finalize, the barrier-free data-gradient stage with widened 8-byte loads.
static bool bnw_supported_bf16(int64_t M, int K, int N) {
    // (stage 1 is instantiated for up to three column tiles per slice: with MNY_BNW_NOSLICE set the 144 / 192-column units would need
    // five / six and every replay would fail — they are then not offered at all)
    return bnw_supported(M, K, N) && (K & 7) == 0 && (N == 64 || N == 72 || N == 96 || N == 144 || N == 192) && bnw_plan(M, K, N).TIs <= 3;
}
extern "C" int mny_pw_bnbwd_supported_bf16(int64_t M, int K, int Nc) { return bnw_supported_bf16(M, K, Nc) ? 1 : 0; }
extern "C" int mny_pw_bnbwd_bf16(const void* g, const void* y, const float* scale, const float* shift, int act,
                                 const float* mean, const float* invstd, const float* gamma,
                                 const void* x, const float* in_scale, const float* in_shift, int in_act,
                                 const float* w, const void* addend, void* dx /* may be NULL: no data gradient */,
                                 float* dw, float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(g && y && scale && shift && mean && invstd && gamma && x && w && dw && dgamma && dbeta && ws, "pw_bnbwd_bf16: null pointer");
    MNY_REQUIRE(bnw_supported_bf16(M, K, Nc), "pw_bnbwd_bf16: shape M=%lld K=%d N=%d not supported (mny_pw_bnbwd_supported_bf16)", (long long)M, K, Nc);
    MNY_REQUIRE(in_act != MNY_ACT_HSWISH && in_act != MNY_ACT_HSIGMOID && act != MNY_ACT_HSWISH && act != MNY_ACT_HSIGMOID,
                "pw_bnbwd_bf16: h-swish / h-sigmoid activations are not supported");
    hipStream_t st = (hipStream_t)stream;
    if (pwe_ok(M, K, Nc))                                    // wave form on the bf16 matrix cores (gate.hip)
        return pwe_launch(g, y, scale, shift, act, mean, invstd, gamma, x, in_scale, in_shift, in_act, w, addend, dx, dw, dgamma, dbeta, ws, M, K, Nc, st);
    BnwPlan pl = bnw_plan(M, K, Nc);
    const int64_t stride = bnw_stride(Nc, K);
    float* red = ws + (size_t)pl.splits * stride;
    float* B1 = red + stride;
    float* Q = B1 + (size_t)Nc * K;
    float* bias = Q + (size_t)K * K;
    size_t moff = (size_t)(bias + 64 - ws);
    moff = (moff + 15) / 16 * 16;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + moff);
    const int64_t npairs = (M / 2 + 66) / 2 * 2;
    BnwArgs a{(const float*)g, (const float*)y, scale, shift, act, mean, invstd, (const float*)x, in_scale, in_shift, in_act, ws, mask, npairs, M, K, Nc,
              pl.rows_per_block, pl.TI};
    dim3 grid(pl.splits, pl.nsl), block(256);
    // LDS: the bf16 ring is half the fp32 one, but the block reductions at the end use [3][16][64] + vector scratch in fp32
    static const int deep = getenv("MNY_BNW_RING") ? atoi(getenv("MNY_BNW_RING")) : 3;       // stages of the bf16 ring: MNY_BNW_RING=6 measured round 5, same box: 15.24 / 15.24 vs 15.20 / 15.25 ms — no gain, 3 stays
    const int S1 = deep == 3 ? 3 : 6;
    size_t lds1 = (size_t)S1 * 16 * (2 * 32 * pl.TIs + 32) * sizeof(bf16_t);
    const size_t need = (size_t)4 * (2 * pl.TIs + 1) * 32 * sizeof(float);
    if (lds1 < need) lds1 = need;
    if (lds1 < 3 * 16 * 64 * sizeof(float)) lds1 = 3 * 16 * 64 * sizeof(float);
    switch (pl.TIs * 10 + S1) {
        case 23: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<2, bf16_t, 3>), grid, block, lds1, st, a); break;
        case 33: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<3, bf16_t, 3>), grid, block, lds1, st, a); break;
        case 26: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<2, bf16_t, 6>), grid, block, lds1, st, a); break;
        case 36: hipLaunchKernelGGL((pw_bnbwd_stage1b_kernel<3, bf16_t, 6>), grid, block, lds1, st, a); break;
        default: set_error("pw_bnbwd_bf16: no stage-1 kernel for %d column tiles", pl.TIs); return MNY_EUNSUPPORTED;
    }
    int rc = check_launch("pw_bnbwd_stage1b_kernel<bf16>");
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(stride, 32)), dim3(256), 0, st, ws, pl.splits, stride, red);
    hipLaunchKernelGGL(pw_bnbwd_finalize_kernel, dim3((unsigned)cdiv((int64_t)Nc * K, 256)), dim3(256), bnw_finalize_lds(Nc, K), st, red, w, gamma, mean, invstd,
                       (double)M, Nc, K, dw, dgamma, dbeta, B1, Q, bias);
    rc = check_launch("pw_bnbwd_finalize_kernel");
    if (rc || !dx) return rc;
    BndArgs d{(const float*)g, mask, act, (const float*)x, in_scale, in_shift, in_act, B1, Q, bias, (const float*)addend, (float*)dx, npairs, M, Nc, K, pl.TI,
              pl.m_tiles, pl.tiles_per_block};
    static const int v2_grid = getenv("MNY_BND_V2_GRID") ? atoi(getenv("MNY_BND_V2_GRID")) : 512;
    const int64_t tiles = cdiv(M, 32);
    const int grid2 = (int)(cdiv(tiles, 4) < v2_grid ? cdiv(tiles, 4) : v2_grid);
    switch (Nc) {
        case 64: hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<4, bf16_t, false>), dim3(grid2), dim3(256), 0, st, d); break;
        case 72: hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<5, bf16_t, true>), dim3(grid2), dim3(256), 0, st, d); break;
        case 96: hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<6, bf16_t, false>), dim3(grid2), dim3(256), 0, st, d); break;
        case 144: hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<9, bf16_t, false>), dim3(grid2), dim3(256), 0, st, d); break;
        default: hipLaunchKernelGGL((pw_bnbwd_dgrad2_kernel<12, bf16_t, false>), dim3(grid2), dim3(256), 0, st, d); break;
    }
    return check_launch("pw_bnbwd_dgrad2_kernel<bf16>");
}
