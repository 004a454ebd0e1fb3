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
#include "common.h"

namespace mny {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int LDP = BK + 4;   // LDS row pitch (floats)

struct GemmArgs {
    const float* A; const float* in_scale; const float* in_shift; int in_act;
    const float* B; const float* bias; const float* addend; float* C; float* stats;
    int64_t M; int K; int N;
    int m_tiles; int tiles_per_block;
};

template <int TN>
__global__ __launch_bounds__(256) void pw_gemm_nt_kernel(GemmArgs p) {
    constexpr int BN = 32 * TN;
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

    for (int k = tid; k < Kpad; k += 256) {
        sScale[k] = (has_xf && k < p.K) ? p.in_scale[k] : 1.f;
        sShift[k] = (has_xf && k < p.K) ? p.in_shift[k] : 0.f;
    }
    __syncthreads();

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
                    const float* src = p.A + m * p.K + k;
                    if (kvec) v = ld4(src);
                    else { v.x = src[0]; if (k + 1 < p.K) v.y = src[1]; if (k + 2 < p.K) v.z = src[2]; if (k + 3 < p.K) v.w = src[3]; }
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
                        const float* src = p.B + (int64_t)n * p.K + k;
                        if (kvec) v = ld4(src);
                        else { v.x = src[0]; if (k + 1 < p.K) v.y = src[1]; if (k + 2 < p.K) v.z = src[2]; if (k + 3 < p.K) v.w = src[3]; }
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
                    if (p.addend) v += p.addend[row * p.N + col];
                    p.C[row * p.N + col] = v;
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

struct NtPlan { int TN; int n_tiles; int m_tiles; int gx; int tiles_per_block; size_t lds; };

static NtPlan nt_plan(int64_t M, int K, int N) {
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
    const int Kpad = (int)cdiv(K, BK) * BK;
    pl.lds = (size_t)(2 * BM * LDP + 2 * 32 * best * LDP + 2 * Kpad) * sizeof(float);
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
struct WgradArgs {
    const float* X; const float* in_scale; const float* in_shift; int in_act;
    const float* dY; float* partial;
    int64_t M; int K; int N;
    int64_t rows_per_block;
};

template <int MODE, int TI, int TJ>
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
                    const float* src = p.dY + m * p.N + co;
                    if (vecA) v = ld4(src);
                    else { v.x = src[0]; if (co + 1 < p.N) v.y = src[1]; if (co + 2 < p.N) v.z = src[2]; if (co + 3 < p.N) v.w = src[3]; }
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
                    const float* src = p.X + m * p.K + ci;
                    if (vecB) v = ld4(src);
                    else { v.x = src[0]; if (ci + 1 < p.K) v.y = src[1]; if (ci + 2 < p.K) v.z = src[2]; if (ci + 3 < p.K) v.w = src[3]; }
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

struct WgPlan { int mode, TI, TJ, gx, gy, splits; int64_t rows_per_block; size_t lds; };

static int pick_block(int c) {      // 64 or 128: minimise the padded extent, ties -> 128
    const int p64 = (int)cdiv(c, 64) * 64, p128 = (int)cdiv(c, 128) * 128;
    return p128 <= p64 ? 128 : 64;
}

static WgPlan wg_plan(int64_t M, int K, int N) {
    WgPlan pl;
    const int nco = (int)cdiv(N, 32), nci = (int)cdiv(K, 32);
    int BI, BJ, KC;
    if (nco * nci <= 6) {
        pl.mode = 1; pl.TI = nco; pl.TJ = nci; BI = 32 * nco; BJ = 32 * nci; KC = 32;
        pl.gx = pl.gy = 1;
    } else {
        pl.mode = 0; BI = pick_block(N); BJ = pick_block(K); KC = 16;
        pl.TI = BI / 64; pl.TJ = BJ / 64;
        pl.gx = (int)cdiv(N, BI); pl.gy = (int)cdiv(K, BJ);
    }
    int64_t splits = 1536 / ((int64_t)pl.gx * pl.gy);
    if (splits < 1) splits = 1;
    const int64_t max_splits = cdiv(M, 4 * KC);
    if (splits > max_splits) splits = max_splits;
    if (splits > 768) splits = 768;
    const int64_t rpb = cdiv(cdiv(M, splits), KC) * KC;
    pl.rows_per_block = rpb;
    pl.splits = (int)cdiv(M, rpb);
    size_t stage = (size_t)2 * KC * (BI + BJ) * sizeof(float);
    if (stage < 3 * 16 * 64 * sizeof(float)) stage = 3 * 16 * 64 * sizeof(float);
    pl.lds = stage + 2 * BJ * sizeof(float);
    return pl;
}

// partial rows [parts][n] -> out[n]: 32 outputs x 8 part-slices per block, fp64, fixed order
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* __restrict__ parts, int nparts, int64_t n, float* __restrict__ out) {
    __shared__ double red[8][32];
    const int ol = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + ol;
    double s0 = 0.0, s1 = 0.0;
    if (i < n) {
        int pidx = slice;
        for (; pidx + 8 < nparts; pidx += 16) { s0 += (double)parts[(int64_t)pidx * n + i]; s1 += (double)parts[(int64_t)(pidx + 8) * n + i]; }
        if (pidx < nparts) s0 += (double)parts[(int64_t)pidx * n + i];
    }
    red[slice][ol] = s0 + s1;
    __syncthreads();
    if (slice == 0 && i < n) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][ol];
        out[i] = (float)s;
    }
}

// column sums of a [M][C] matrix -> partial rows; used for the head convs' bias gradient
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ y, float* __restrict__ parts, int64_t M, int C,
                                                     int64_t rows_per_block) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int slot = threadIdx.x >> 6;
    __shared__ float red[4][64];
    const int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t m1 = min(m0 + rows_per_block, M);
    float s = 0.f;
    if (c < C)
        for (int64_t m = m0 + slot; m < m1; m += 4) s += y[m * C + c];
    red[slot][threadIdx.x & 63] = s;
    __syncthreads();
    if (slot == 0 && c < C) parts[(int64_t)blockIdx.x * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = by + j, c = bx + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = bx + j, r = by + threadIdx.x;
        if (r < R && c < Cc) dst[(int64_t)c * R + r] = tile[threadIdx.x][j];
    }
}

static int colsum_parts(int64_t M) { int64_t p = cdiv(M, 256); return (int)(p < 512 ? p : 512); }

}  // namespace mny

using namespace mny;

extern "C" int mny_pw_stat_parts(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return MNY_EINVAL;
    return nt_plan(M, K, Nc).gx;
}

extern "C" int mny_pw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                          const float* bias, const float* addend, float* y, float* stats, int64_t M, int K, int Nc,
                          void* stream) {
    MNY_REQUIRE(x && w && y, "pw_fwd: null pointer");
    MNY_REQUIRE(M > 0 && K > 0 && Nc > 0, "pw_fwd: empty problem");
    MNY_REQUIRE(!(stats && bias), "pw_fwd: stats and bias are mutually exclusive");
    NtPlan pl = nt_plan(M, K, Nc);
    MNY_REQUIRE(pl.lds <= 64 * 1024, "pw_fwd: K=%d too large for the LDS scale cache", K);
    GemmArgs a{x, in_scale, in_shift, in_act, w, bias, addend, y, stats, M, K, Nc, pl.m_tiles, pl.tiles_per_block};
    dim3 grid(pl.gx, pl.n_tiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (pl.TN) {
        case 1: hipLaunchKernelGGL((pw_gemm_nt_kernel<1>), grid, block, pl.lds, st, a); break;
        case 2: hipLaunchKernelGGL((pw_gemm_nt_kernel<2>), grid, block, pl.lds, st, a); break;
        case 3: hipLaunchKernelGGL((pw_gemm_nt_kernel<3>), grid, block, pl.lds, st, a); break;
        default: hipLaunchKernelGGL((pw_gemm_nt_kernel<4>), grid, block, pl.lds, st, a); break;
    }
    return check_launch("pw_gemm_nt_kernel");
}

extern "C" size_t mny_pw_wgrad_ws_floats(int64_t M, int K, int Nc) {
    if (M <= 0 || K <= 0 || Nc <= 0) return 0;
    WgPlan pl = wg_plan(M, K, Nc);
    return (size_t)pl.splits * Nc * K + (size_t)colsum_parts(M) * Nc;
}

extern "C" int mny_pw_wgrad(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy,
                            float* dw, float* dbias, float* ws, int64_t M, int K, int Nc, void* stream) {
    MNY_REQUIRE(x && dy && dw && ws, "pw_wgrad: null pointer");
    MNY_REQUIRE(M > 0 && K > 0 && Nc > 0, "pw_wgrad: empty problem");
    WgPlan pl = wg_plan(M, K, Nc);
    WgradArgs a{x, in_scale, in_shift, in_act, dy, ws, M, K, Nc, pl.rows_per_block};
    dim3 grid(pl.gx, pl.gy, pl.splits), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MNY_WG(MD, I, J) hipLaunchKernelGGL((pw_wgrad_kernel<MD, I, J>), grid, block, pl.lds, st, a)
    const int key = pl.mode * 100 + pl.TI * 10 + pl.TJ;
    switch (key) {
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
    const int64_t n = (int64_t)Nc * K;
    hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(n, 32)), dim3(256), 0, st, ws, pl.splits, n, dw);
    rc = check_launch("reduce_parts_kernel");
    if (rc) return rc;
    if (dbias) {
        const int parts = colsum_parts(M);
        float* cs = ws + (size_t)pl.splits * Nc * K;
        const int64_t rpb = cdiv(M, parts);
        hipLaunchKernelGGL(colsum_kernel, dim3(parts, (unsigned)cdiv(Nc, 64)), dim3(256), 0, st, dy, cs, M, Nc, rpb);
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)cdiv(Nc, 32)), dim3(256), 0, st, cs, parts, (int64_t)Nc, dbias);
        rc = check_launch("colsum");
    }
    return rc;
}

extern "C" int mny_transpose(const float* src, float* dst, int R, int Cc, void* stream) {
    MNY_REQUIRE(src && dst && R > 0 && Cc > 0, "transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(Cc, 32), (unsigned)cdiv(R, 32)), dim3(32, 8), 0, (hipStream_t)stream, src, dst, R, Cc);
    return check_launch("transpose_kernel");
}
