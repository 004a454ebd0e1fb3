// MobileNetV3's per-pixel gate ("SeModule", models/mobilenetv3.py:26-41 — its avg_pool is never called, so the module gates every pixel
// on its own) as ONE unit on bf16 storage:
//
//     t = BN3(y3)                       the project conv's output, read through its view (models/mobilenetv3.py:69)
//     h = relu(BN1(W1 t))               1x1 conv C -> R = C/4, BatchNorm, ReLU                      (:31-34)
//     gate = hsigmoid(BN2(W2 h))        1x1 conv R -> C, BatchNorm, relu6(x + 3) / 6                 (:35-37)
//     out = t * gate (+ shortcut)       (:41) and, where the block has one, the residual add of :72
//
// Un-fused this is five launches forward (two 1x1 convs with 10 / 28 / 40 hidden channels that fit no GEMM tile: 5 TFLOP/s, two BN
// finalizes, the multiply) + the add, and eleven backward.  Training-mode BatchNorm needs the batch statistics of W1 t before h exists and
// those of W2 h before the gate exists (and, backward, the sums of BN2 before those of BN1), so "one kernel" is not available: the unit is
// three streaming passes over y3 forward (statistics of W1 t, statistics of W2 h, output) and three backward (BN2 sums; BN1 sums + dW2;
// dt + dW1), with the ordinary finalize launches in between — the hidden tensors h, W2 h and their gradients are never written.
//
// Thread mapping (all passes): a WAVE owns 16 consecutive pixels and ALL channels; the products run on the matrix cores as
// D[out channel][pixel] = W[out channel][k] . X[k][pixel] with v_mfma_f32_16x16x32_bf16:
//   * lane (px = lane & 15, rg = lane >> 4) holds, of every 16-channel tile T, the four channels 16 T + 4 rg + {0..3} of its pixel — as
//     loaded (one 8-byte load per tile), as accumulator (the C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg) and as the B
//     operand of the NEXT product: the matrix core pairs element e of k-group lane >> 4 of A with the same element of B, and k is only a
//     summation index, so "k-step u, group rg, element e" is DEFINED as channel 32 u + (e < 4 ? 4 rg + e : 16 + 4 rg + e - 4) — the eight
//     accumulator registers of tiles 2u and 2u+1 ARE the B fragment: no transpose, no LDS round trip for activations, forward or backward;
//   * the weights are the A operand, cut once per pass (mny_gate_cut_batch_bf16, one launch for all gates) into that k order as bf16
//     chunks [row tile][k-step][lane]; a workgroup copies its chunk sets into LDS, one ds_read_b128 per MFMA;
//   * per-channel constants (BN scale / shift / coefficients) sit in LDS as zero-padded rows, read as float4 by (tile, rg);
//   * weight gradients contract over PIXELS, i.e. need the transposed layout: the 8 waves of a workgroup park their 16-pixel columns of
//     the two operands as bf16 in LDS ([channel][128 pixels]), and every wave owns a few 16x16 tiles of dW over those 128 pixels.
// Arithmetic = the bf16-storage GEMM path's: matrix operands rounded to bf16 (RNE), fp32 accumulate, everything else fp32; the hidden
// tensors are NOT rounded through storage any more (oracle/bf16_storage.py models exactly that).
#include <stdlib.h>

#include "common.h"
#include "x6.h"

namespace mny {

typedef float gate_f4 __attribute__((ext_vector_type(4)));

template <int C, int R>
struct GateDim {
    static constexpr int CT = (C + 15) / 16, CU = (CT + 1) / 2, CP = CU * 32;      // 16-channel tiles / 32-deep k-steps / padded extent of the wide side
    static constexpr int RT = (R + 15) / 16, RU = (RT + 1) / 2, RP = RU * 32;      // ... of the hidden side
    // chunk sets (uint4 units): A1 = W1 (rows r, k = c), A2 = W2 (rows c, k = r), A3 = W2^T (rows r, k = c), A4 = W1^T (rows c, k = r)
    static constexpr int N1 = RT * CU * 64, N2 = CT * RU * 64, O1 = 0, O2 = N1, O3 = N1 + N2, O4 = 2 * N1 + N2, NQ = 2 * (N1 + N2);
};

__device__ __forceinline__ uint32_t gate_pack2(float a, float b) { return pack_bf16x2(a, b); }

// one A-operand chunk: row `row` of a [ROWS][COLS] matrix (element (r, k) = w[r * sr + k * sk]), k-step u, lane -> 8 bf16 in the k order above
__device__ __forceinline__ uint4 gate_chunk(const float* __restrict__ w, int ROWS, int COLS, int sr, int sk, int tr, int u, int lane) {
    const int row = 16 * tr + (lane & 15), kg = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * u + (e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4));
        v[e] = (row < ROWS && k < COLS) ? w[row * sr + k * sk] : 0.f;
    }
    return make_uint4(gate_pack2(v[0], v[1]), gate_pack2(v[2], v[3]), gate_pack2(v[4], v[5]), gate_pack2(v[6], v[7]));
}

// all chunk sets of all gates of a plan in one launch: grid = (16, jobs)
__global__ __launch_bounds__(256) void gate_cut_kernel(const mny_gate_cut_job* __restrict__ jobs) {
    const mny_gate_cut_job jb = jobs[blockIdx.y];
    const int C = jb.C, R = jb.R;
    const int CT = (C + 15) / 16, CU = (CT + 1) / 2, RT = (R + 15) / 16, RU = (RT + 1) / 2;
    const int n1 = RT * CU * 64, n2 = CT * RU * 64;
    uint4* dst = reinterpret_cast<uint4*>(jb.wq);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 2 * (n1 + n2); idx += gridDim.x * 256) {
        int set, i = idx;
        if (i < n1) set = 0; else if ((i -= n1) < n2) set = 1; else if ((i -= n2) < n1) set = 2; else { i -= n1; set = 3; }
        const int lane = i & 63;
        if (set == 0) dst[idx] = gate_chunk(jb.w1, R, C, C, 1, i / (64 * CU), (i >> 6) % CU, lane);          // W1 [R][C]
        else if (set == 1) dst[idx] = gate_chunk(jb.w2, C, R, R, 1, i / (64 * RU), (i >> 6) % RU, lane);     // W2 [C][R]
        else if (set == 2) dst[idx] = gate_chunk(jb.w2, R, C, 1, R, i / (64 * CU), (i >> 6) % CU, lane);     // W2^T: (r, c) = W2[c][r]
        else dst[idx] = gate_chunk(jb.w1, C, R, 1, C, i / (64 * RU), (i >> 6) % RU, lane);                    // W1^T: (c, r) = W1[r][c]
    }
}

__device__ __forceinline__ void gate_copy_chunks(uint4* __restrict__ dst, const uint4* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
// a zero-padded constant row of n valid floats (`fill` where the source is absent)
__device__ __forceinline__ void gate_fill_row(float* __restrict__ dst, const float* __restrict__ src, int n, int padded, float fill = 0.f) {
    for (int i = threadIdx.x; i < padded; i += blockDim.x) dst[i] = i < n ? (src != nullptr ? src[i] : fill) : 0.f;
}
__device__ __forceinline__ float4 gate_widen(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ bf16x8_t gate_frag(float4 a, float4 b) {
    const uint4 u = make_uint4(gate_pack2(a.x, a.y), gate_pack2(a.z, a.w), gate_pack2(b.x, b.y), gate_pack2(b.z, b.w));
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ float4 gate_fma4(float4 y, float4 s, float4 b) {
    return make_float4(fmaf(y.x, s.x, b.x), fmaf(y.y, s.y, b.y), fmaf(y.z, s.z, b.z), fmaf(y.w, s.w, b.w));
}
__device__ __forceinline__ float4 gate_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 gate_f(gate_f4 a) { return make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ gate_f4 gate_zero() { return gate_f4{0.f, 0.f, 0.f, 0.f}; }

// D[16T .. 16T+15][px] += A-chunks[T][u] . B[u] for all tiles T, k-steps u (NT tiles, NU k-steps; bfrag(u) builds B of k-step u)
template <int NT, int NU, typename BF>
__device__ __forceinline__ void gate_product(gate_f4 (&acc)[NT], const uint4* __restrict__ chunks_at_lane, BF&& bfrag) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = gate_zero();
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const bf16x8_t b = bfrag(u);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, chunks_at_lane[(t * NU + u) * 64]), b, acc[t], 0, 0, 0);
    }
}

// per-channel sums held per lane (rows 16 t + 4 rg + i of its pixel) -> one partial row [2][NV] of the workgroup: over the 16 pixel lanes
// (xor butterflies stay inside the 16-lane group), then the waves in order.  red: [waves][2][NP] floats of LDS.
template <int NS, int NP, int NV, int NW>
__device__ __forceinline__ void gate_write_sums(const gate_f4 (&ssum)[NS], const gate_f4 (&qsum)[NS], float* __restrict__ red, float* __restrict__ parts) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = ssum[t][i], q = qsum[t][i];
#pragma unroll
            for (int k = 1; k < 16; k <<= 1) { s += __shfl_xor(s, k); q += __shfl_xor(q, k); }
            if (px == 0) { red[(wave * 2 + 0) * NP + 16 * t + 4 * rg + i] = s; red[(wave * 2 + 1) * NP + 16 * t + 4 * rg + i] = q; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < NV; c += blockDim.x) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { s += red[(w * 2 + 0) * NP + c]; q += red[(w * 2 + 1) * NP + c]; }
        parts[(int64_t)blockIdx.x * 2 * NV + c] = s;
        parts[(int64_t)blockIdx.x * 2 * NV + NV + c] = q;
    }
}

struct GateFwdArgs {
    const bf16_t* y3; const float* s3; const float* b3;           // the project unit's raw output and BN3 scale / shift
    const uint4* wq;                                              // chunk sets (mny_gate_cut_batch_bf16)
    const float* sc1; const float* sh1;                           // BN1 scale / shift (passes 2, 3)
    const float* sc2; const float* sh2;                           // BN2 scale / shift (pass 3)
    const bf16_t* add_x; const float* add_scale; const float* add_shift; int add_act;      // pass 3: optional residual operand (a view)
    bf16_t* out;                                                  // pass 3
    float* parts;                                                 // passes 1, 2: partial rows [gridDim.x][2][R | C]
    int64_t M;
};

// PASS 1: statistics of W1 t;  PASS 2: statistics of W2 h;  PASS 3: out = t * gate (+ residual)
template <int C, int R, int PASS>
__global__ __launch_bounds__(256) void gate_fwd_kernel(GateFwdArgs p) {
    typedef GateDim<C, R> D;
    constexpr int CT = D::CT, CU = D::CU, CP = D::CP, RT = D::RT, RU = D::RU, RP = D::RP;
    __shared__ uint4 wa1[D::N1];
    __shared__ uint4 wa2[PASS >= 2 ? D::N2 : 1];
    __shared__ __attribute__((aligned(16))) float cs3[CP], cb3[CP], c1s[RP], c1b[RP], c2s[PASS == 3 ? CP : 4], c2b[PASS == 3 ? CP : 4];
    __shared__ __attribute__((aligned(16))) float cas[PASS == 3 ? CP : 4], cab[PASS == 3 ? CP : 4];
    __shared__ float red[PASS == 3 ? 1 : 4 * 2 * (PASS == 1 ? RP : CP)];

    gate_copy_chunks(wa1, p.wq + D::O1, D::N1);
    if (PASS >= 2) gate_copy_chunks(wa2, p.wq + D::O2, D::N2);
    gate_fill_row(cs3, p.s3, C, CP); gate_fill_row(cb3, p.b3, C, CP);
    if (PASS >= 2) { gate_fill_row(c1s, p.sc1, R, RP); gate_fill_row(c1b, p.sh1, R, RP); }
    if (PASS == 3) {
        gate_fill_row(c2s, p.sc2, C, CP); gate_fill_row(c2b, p.sh2, C, CP);
        gate_fill_row(cas, p.add_scale, C, CP, 1.f); gate_fill_row(cab, p.add_shift, C, CP, 0.f);
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    constexpr int NS = PASS == 1 ? RT : (PASS == 2 ? CT : 1);
    gate_f4 ssum[NS], qsum[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { ssum[i] = gate_zero(); qsum[i] = gate_zero(); }

    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        // opaque per iteration: the LDS reads of weight chunks and constants stay IN the loop (hoisted they are 140+ VGPRs live across it)
        int lo = lane, cq = 4 * rg;
        asm volatile("" : "+v"(lo), "+v"(cq));
        const int64_t m = tile * 16 + px;
        const bool valid = m < p.M;
        const float vm = valid ? 1.f : 0.f;
        const bf16_t* yrow = p.y3 + (valid ? m : 0) * C;
        uint2 raw[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) raw[t] = (16 * t + 4 * rg < C) ? *reinterpret_cast<const uint2*>(yrow + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        auto tval = [&](int t) { return gate_fma4(gate_widen(raw[t]), gate_ld4(cs3 + 16 * t + cq), gate_ld4(cb3 + 16 * t + cq)); };
        // product 1: (W1 t)[r][px]
        gate_f4 acc1[RT];
        gate_product<RT, CU>(acc1, wa1 + lo, [&](int u) { return gate_frag(tval(2 * u), 2 * u + 1 < CT ? tval(2 * u + 1 < CT ? 2 * u + 1 : 0) : f4zero()); });
        if (PASS == 1) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const gate_f4 v = acc1[t] * vm;
                ssum[t] += v;
                qsum[t] += v * acc1[t];
            }
            continue;
        }
        // h = relu(BN1(.)), product 2: (W2 h)[c][px]
        auto hval = [&](int t) {
            const float4 s = gate_ld4(c1s + 16 * t + cq), b = gate_ld4(c1b + 16 * t + cq);
            const gate_f4 a = acc1[t];
            return make_float4(fmaxf(fmaf(a[0], s.x, b.x), 0.f), fmaxf(fmaf(a[1], s.y, b.y), 0.f), fmaxf(fmaf(a[2], s.z, b.z), 0.f), fmaxf(fmaf(a[3], s.w, b.w), 0.f));
        };
        gate_f4 acc2[CT];
        gate_product<CT, RU>(acc2, wa2 + lo, [&](int u) { return gate_frag(hval(2 * u), 2 * u + 1 < RT ? hval(2 * u + 1 < RT ? 2 * u + 1 : 0) : f4zero()); });
        if (PASS == 2) {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const gate_f4 v = acc2[t] * vm;
                ssum[t] += v;
                qsum[t] += v * acc2[t];
            }
            continue;
        }
        // PASS 3
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int c0 = 16 * t + 4 * rg;
            if (c0 >= C || !valid) continue;
            const int cl = 16 * t + cq;
            const float4 s2 = gate_ld4(c2s + cl), b2 = gate_ld4(c2b + cl);
            const float4 tt = tval(t);
            const gate_f4 a = acc2[t];
            float4 o;
            o.x = tt.x * (__builtin_amdgcn_fmed3f(fmaf(a[0], s2.x, b2.x) + 3.f, 0.f, 6.f) * (1.f / 6.f));
            o.y = tt.y * (__builtin_amdgcn_fmed3f(fmaf(a[1], s2.y, b2.y) + 3.f, 0.f, 6.f) * (1.f / 6.f));
            o.z = tt.z * (__builtin_amdgcn_fmed3f(fmaf(a[2], s2.z, b2.z) + 3.f, 0.f, 6.f) * (1.f / 6.f));
            o.w = tt.w * (__builtin_amdgcn_fmed3f(fmaf(a[3], s2.w, b2.w) + 3.f, 0.f, 6.f) * (1.f / 6.f));
            if (p.add_x != nullptr) {
                const float4 av = ld4(p.add_x + m * C + c0);
                const float4 as = gate_ld4(cas + cl), ab = gate_ld4(cab + cl);
                o.x += act_fwd(fmaf(av.x, as.x, ab.x), p.add_act); o.y += act_fwd(fmaf(av.y, as.y, ab.y), p.add_act);
                o.z += act_fwd(fmaf(av.z, as.z, ab.z), p.add_act); o.w += act_fwd(fmaf(av.w, as.w, ab.w), p.add_act);
            }
            st4(p.out + m * C + c0, o);
        }
    }
    if (PASS == 1) gate_write_sums<NS, RP, R, 4>(ssum, qsum, red, p.parts);
    if (PASS == 2) gate_write_sums<NS, CP, C, 4>(ssum, qsum, red, p.parts);
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
// Given dO = dL/d(out) [M,C] (the residual operand's gradient is dO itself: the caller aliases it):
//   dgate = dO * t;  dz2 = dgate * hsig'(z2), z2 = BN2(g), g = W2 hb;          PASS 1: sum dz2, sum dz2 * ghat  -> mny_bn_bwd_finalize -> coef2
//   dg = ca2 dz2 + cb2 g + cc2;  dh = W2^T dg;  dz1 = dh * (z1 > 0);           PASS 2: sum dz1, sum dz1 * hhat  -> mny_bn_bwd_finalize -> coef1;  dW2 += dg hb^T
//   dhr = ca1 dz1 + cb1 hr + cc1;  dt = dO * gate + W1^T dhr                   PASS 3: dt (+ the project unit's BN-backward sums), dW1 += dhr tb^T
// (hb, tb: the bf16 operand values of the forward products; dg, dhr enter their products as bf16 operands like every stored gradient.)
struct GateBwdArgs {
    const bf16_t* y3; const float* s3; const float* b3; const bf16_t* dout;
    const uint4* wq;
    const float* sc1; const float* sh1; const float* mean1; const float* invstd1;
    const float* sc2; const float* sh2; const float* mean2; const float* invstd2;
    const float* coef2; const float* coef1;                       // [3][C] (passes 2, 3), [3][R] (pass 3)
    const float* mean3; const float* invstd3;                     // pass 3 with red3: the project unit's statistics
    float* parts;                                                 // pass 1: [grid][2][C];  pass 2: [grid][2][R]
    float* dwp;                                                   // pass 2: dW2 partials [grid][C][R];  pass 3: dW1 partials [grid][R][C]
    bf16_t* dt; float* red3;                                      // pass 3: gradient wrt the project unit's BN output; its BN-backward sums [grid][2][C] (or NULL)
    int64_t M;
};

constexpr int kGateBW = 8;                  // waves per workgroup of the backward passes
constexpr int kGatePitch = 16 * kGateBW + 8; // bf16 elements per row of the transposed operand tiles (272 B: rows 4 banks apart -> conflict-free b128 reads)

template <int C, int R, int PASS>
struct GateBwdLds {
    typedef GateDim<C, R> D;
    static constexpr int NCONST_C = PASS == 1 ? 6 : (PASS == 2 ? 7 : 9), NCONST_R = PASS == 1 ? 2 : (PASS == 2 ? 4 : 5);
    static constexpr size_t chunks = (size_t)(D::N1 + D::N2 + (PASS >= 2 ? D::N1 : 0) + (PASS == 3 ? D::N2 : 0)) * 16;
    static constexpr size_t consts = (size_t)(NCONST_C * D::CP + NCONST_R * D::RP) * 4;
    static constexpr size_t xch = PASS == 1 ? 0 : (size_t)(16 * D::CT + 16 * D::RT) * kGatePitch * 2;
    static constexpr size_t red = (size_t)kGateBW * 2 * (PASS == 2 ? D::RP : D::CP) * 4;
    static constexpr size_t total = chunks + consts + xch + red;
};

template <int C, int R, int PASS, bool RED3>
__global__ __launch_bounds__(64 * kGateBW) void gate_bwd_kernel(GateBwdArgs p) {
    typedef GateDim<C, R> D;
    typedef GateBwdLds<C, R, PASS> L;
    constexpr int CT = D::CT, CU = D::CU, CP = D::CP, RT = D::RT, RU = D::RU, RP = D::RP;
    extern __shared__ __attribute__((aligned(16))) unsigned char gate_lds[];
    uint4* wa1 = reinterpret_cast<uint4*>(gate_lds);
    uint4* wa2 = wa1 + D::N1;
    uint4* wa3 = wa2 + D::N2;                                     // PASS >= 2
    uint4* wa4 = wa3 + D::N1;                                     // PASS == 3
    float* cst = reinterpret_cast<float*>(gate_lds + L::chunks);
    // constant rows over the wide side: 0 s3, 1 b3, 2 sc2, 3 sh2, 4 mean2 | ca2, 5 invstd2 | cb2, 6 cc2, 7 mean3, 8 invstd3;  hidden side: 0 sc1, 1 sh1, 2 mean1 | ca1, 3 invstd1 | cb1, 4 cc1
    float* cr = cst + L::NCONST_C * CP;
    unsigned short* xa = reinterpret_cast<unsigned short*>(gate_lds + L::chunks + L::consts);        // PASS 2: dg^T [16 CT][pitch];  PASS 3: tb^T
    unsigned short* xb = xa + 16 * CT * kGatePitch;                                                  // PASS 2: hb^T [16 RT][pitch];  PASS 3: dhr^T
    float* red = reinterpret_cast<float*>(gate_lds + L::chunks + L::consts + L::xch);

    gate_copy_chunks(wa1, p.wq + D::O1, D::N1);
    gate_copy_chunks(wa2, p.wq + D::O2, D::N2);
    if (PASS >= 2) gate_copy_chunks(wa3, p.wq + D::O3, D::N1);
    if (PASS == 3) gate_copy_chunks(wa4, p.wq + D::O4, D::N2);
    gate_fill_row(cst + 0 * CP, p.s3, C, CP); gate_fill_row(cst + 1 * CP, p.b3, C, CP);
    gate_fill_row(cst + 2 * CP, p.sc2, C, CP); gate_fill_row(cst + 3 * CP, p.sh2, C, CP);
    gate_fill_row(cr + 0 * RP, p.sc1, R, RP); gate_fill_row(cr + 1 * RP, p.sh1, R, RP);
    if (PASS == 1) { gate_fill_row(cst + 4 * CP, p.mean2, C, CP); gate_fill_row(cst + 5 * CP, p.invstd2, C, CP); }
    if (PASS >= 2) { gate_fill_row(cst + 4 * CP, p.coef2, C, CP); gate_fill_row(cst + 5 * CP, p.coef2 + C, C, CP); gate_fill_row(cst + 6 * CP, p.coef2 + 2 * C, C, CP); }
    if (PASS == 2) { gate_fill_row(cr + 2 * RP, p.mean1, R, RP); gate_fill_row(cr + 3 * RP, p.invstd1, R, RP); }
    if (PASS == 3) {
        gate_fill_row(cr + 2 * RP, p.coef1, R, RP); gate_fill_row(cr + 3 * RP, p.coef1 + R, R, RP); gate_fill_row(cr + 4 * RP, p.coef1 + 2 * R, R, RP);
        if (RED3) { gate_fill_row(cst + 7 * CP, p.mean3, C, CP); gate_fill_row(cst + 8 * CP, p.invstd3, C, CP); }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    const int64_t niter = (ntiles + kGateBW - 1) / kGateBW;      // workgroup iterations of 8 wave tiles = 128 pixels (uniform: barriers inside)
    constexpr int NS = PASS == 2 ? RT : ((PASS == 1 || RED3) ? CT : 1);
    gate_f4 ssum[NS], qsum[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { ssum[i] = gate_zero(); qsum[i] = gate_zero(); }
    // weight-gradient tiles of this wave: q = wave + 8 j over the (row tile, column tile) grid — PASS 2: dW2 (CT x RT), PASS 3: dW1 (RT x CT)
    constexpr int NTILE = CT * RT, JT = PASS == 1 ? 1 : (NTILE + kGateBW - 1) / kGateBW;
    gate_f4 wacc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) wacc[j] = gate_zero();

    for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
        int lo = lane, cq = 4 * rg;
        asm volatile("" : "+v"(lo), "+v"(cq));
        const int64_t m = (it * kGateBW + wave) * 16 + px;
        const bool valid = m < p.M;
        const float vm = valid ? 1.f : 0.f;
        const int64_t mrow = (valid ? m : 0) * C;
        uint2 raw[CT], rdo[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const bool cok = 16 * t + 4 * rg < C;
            raw[t] = cok ? *reinterpret_cast<const uint2*>(p.y3 + mrow + 16 * t + 4 * rg) : make_uint2(0u, 0u);
            rdo[t] = cok ? *reinterpret_cast<const uint2*>(p.dout + mrow + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        }
        auto tval = [&](int t) { return gate_fma4(gate_widen(raw[t]), gate_ld4(cst + 0 * CP + 16 * t + cq), gate_ld4(cst + 1 * CP + 16 * t + cq)); };
        gate_f4 acc1[RT];                                         // hr = W1 tb
        gate_product<RT, CU>(acc1, wa1 + lo, [&](int u) { return gate_frag(tval(2 * u), 2 * u + 1 < CT ? tval(2 * u + 1 < CT ? 2 * u + 1 : 0) : f4zero()); });
        auto z1val = [&](int t) { return gate_fma4(gate_f(acc1[t]), gate_ld4(cr + 0 * RP + 16 * t + cq), gate_ld4(cr + 1 * RP + 16 * t + cq)); };
        auto hval = [&](int t) { const float4 z = z1val(t); return make_float4(fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f)); };
        gate_f4 acc2[CT];                                         // g = W2 hb
        gate_product<CT, RU>(acc2, wa2 + lo, [&](int u) { return gate_frag(hval(2 * u), 2 * u + 1 < RT ? hval(2 * u + 1 < RT ? 2 * u + 1 : 0) : f4zero()); });
        // dz2 (PASS 1: summed) / dg (PASS >= 2: replaces g in acc2), dO * gate (PASS 3: kept in rdo's place as fp32 -> dog)
        float4 dog[PASS == 3 ? CT : 1];
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int cl = 16 * t + cq;
            const float4 tt = tval(t), dO = gate_widen(rdo[t]);
            const float4 z2 = gate_fma4(gate_f(acc2[t]), gate_ld4(cst + 2 * CP + cl), gate_ld4(cst + 3 * CP + cl));
            const float zz[4] = {z2.x, z2.y, z2.z, z2.w}, tv[4] = {tt.x, tt.y, tt.z, tt.w}, dv[4] = {dO.x, dO.y, dO.z, dO.w};
            float dz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dz[i] = dv[i] * tv[i] * ((zz[i] > -3.f && zz[i] < 3.f) ? (1.f / 6.f) : 0.f) * vm;
            if (PASS == 1) {
                const float4 mu = gate_ld4(cst + 4 * CP + cl), is = gate_ld4(cst + 5 * CP + cl);
                const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { ssum[t][i] += dz[i]; qsum[t][i] = fmaf(dz[i], (acc2[t][i] - mv[i]) * iv[i], qsum[t][i]); }
            } else {
                const float4 ca = gate_ld4(cst + 4 * CP + cl), cb = gate_ld4(cst + 5 * CP + cl), cc = gate_ld4(cst + 6 * CP + cl);
                const float av[4] = {ca.x, ca.y, ca.z, ca.w}, bv[4] = {cb.x, cb.y, cb.z, cb.w}, cv[4] = {cc.x, cc.y, cc.z, cc.w};
                if (PASS == 3) {
                    float gt[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gt[i] = dv[i] * (__builtin_amdgcn_fmed3f(zz[i] + 3.f, 0.f, 6.f) * (1.f / 6.f));
                    dog[PASS == 3 ? t : 0] = make_float4(gt[0], gt[1], gt[2], gt[3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc2[t][i] = fmaf(av[i], dz[i], fmaf(bv[i], acc2[t][i], cv[i])) * vm;      // dg (0 on pixels past M and, through zero constants, on padding channels)
            }
        }
        if (PASS == 1) continue;
        // dh = W2^T dg (k = wide channels: acc2's tiles ARE the B fragments), dz1 = dh * relu'(z1)
        gate_f4 acc3[RT];
        gate_product<RT, CU>(acc3, wa3 + lo, [&](int u) { return gate_frag(gate_f(acc2[2 * u]), 2 * u + 1 < CT ? gate_f(acc2[2 * u + 1 < CT ? 2 * u + 1 : 0]) : f4zero()); });
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float4 z1 = z1val(t);
            const float zz[4] = {z1.x, z1.y, z1.z, z1.w};
            const int rl = 16 * t + cq;
            if (PASS == 2) {
                const float4 mu = gate_ld4(cr + 2 * RP + rl), is = gate_ld4(cr + 3 * RP + rl);
                const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dz1 = zz[i] > 0.f ? acc3[t][i] : 0.f;            // (dg is already 0 on masked pixels -> so is dh)
                    ssum[t][i] += dz1;
                    qsum[t][i] = fmaf(dz1, (acc1[t][i] - mv[i]) * iv[i], qsum[t][i]);
                }
            } else {
                const float4 ca = gate_ld4(cr + 2 * RP + rl), cb = gate_ld4(cr + 3 * RP + rl), cc = gate_ld4(cr + 4 * RP + rl);
                const float av[4] = {ca.x, ca.y, ca.z, ca.w}, bv[4] = {cb.x, cb.y, cb.z, cb.w}, cv[4] = {cc.x, cc.y, cc.z, cc.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) acc3[t][i] = fmaf(av[i], zz[i] > 0.f ? acc3[t][i] : 0.f, fmaf(bv[i], acc1[t][i], cv[i])) * vm;     // dhr
            }
        }
        // park the two operands of the weight gradient transposed: [channel][128 pixels] bf16, this wave's 16 columns
        const int col = 16 * wave + px;
        if (PASS == 2) {
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(acc2[t][i], 0.f) & 0xffffu);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const float4 hv = hval(t);
                const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(hh[i] * vm, 0.f) & 0xffffu);
            }
        } else {
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const float4 tt = tval(t);
                const float tv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) xa[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(tv[i] * vm, 0.f) & 0xffffu);
            }
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(acc3[t][i], 0.f) & 0xffffu);
        }
        if (PASS == 3) {
            // dt = dO * gate + W1^T dhr (k = hidden channels: acc3's tiles are the B fragments)
            gate_f4 acc4[CT];
            gate_product<CT, RU>(acc4, wa4 + lo, [&](int u) { return gate_frag(gate_f(acc3[2 * u]), 2 * u + 1 < RT ? gate_f(acc3[2 * u + 1 < RT ? 2 * u + 1 : 0]) : f4zero()); });
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int c0 = 16 * t + 4 * rg;
                const float4 g = dog[PASS == 3 ? t : 0];
                const float4 o = make_float4(g.x + acc4[t][0], g.y + acc4[t][1], g.z + acc4[t][2], g.w + acc4[t][3]);
                if (c0 < C && valid) st4(p.dt + m * C + c0, o);
                if (RED3) {                                        // the project unit's BN-backward sums over the STORED gradient (its activation is the identity)
                    const int cl = 16 * t + cq;
                    const float4 os = stored4<bf16_t>(o), y = gate_widen(raw[t]), mu = gate_ld4(cst + 7 * CP + cl), is = gate_ld4(cst + 8 * CP + cl);
                    const float ov[4] = {os.x, os.y, os.z, os.w}, yv[4] = {y.x, y.y, y.z, y.w}, mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float d = ov[i] * vm; ssum[t][i] += d; qsum[t][i] = fmaf(d, (yv[i] - mv[i]) * iv[i], qsum[t][i]); }
                }
            }
        }
        __syncthreads();
        // this wave's tiles of the weight gradient over the 128 parked pixels: A = rows of xa (PASS 2) / xb (PASS 3), B = the other
        {
            const unsigned short* ra = PASS == 2 ? xa : xb;       // rows of dW
            const unsigned short* cb_ = PASS == 2 ? xb : xa;      // columns of dW
            constexpr int NCOLT = PASS == 2 ? RT : CT;
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int q = wave + kGateBW * j;
                if (q < NTILE) {
                    const int tr = q / NCOLT, tc = q % NCOLT;
                    const unsigned short* pa = ra + (16 * tr + px) * kGatePitch + 8 * rg;
                    const unsigned short* pb = cb_ + (16 * tc + px) * kGatePitch + 8 * rg;
#pragma unroll
                    for (int ks = 0; ks < kGateBW / 2; ++ks) {
                        const uint4 a = *reinterpret_cast<const uint4*>(pa + 32 * ks), b = *reinterpret_cast<const uint4*>(pb + 32 * ks);
                        wacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), wacc[j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (PASS == 1) gate_write_sums<NS, CP, C, kGateBW>(ssum, qsum, red, p.parts);
    if (PASS == 2) gate_write_sums<NS, RP, R, kGateBW>(ssum, qsum, red, p.parts);
    if (PASS == 3 && RED3) gate_write_sums<NS, CP, C, kGateBW>(ssum, qsum, red, p.red3);
    if (PASS >= 2) {
        constexpr int NROW = PASS == 2 ? C : R, NCOL = PASS == 2 ? R : C, NCOLT = PASS == 2 ? RT : CT;
        float* dst = p.dwp + (int64_t)blockIdx.x * NROW * NCOL;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int q = wave + kGateBW * j;
            if (q < NTILE) {
                const int tr = q / NCOLT, tc = q % NCOLT;
                const int c = 16 * tc + px;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * tr + 4 * rg + i;
                    if (r < NROW && c < NCOL) dst[r * NCOL + c] = wacc[j][i];
                }
            }
        }
    }
}

// ---- thin project unit, backward as one pass, bf16 storage (the MobileNetV3 counterpart of csrc/pjbwd.hip) --------------------------------
// Autograd of the linear bottleneck conv3 + bn3 (models/mobilenetv3.py:57-58,69): given G = dL/d(BN output) [M,No], the unit's raw output
// Y [M,No], its BN-backward coefficients coef[3][No], and the raw output D [M,Ki] of the depthwise unit in front (view d_scale / d_shift /
// d_act, statistics d_mean / d_invstd), consumed only here:
//     dY = ca G + cb Y + cc;   gd = W^T dY  (the gradient wrt the ACTIVATED d, stored);   the depthwise unit's BN-backward sums over
//     (stored gd * act'(z_d), d);   dW[no][ki] = sum_px dY[no][px] a_d[ki][px],  a_d = bf16(act(z_d)) — the forward GEMM's A operand.
// One launch instead of mny_bn_bwd_apply + mny_pw_dgrad_bnred + mny_pw_wgrad: G, Y, D are read once, dY never reaches HBM.
// Same machinery as the gate: a wave owns 16 pixels x all channels, W^T is the A operand (cut once per workgroup), dY in the accumulator
// layout IS the B fragment; the weight gradient contracts over the 128 pixels the 8 waves of the workgroup park transposed in LDS.
struct Pj16Args {
    const bf16_t* g; const bf16_t* y; const float* coef;
    const bf16_t* d; const float* d_scale; const float* d_shift; const float* d_mean; const float* d_invstd; int d_act;
    const float* w;                                               // [No][Ki] fp32 master
    bf16_t* gd; float* dw_parts; float* red;
    int64_t M;
};
template <int KI, int NO>
struct Pj16Lds {
    static constexpr int CT = (KI + 15) / 16, CP = ((CT + 1) / 2) * 32, NT = (NO + 15) / 16, NU = (NT + 1) / 2, NP = NU * 32;
    static constexpr size_t chunks = (size_t)CT * NU * 64 * 16;
    static constexpr size_t consts = (size_t)(4 * CP + 3 * NP) * 4;
    static constexpr size_t xch = (size_t)(16 * CT + 16 * NT) * kGatePitch * 2;
    static constexpr size_t red = (size_t)kGateBW * 2 * CP * 4;
    // the 16 x KI outputs of a wave's pixel tile are one contiguous run of gd: parked here ([pixel][CP + 8] bf16) and stored as whole 16-byte lanes
    // (straight from the accumulator layout every store instruction writes 16 separate 32-byte pieces, non-temporal at that)
    static constexpr bool flat = CT >= 2;
    static constexpr int spitch = CP + 8;
    static constexpr size_t stage = flat ? (size_t)kGateBW * 16 * spitch * 2 : 0;
    static constexpr size_t total = chunks + consts + xch + red + stage;
};

template <int KI, int NO>
__global__ __launch_bounds__(64 * kGateBW) void pj16_bwd_kernel(Pj16Args p) {
    typedef Pj16Lds<KI, NO> L;
    constexpr int CT = L::CT, CP = L::CP, NT = L::NT, NU = L::NU, NP = L::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];
    uint4* wa = reinterpret_cast<uint4*>(pj_lds);                  // W^T chunks: rows ki, k = no
    float* cd = reinterpret_cast<float*>(pj_lds + L::chunks);     // wide side: 0 d_scale, 1 d_shift, 2 d_mean, 3 d_invstd
    float* cn = cd + 4 * CP;                                      // thin side: 0 ca, 1 cb, 2 cc
    unsigned short* xa = reinterpret_cast<unsigned short*>(pj_lds + L::chunks + L::consts);       // a_d^T [16 CT][pitch]
    unsigned short* xb = xa + 16 * CT * kGatePitch;                                               // dY^T  [16 NT][pitch]
    float* red = reinterpret_cast<float*>(pj_lds + L::chunks + L::consts + L::xch);
    unsigned short* stage = reinterpret_cast<unsigned short*>(pj_lds + L::chunks + L::consts + L::xch + L::red);

    for (int idx = threadIdx.x; idx < CT * NU * 64; idx += blockDim.x)
        wa[idx] = gate_chunk(p.w, KI, NO, 1, KI, idx / (64 * NU), (idx >> 6) % NU, idx & 63);     // (ki, no) = w[no * KI + ki]
    gate_fill_row(cd + 0 * CP, p.d_scale, KI, CP); gate_fill_row(cd + 1 * CP, p.d_shift, KI, CP);
    gate_fill_row(cd + 2 * CP, p.d_mean, KI, CP); gate_fill_row(cd + 3 * CP, p.d_invstd, KI, CP);
    gate_fill_row(cn + 0 * NP, p.coef, NO, NP); gate_fill_row(cn + 1 * NP, p.coef + NO, NO, NP); gate_fill_row(cn + 2 * NP, p.coef + 2 * NO, NO, NP);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    const int64_t niter = (ntiles + kGateBW - 1) / kGateBW;
    gate_f4 ssum[CT], qsum[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) { ssum[i] = gate_zero(); qsum[i] = gate_zero(); }
    constexpr int NTILE = NT * CT, JT = (NTILE + kGateBW - 1) / kGateBW;
    gate_f4 wacc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) wacc[j] = gate_zero();

    // the rows of the NEXT workgroup iteration are requested before this one is processed (one workgroup of 8 waves per CU at ~200 VGPRs:
    // without it every iteration is a full memory round trip in front of two barriers)
    uint2 ng[NT], ny[NT], nd_[CT];
    auto request = [&](int64_t it) {
        const int64_t m = (it * kGateBW + wave) * 16 + px;
        const int64_t mr = (it < niter && m < p.M) ? m : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = 16 * t + 4 * rg < NO;
            ng[t] = ok ? *reinterpret_cast<const uint2*>(p.g + mr * NO + 16 * t + 4 * rg) : make_uint2(0u, 0u);
            ny[t] = ok ? *reinterpret_cast<const uint2*>(p.y + mr * NO + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int t = 0; t < CT; ++t)
            nd_[t] = (16 * t + 4 * rg < KI) ? *reinterpret_cast<const uint2*>(p.d + mr * KI + 16 * t + 4 * rg) : make_uint2(0u, 0u);
    };
    request(blockIdx.x);
    for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
        int lo = lane, cq = 4 * rg;
        asm volatile("" : "+v"(lo), "+v"(cq));
        const int64_t m = (it * kGateBW + wave) * 16 + px;
        const bool valid = m < p.M;
        const float vm = valid ? 1.f : 0.f;
        uint2 rg_[NT], ry[NT], rd[CT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { rg_[t] = ng[t]; ry[t] = ny[t]; }
#pragma unroll
        for (int t = 0; t < CT; ++t) rd[t] = nd_[t];
        request(it + gridDim.x);
        // dY = ca G + cb Y + cc (0 on pixels past M; zero constants on padding channels)
        auto dyval = [&](int t) {
            const float4 G = gate_widen(rg_[t]), Y = gate_widen(ry[t]);
            const float4 ca = gate_ld4(cn + 0 * NP + 16 * t + cq), cb = gate_ld4(cn + 1 * NP + 16 * t + cq), cc = gate_ld4(cn + 2 * NP + 16 * t + cq);
            return make_float4(fmaf(ca.x, G.x, fmaf(cb.x, Y.x, cc.x)) * vm, fmaf(ca.y, G.y, fmaf(cb.y, Y.y, cc.y)) * vm,
                               fmaf(ca.z, G.z, fmaf(cb.z, Y.z, cc.z)) * vm, fmaf(ca.w, G.w, fmaf(cb.w, Y.w, cc.w)) * vm);
        };
        gate_f4 acc[CT];                                          // gd = W^T dY
        gate_product<CT, NU>(acc, wa + lo, [&](int u) { return gate_frag(dyval(2 * u), 2 * u + 1 < NT ? dyval(2 * u + 1 < NT ? 2 * u + 1 : 0) : f4zero()); });
        const int col = 16 * wave + px;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int c0 = 16 * t + 4 * rg, cl = 16 * t + cq;
            const float4 dv = gate_widen(rd[t]);
            const float4 z = gate_fma4(dv, gate_ld4(cd + 0 * CP + cl), gate_ld4(cd + 1 * CP + cl));
            const float4 mu = gate_ld4(cd + 2 * CP + cl), is = gate_ld4(cd + 3 * CP + cl);
            const float4 o = gate_f(acc[t]);
            if (L::flat) *reinterpret_cast<uint2*>(stage + (wave * 16 + px) * L::spitch + c0) = make_uint2(gate_pack2(o.x, o.y), gate_pack2(o.z, o.w));
            else if (c0 < KI && valid) st4(p.gd + m * KI + c0, o);
            const float4 os = stored4<bf16_t>(o);
            const float zz[4] = {z.x, z.y, z.z, z.w}, ov[4] = {os.x, os.y, os.z, os.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
            const float mv[4] = {mu.x, mu.y, mu.z, mu.w}, iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dz = ov[i] * act_bwd(zz[i], p.d_act) * vm;
                ssum[t][i] += dz;
                qsum[t][i] = fmaf(dz, (dd[i] - mv[i]) * iv[i], qsum[t][i]);
                xa[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(act_fwd(zz[i], p.d_act) * vm, 0.f) & 0xffffu);
            }
        }
        if (L::flat) {                                              // the wave's own rows: LDS is in order per wave
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t m0 = (it * kGateBW + wave) * 16;
            const int64_t left = p.M - m0;
            const int rows = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
            constexpr int cpr = KI / 8;                              // 16-byte pieces per pixel row (KI % 8 == 0)
            const int chunks = rows * cpr;
            uint4* const dst = reinterpret_cast<uint4*>(p.gd + (left > 0 ? m0 : 0) * KI);
#pragma unroll
            for (int j = 0; j < (16 * cpr + 63) / 64; ++j) {
                const int q = lane + 64 * j;
                if (q < chunks) {
                    const int r = q / cpr, cc = q - r * cpr;
                    dst[q] = *reinterpret_cast<const uint4*>(stage + (wave * 16 + r) * L::spitch + cc * 8);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 dy = dyval(t);
            const float dv[4] = {dy.x, dy.y, dy.z, dy.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) xb[(16 * t + 4 * rg + i) * kGatePitch + col] = (unsigned short)(gate_pack2(dv[i], 0.f) & 0xffffu);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JT; ++j) {                             // dW tiles: rows no (xb), columns ki (xa)
            const int q = wave + kGateBW * j;
            if (q < NTILE) {
                const int tr = q / CT, tc = q % CT;
                const unsigned short* pa = xb + (16 * tr + px) * kGatePitch + 8 * rg;
                const unsigned short* pb = xa + (16 * tc + px) * kGatePitch + 8 * rg;
#pragma unroll
                for (int ks = 0; ks < kGateBW / 2; ++ks) {
                    const uint4 a = *reinterpret_cast<const uint4*>(pa + 32 * ks), b = *reinterpret_cast<const uint4*>(pb + 32 * ks);
                    wacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), wacc[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    gate_write_sums<CT, CP, KI, kGateBW>(ssum, qsum, red, p.red);
    float* dst = p.dw_parts + (int64_t)blockIdx.x * NO * KI;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int q = wave + kGateBW * j;
        if (q < NTILE) {
            const int tr = q / CT, tc = q % CT;
            const int c = 16 * tc + px;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * tr + 4 * rg + i;
                if (r < NO && c < KI) dst[r * KI + c] = wacc[j][i];
            }
        }
    }
}

static bool pj16_shape_ok(int64_t M, int Ki, int No) {
    return M >= 16 && ((Ki == 16 && No == 16) || (Ki == 64 && No == 24) || (Ki == 72 && No == 24) || (Ki == 72 && No == 40) || (Ki == 120 && No == 40));
}
static int pj16_grid(int64_t M) {
    const int64_t want = cdiv(cdiv(M, 16), kGateBW);
    return (int)(want < 512 ? want : 512);
}
template <int KI, int NO>
static int pj16_launch(const Pj16Args& a, int grid, hipStream_t st) {
    constexpr size_t lds = Pj16Lds<KI, NO>::total;
    static_assert(lds <= 160 * 1024, "pj16: LDS budget");
    if (lds > 64 * 1024 && !allow_lds((const void*)pj16_bwd_kernel<KI, NO>, lds)) { set_error("pj_bwd_bf16: hipFuncSetAttribute failed"); return MNY_EHIP; }
    hipLaunchKernelGGL((pj16_bwd_kernel<KI, NO>), dim3(grid), dim3(64 * kGateBW), lds, st, a);
    return check_launch("pj16_bwd_kernel");
}

// ---- thin pointwise conv forward / plain data gradient on the same machinery (bf16 storage) -----------------------------------------
// y[M][N] = view(x)[M][K] . W^T for K <= 48 at a large pixel count (MobileNetV3's first blocks at 256x256 / 128x128 / 64x64: K = 16, 24, 40):
// the short-reduction vector-ALU kernel these shapes took runs 1.3-2.5 TB/s (16-40 fma per output element), the LDS-DMA tile kernel has no
// tile for them.  Here a wave owns 16 pixels: the input row is loaded in the accumulator layout (8 bytes per lane and 16-channel tile),
// activated, and IS the B operand; the weights (<= 240 x 48) are cut into A-operand chunks by every workgroup in its prologue (a few KB from
// L2); every 16x16 output tile leaves as one 8-byte store per lane; the BatchNorm statistics (sum, sum of squares of the STORED values) are
// kept per lane and leave as one partial row per workgroup.  The arithmetic is the bf16 GEMM path's: operands rounded to bf16, fp32 accumulate.
// replaces nn.Conv2d(K, N, 1) of the thin expand / project / shortcut convs (models/mobilenetv3.py:49,57,63) and its plain data gradient.
struct PwtArgs {
    const bf16_t* x; const float* xs; const float* xb; int xact;
    const bf16_t* w;                                              // [N][K] bf16 (the GEMM weight shadow)
    const bf16_t* addend; bf16_t* y; float* parts; int64_t M; int K, N;
    // RED (data gradient of a 1x1 conv whose INPUT was the raw output rY of a conv+BN+act unit): that unit's BN-backward sums leave with dx
    const bf16_t* rY; const float* r_scale; const float* r_shift; const float* r_mean; const float* r_invstd; int r_act;
};

// SUMS: 0 none, 1 BatchNorm statistics of the stored output, 2 RED (sum dz, sum dz * yhat with dz = stored(dx) * act'(r_scale * rY + r_shift))
template <int KT, int NT, int XF, int SUMS>
__global__ __launch_bounds__(256) void pwt_fwd_kernel(PwtArgs p) {
    constexpr bool STATS = SUMS != 0;
    constexpr int KU = (KT + 1) / 2, KP = KU * 32, NP = NT * 16;
    __shared__ uint4 wa[NT * KU * 64];
    __shared__ __attribute__((aligned(16))) float cs[KP], cb[KP];
    __shared__ float red[STATS ? 4 * 2 * NP : 1];
    __shared__ __attribute__((aligned(16))) float rc[SUMS == 2 ? 4 * NP : 4];          // RED: scale | shift | mean | invstd of the producer unit, zero-padded rows
    // The 16 x N outputs of a wave's pixel tile are ONE contiguous run of y (row-major, consecutive pixels).  Stored straight from the accumulator
    // layout every store instruction writes 16 separate 32-byte pieces (lane (px, rg) holds 4 channels of its pixel): 16 -> 64 @256x256 (537 MB,
    // past the last-level cache) ran 2.5 TB/s against 3.3 of the vector-ALU kernel, 4.6 TB/s with the detour.  For 2-5 output tiles (N = 24 ... 80) the tile is parked in LDS
    // ([pixel][N + 8] bf16: the row pitch keeps the 16 pixel lanes of a write on different banks) and leaves as whole 16-byte lanes of that run.
    constexpr bool FLAT = NT >= 2 && NT <= 5;      // (N = 16 is already one contiguous 512-byte store per wave: 0.053 vs 0.060 ms; N = 120 stays in the
                                                     // last-level cache at these sizes: 0.040 vs 0.045 ms)
    constexpr int SPITCH = NP + 8;                                   // bf16 elements per staged pixel row
    __shared__ __attribute__((aligned(16))) uint16_t stage[FLAT ? 4 * 16 * SPITCH : 8];
    const int K = p.K, N = p.N;
    for (int i = threadIdx.x; i < NT * KU * 64; i += blockDim.x) {
        const int ln = i & 63, u = (i >> 6) % KU, t = i / (64 * KU);
        const int row = 16 * t + (ln & 15), kg = ln >> 4;
        uint32_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * u + (e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4));
            v[e] = (row < N && k < K) ? (uint32_t)p.w[(int64_t)row * K + k].v : 0u;
        }
        wa[i] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
    }
    gate_fill_row(cs, XF != 0 ? p.xs : nullptr, K, KP, 1.f);
    gate_fill_row(cb, XF != 0 ? p.xb : nullptr, K, KP, 0.f);
    if (SUMS == 2) {
        gate_fill_row(rc, p.r_scale, N, NP); gate_fill_row(rc + NP, p.r_shift, N, NP);
        gate_fill_row(rc + 2 * NP, p.r_mean, N, NP); gate_fill_row(rc + 3 * NP, p.r_invstd, N, NP);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    constexpr int NS = STATS ? NT : 1;
    gate_f4 ssum[NS], qsum[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { ssum[i] = gate_zero(); qsum[i] = gate_zero(); }
    const float xslope = act_slope(p.xact);
    // the NEXT tile's input row is requested before this one is multiplied (a wave has one 512-byte load per tile in flight otherwise:
    // 16 -> 16 @256x256 ran 2.5 TB/s); RED: the producer's raw output of THIS tile is requested before the products, used after them
    auto load_row = [&](int64_t tile, uint2 (&r)[KT]) {
        const int64_t mm = tile * 16 + px;
        const bf16_t* xrow = p.x + (mm < p.M ? mm : 0) * K;
#pragma unroll
        for (int t = 0; t < KT; ++t) r[t] = (16 * t + 4 * rg < K) ? *reinterpret_cast<const uint2*>(xrow + 16 * t + 4 * rg) : make_uint2(0u, 0u);
    };
    const int64_t tstep = (int64_t)gridDim.x * 4;
    uint2 nraw[KT];
    load_row((int64_t)blockIdx.x * 4 + wave, nraw);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += tstep) {
        int lo = lane, cq = 4 * rg;
        asm volatile("" : "+v"(lo), "+v"(cq));               // opaque per iteration: chunk and constant reads stay in the loop
        const int64_t m = tile * 16 + px;
        const bool valid = m < p.M;
        uint2 raw[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) raw[t] = nraw[t];
        load_row(tile + tstep < ntiles ? tile + tstep : tile, nraw);
        uint2 yraw[SUMS == 2 ? NT : 1];
        if (SUMS == 2) {
#pragma unroll
            for (int t = 0; t < NT; ++t) yraw[t] = (valid && 16 * t + 4 * rg < N) ? *reinterpret_cast<const uint2*>(p.rY + m * N + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        }
        auto aval = [&](int t) {
            float4 v = gate_widen(raw[t]);
            if (XF != 0) {
                v = gate_fma4(v, gate_ld4(cs + 16 * t + cq), gate_ld4(cb + 16 * t + cq));
                if (XF == 1) v = make_float4(__builtin_amdgcn_fmed3f(v.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.z, 0.f, 6.f),
                                             __builtin_amdgcn_fmed3f(v.w, 0.f, 6.f));
                else if (XF == 2) v = make_float4(v.x * __builtin_amdgcn_fmed3f(v.x + 3.f, 0.f, 6.f) * (1.f / 6.f), v.y * __builtin_amdgcn_fmed3f(v.y + 3.f, 0.f, 6.f) * (1.f / 6.f),
                                                  v.z * __builtin_amdgcn_fmed3f(v.z + 3.f, 0.f, 6.f) * (1.f / 6.f), v.w * __builtin_amdgcn_fmed3f(v.w + 3.f, 0.f, 6.f) * (1.f / 6.f));
                else v = make_float4(fmaxf(v.x, xslope * v.x), fmaxf(v.y, xslope * v.y), fmaxf(v.z, xslope * v.z), fmaxf(v.w, xslope * v.w));
                if (16 * t + 4 * rg >= K) v = f4zero();        // padded channels: the view of 0 is not 0
            }
            return v;
        };
        gate_f4 acc[NT];
        gate_product<NT, KU>(acc, wa + lo, [&](int u) { return gate_frag(aval(2 * u), 2 * u + 1 < KT ? aval(2 * u + 1 < KT ? 2 * u + 1 : 0) : f4zero()); });
        uint16_t* const stw = stage + (FLAT ? wave * 16 * SPITCH : 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c0 = 16 * t + 4 * rg;
            float4 o = gate_f(acc[t]);
            const bool ok = valid && c0 < N;
            if (p.addend != nullptr && ok) add4(o, ld4(p.addend + m * N + c0));
            if (FLAT) *reinterpret_cast<uint2*>(stw + px * SPITCH + c0) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            else if (ok) st4(p.y + m * N + c0, o);               // (plain stores: non-temporal 8-byte pieces measured 2x slower here)
            if (SUMS == 1) {
                const float vm = ok ? 1.f : 0.f;
                const float4 q = stored4<bf16_t>(o);
                const gate_f4 v = gate_f4{q.x, q.y, q.z, q.w} * vm;
                ssum[t] += v;
                qsum[t] += v * v;
            }
            if (SUMS == 2) {
                const float vm = ok ? 1.f : 0.f;
                const float4 q = stored4<bf16_t>(o);
                const float4 yc = gate_widen(yraw[SUMS == 2 ? t : 0]);
                const float4 rs = gate_ld4(rc + 16 * t + cq), rh = gate_ld4(rc + NP + 16 * t + cq), rm = gate_ld4(rc + 2 * NP + 16 * t + cq),
                             ri = gate_ld4(rc + 3 * NP + 16 * t + cq);
                const gate_f4 dz = gate_f4{q.x * act_bwd(fmaf(yc.x, rs.x, rh.x), p.r_act), q.y * act_bwd(fmaf(yc.y, rs.y, rh.y), p.r_act),
                                           q.z * act_bwd(fmaf(yc.z, rs.z, rh.z), p.r_act), q.w * act_bwd(fmaf(yc.w, rs.w, rh.w), p.r_act)} * vm;
                ssum[t] += dz;
                qsum[t] += dz * gate_f4{(yc.x - rm.x) * ri.x, (yc.y - rm.y) * ri.y, (yc.z - rm.z) * ri.z, (yc.w - rm.w) * ri.w};
            }
        }
        if (FLAT) {
            // the wave's own staging rows: LDS is in order per wave, the wait below is all the synchronisation the hand-over needs
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int64_t m0 = tile * 16;
            const int rows = (int)(p.M - m0 < 16 ? p.M - m0 : 16);
            const int chunks = rows * N / 8;                        // 16-byte pieces of the contiguous run (N % 8 == 0)
            uint4* const dst = reinterpret_cast<uint4*>(p.y + m0 * N);
            const int cpr = N / 8;                                   // pieces per pixel row
#pragma unroll
            for (int j = 0; j < (16 * NP / 8 + 63) / 64; ++j) {
                const int q = lane + 64 * j;
                if (q < chunks) {
                    const int r = q / cpr, cc = q - r * cpr;
                    dst[q] = *reinterpret_cast<const uint4*>(stw + r * SPITCH + cc * 8);
                }
            }
        }
    }
    if (STATS) {                                             // as gate_write_sums, with the row length N known at run time
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = ssum[t][i], q = qsum[t][i];
#pragma unroll
                for (int k = 1; k < 16; k <<= 1) { s += __shfl_xor(s, k); q += __shfl_xor(q, k); }
                if (px == 0) { red[(wave * 2 + 0) * NP + 16 * t + 4 * rg + i] = s; red[(wave * 2 + 1) * NP + 16 * t + 4 * rg + i] = q; }
            }
        __syncthreads();
        for (int c = threadIdx.x; c < N; c += blockDim.x) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[(w * 2 + 0) * NP + c]; q += red[(w * 2 + 1) * NP + c]; }
            p.parts[(int64_t)blockIdx.x * 2 * N + c] = s;
            p.parts[(int64_t)blockIdx.x * 2 * N + N + c] = q;
        }
    }
}

// shapes: K a multiple of 8 up to 48, N a multiple of 8 up to 240, and enough pixels that the vector-ALU / tile kernels are the slower choice
bool pwt_ok(int64_t M, int K, int N, int in_act, bool has_bias) {
    static const bool off = getenv("MNY_NO_PWT") != nullptr && atoi(getenv("MNY_NO_PWT")) != 0;
    if (off || has_bias || in_act == MNY_ACT_HSIGMOID) return false;
    if (K <= 0 || N <= 0 || (K & 7) || (N & 7) || K > 48 || N > 240) return false;
    return M >= 131072;
}
int pwt_parts(int64_t M) {
    const int64_t want = cdiv(cdiv(M, 16), 4);
    return (int)(want < 1024 ? want : 1024);
}
template <int KT, int XF, int SUMS>
static int pwt_launch_n(const PwtArgs& a, int grid, hipStream_t st) {
    const int NT = (a.N + 15) / 16;           // 16-column output tiles; instantiated for 1-5, 8, 15 (N = 16 ... 80, 120, 240): the counts in between run the next
                                               // larger one (its extra tiles are all-zero weight chunks and masked stores)
#define MNY_PWT(N_) hipLaunchKernelGGL((pwt_fwd_kernel<KT, N_, XF, SUMS>), dim3(grid), dim3(256), 0, st, a)
    if (NT <= 1) MNY_PWT(1); else if (NT == 2) MNY_PWT(2); else if (NT == 3) MNY_PWT(3); else if (NT == 4) MNY_PWT(4); else if (NT == 5) MNY_PWT(5);
    else if (NT <= 8) MNY_PWT(8); else MNY_PWT(15);
#undef MNY_PWT
    return check_launch("pwt_fwd_kernel");
}
int pwt_launch(const void* x, const float* xs, const float* xb, int xact, const void* w, const void* addend, void* y, float* stats, int64_t M, int K, int N,
               hipStream_t st, const void* rY, const float* r_scale, const float* r_shift, const float* r_mean, const float* r_invstd, int r_act) {
    PwtArgs a{(const bf16_t*)x, xs, xb, xact, (const bf16_t*)w, (const bf16_t*)addend, (bf16_t*)y, stats, M, K, N,
              (const bf16_t*)rY, r_scale, r_shift, r_mean, r_invstd, r_act};
    const int grid = pwt_parts(M);
    const int xf = (xs == nullptr && xact == MNY_ACT_NONE) ? 0 : (xact == MNY_ACT_RELU6 ? 1 : (xact == MNY_ACT_HSWISH ? 2 : 3));
    const int KT = (K + 15) / 16;
    if (rY != nullptr) {                                     // data gradient + BN-backward sums: no input view
        MNY_REQUIRE(xf == 0 && stats && r_scale && r_shift && r_mean && r_invstd, "pw_dgrad_bnred (wave form): bad arguments");
        if (KT == 1) return pwt_launch_n<1, 0, 2>(a, grid, st);
        if (KT == 2) return pwt_launch_n<2, 0, 2>(a, grid, st);
        return pwt_launch_n<3, 0, 2>(a, grid, st);
    }
#define MNY_PWT_X(KT_, S_) do { switch (xf) { case 0: return pwt_launch_n<KT_, 0, S_>(a, grid, st); case 1: return pwt_launch_n<KT_, 1, S_>(a, grid, st); \
        case 2: return pwt_launch_n<KT_, 2, S_>(a, grid, st); default: return pwt_launch_n<KT_, 3, S_>(a, grid, st); } } while (0)
#define MNY_PWT_K(S_) do { if (KT == 1) MNY_PWT_X(1, S_); else if (KT == 2) MNY_PWT_X(2, S_); else MNY_PWT_X(3, S_); } while (0)
    if (stats) MNY_PWT_K(1); else MNY_PWT_K(0);
#undef MNY_PWT_K
#undef MNY_PWT_X
    return MNY_OK;
}

// ---- thin expand unit backward (conv K -> N + BN + ReLU-family activation), bf16 storage, on the same machinery ------------------------
// mny_pw_bnbwd's algebra (pwgemm.hip): dW = ca o (dz^T B) + cb o (W B^T B) + cc x colsum(B), dX = dz (ca o W) + B Q + bias with dz = G * act'(BN(Y)),
// B = the activated input view.  Its bf16 instantiation keeps the fp32 32x32x2 matrix instructions and reads LDS one element per lane and
// instruction: element-rate-bound at 2.9 TB/s on 16 -> 64 @256x256 (0.41 + 0.24 ms).  Here
//   stage A (pwe_sums_kernel): a wave per 16 pixels loads G, Y (N wide) and X (K thin) in the accumulator layout, forms dz and B per lane, keeps the
//     per-channel sums (s1 = sum dz, s2 = sum dz * yhat, s3 = sum B) per lane, leaves the act' bits of its 4 N channels per tile as one word per lane,
//     and parks dz^T, B^T (bf16) for the workgroup's 128 pixels in LDS like the gate's weight-gradient passes: every wave owns tiles of
//     P1 = dz^T B [N x K] and of the Gram matrix B^T B [K x K] over those pixels (v_mfma_f32_16x16x32_bf16).  One partial row per workgroup in the
//     layout pw_bnbwd_finalize_kernel reads (P1 | Gram | s1 | s2 | s3);
//   stage B (pwe_dgrad_kernel): dX tile = B1 . dz + Q . B + bias, B1 = ca o W^T [K][N] and Q [K][K] cut to bf16 A-operand chunks per workgroup,
//     dz rebuilt from G and the mask word (Y is not read again).
// Operands of the matrix products are rounded to bf16 (what the bf16 GEMM path does with every operand); sums, coefficients, outputs fp32 / as stored.
// autograd of nn.Conv2d(K, N, 1) + BatchNorm2d + ReLU of the first MobileNetV3 blocks (models/mobilenetv3.py:49-51,67).
struct PweArgs {
    const bf16_t* g; const bf16_t* y; const float* sc; const float* sh; int act; const float* mean; const float* invstd;
    const bf16_t* x; const float* xs; const float* xb; int xact;
    float* partial; uint32_t* mask; int64_t M; int K, N;
    const float* B1; const float* Q; const float* bias; const bf16_t* addend; bf16_t* dx;
};
// Stage A: BW waves per workgroup park 16 BW pixels and share the tiles.  Measured (whole unit, same box): 16 -> 64 @256x256 8 waves 0.701 ms, 4 waves
// 0.592 (two workgroups per CU: one's barriers overlap the other's loads), a barrier-free variant (every wave transposes its own 16 pixels, 32-deep
// matrix instruction half empty) 0.628, the fp32-MFMA kernel of pwgemm.hip 0.779; 24 -> 72 @128x128: 8 waves 0.220, 4 waves 0.272, pwgemm.hip 0.256.
template <int KT, int NT, int BW>
struct PweLds {
    static constexpr int kPweBW = BW, kPwePitch = 16 * BW + 8;
    static constexpr int KP = KT * 16, NP = NT * 16;
    static constexpr size_t consts = (size_t)(4 * NP + 2 * KP) * 4;
    static constexpr size_t xch = (size_t)(16 * KT + 16 * NT) * kPwePitch * 2;
    static constexpr size_t red = (size_t)kPweBW * (2 * NP + KP) * 4;
    static constexpr size_t total = consts + xch + red;
};

template <int KT, int NT, int BW>
__global__ __launch_bounds__(64 * BW) void pwe_sums_kernel(PweArgs p) {
    typedef PweLds<KT, NT, BW> L;
    constexpr int kPweBW = BW, kPwePitch = L::kPwePitch;
    constexpr int KP = L::KP, NP = L::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char pe_lds[];
    float* cw = reinterpret_cast<float*>(pe_lds);                  // wide side: 0 scale, 1 shift, 2 mean, 3 invstd
    float* ck = cw + 4 * NP;                                       // thin side: 0 in_scale, 1 in_shift
    unsigned short* xa = reinterpret_cast<unsigned short*>(pe_lds + L::consts);          // B^T  [16 KT][pitch]
    unsigned short* xb = xa + 16 * KT * kPwePitch;                                      // dz^T [16 NT][pitch]
    float* red = reinterpret_cast<float*>(pe_lds + L::consts + L::xch);
    const int K = p.K, N = p.N;
    gate_fill_row(cw, p.sc, N, NP); gate_fill_row(cw + NP, p.sh, N, NP); gate_fill_row(cw + 2 * NP, p.mean, N, NP); gate_fill_row(cw + 3 * NP, p.invstd, N, NP);
    gate_fill_row(ck, p.xs, K, KP, 1.f); gate_fill_row(ck + KP, p.xb, K, KP, 0.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    const int64_t niter = (ntiles + kPweBW - 1) / kPweBW;
    const float aslope = act_slope(p.act), ahi = act_hi(p.act), xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    const bool has_xf = p.xs != nullptr || p.xact != MNY_ACT_NONE;
    gate_f4 s1[NT], s2[NT], s3[KT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { s1[i] = gate_zero(); s2[i] = gate_zero(); }
#pragma unroll
    for (int i = 0; i < KT; ++i) s3[i] = gate_zero();
    constexpr int NTILE = NT * KT + KT * KT, JT = (NTILE + kPweBW - 1) / kPweBW;       // P1 tiles, then Gram tiles
    gate_f4 wacc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) wacc[j] = gate_zero();
    uint2 ng[NT], ny[NT], nx[KT];
    auto request = [&](int64_t it) {
        const int64_t m = (it * kPweBW + wave) * 16 + px;
        const int64_t mr = (it < niter && m < p.M) ? m : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool ok = 16 * t + 4 * rg < N;
            ng[t] = ok ? *reinterpret_cast<const uint2*>(p.g + mr * N + 16 * t + 4 * rg) : make_uint2(0u, 0u);
            ny[t] = ok ? *reinterpret_cast<const uint2*>(p.y + mr * N + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) nx[t] = (16 * t + 4 * rg < K) ? *reinterpret_cast<const uint2*>(p.x + mr * K + 16 * t + 4 * rg) : make_uint2(0u, 0u);
    };
    request(blockIdx.x);
    for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
        int cq = 4 * rg;
        asm volatile("" : "+v"(cq));                               // opaque per iteration: the constant reads stay in the loop
        const int64_t tile = it * kPweBW + wave;
        const int64_t m = tile * 16 + px;
        const bool valid = m < p.M;
        const float vm = valid ? 1.f : 0.f;
        uint2 rgv[NT], ryv[NT], rxv[KT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { rgv[t] = ng[t]; ryv[t] = ny[t]; }
#pragma unroll
        for (int t = 0; t < KT; ++t) rxv[t] = nx[t];
        request(it + gridDim.x);
        const int col = 16 * wave + px;
        uint32_t bits = 0u;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int cl = 16 * t + cq;
            const float4 G = gate_widen(rgv[t]), Y = gate_widen(ryv[t]);
            const float4 sc = gate_ld4(cw + cl), sh = gate_ld4(cw + NP + cl), mu = gate_ld4(cw + 2 * NP + cl), is = gate_ld4(cw + 3 * NP + cl);
            const float gv[4] = {G.x, G.y, G.z, G.w}, yv[4] = {Y.x, Y.y, Y.z, Y.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float z = fmaf(yv[i], scv[i], shv[i]);
                const bool on = z > 0.f && z < ahi;                  // act' = on ? 1 : slope
                bits |= on ? (1u << (4 * t + i)) : 0u;
                const float dz = (on ? gv[i] : gv[i] * aslope) * vm;
                s1[t][i] += dz;
                s2[t][i] = fmaf(dz, (yv[i] - muv[i]) * isv[i], s2[t][i]);
                xb[(16 * t + 4 * rg + i) * kPwePitch + col] = (unsigned short)(gate_pack2(dz, 0.f) & 0xffffu);
            }
        }
        if (valid) p.mask[tile * 64 + lane] = bits;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int cl = 16 * t + cq;
            const float4 X = gate_widen(rxv[t]);
            const float4 xs = gate_ld4(ck + cl), xh = gate_ld4(ck + KP + cl);
            const float xv[4] = {X.x, X.y, X.z, X.w}, xsv[4] = {xs.x, xs.y, xs.z, xs.w}, xhv[4] = {xh.x, xh.y, xh.z, xh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float b = xv[i];
                if (has_xf) { const float xz = fmaf(xv[i], xsv[i], xhv[i]); b = fminf(fmaxf(xz, xslope * xz), xhi); }
                b = (16 * t + 4 * rg + i < K) ? b * vm : 0.f;        // padded channels: the view of 0 is not 0
                s3[t][i] += b;
                xa[(16 * t + 4 * rg + i) * kPwePitch + col] = (unsigned short)(gate_pack2(b, 0.f) & 0xffffu);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < JT; ++j) {                             // P1 tiles: rows n (xb), columns k (xa); Gram tiles: rows k (xa), columns k' (xa)
            const int q = wave + kPweBW * j;
            if (q < NTILE) {
                const bool gram = q >= NT * KT;
                const int qq = gram ? q - NT * KT : q;
                const int tr = qq / KT, tc = qq % KT;
                const unsigned short* pa = (gram ? xa : xb) + (16 * tr + px) * kPwePitch + 8 * rg;
                const unsigned short* pb = xa + (16 * tc + px) * kPwePitch + 8 * rg;
#pragma unroll
                for (int ks = 0; ks < kPweBW / 2; ++ks) {
                    const uint4 a = *reinterpret_cast<const uint4*>(pa + 32 * ks), b = *reinterpret_cast<const uint4*>(pb + 32 * ks);
                    wacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), wacc[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // one partial row of the workgroup: P1[N*K] | Gram[K*K] | s1[N] | s2[N] | s3[K]
    float* dst = p.partial + (int64_t)blockIdx.x * bnw_stride(N, K);
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        const int q = wave + kPweBW * j;
        if (q < NTILE) {
            const bool gram = q >= NT * KT;
            const int qq = gram ? q - NT * KT : q;
            const int tr = qq / KT, tc = qq % KT;
            const int c = 16 * tc + px;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * tr + 4 * rg + i;
                if (c < K && r < (gram ? K : N)) dst[(gram ? (int64_t)N * K : 0) + (int64_t)r * K + c] = wacc[j][i];
            }
        }
    }
    // sums: over the 16 pixel lanes, then the waves in order
    float* vdst = dst + (int64_t)N * K + (int64_t)K * K;
    auto fold = [&](float v) {
#pragma unroll
        for (int k = 1; k < 16; k <<= 1) v += __shfl_xor(v, k);
        return v;
    };
    constexpr int RW = 2 * NP + KP;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = fold(s1[t][i]), b = fold(s2[t][i]);
            if (px == 0) { red[wave * RW + 16 * t + 4 * rg + i] = a; red[wave * RW + NP + 16 * t + 4 * rg + i] = b; }
        }
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = fold(s3[t][i]);
            if (px == 0) red[wave * RW + 2 * NP + 16 * t + 4 * rg + i] = a;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * N + K; c += blockDim.x) {
        const int src = c < N ? c : (c < 2 * N ? NP + (c - N) : 2 * NP + (c - 2 * N));
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < kPweBW; ++w) a += red[w * RW + src];
        vdst[c] = a;
    }
}

template <int KT, int NT>
__global__ __launch_bounds__(256) void pwe_dgrad_kernel(PweArgs p) {
    constexpr int NU = (NT + 1) / 2, KU = (KT + 1) / 2, KP = KT * 16;
    __shared__ uint4 wb1[KT * NU * 64];                            // B1 chunks: rows k, contraction over n
    __shared__ uint4 wq[KT * KU * 64];                             // Q chunks: rows k, contraction over k' (Q is symmetric)
    __shared__ __attribute__((aligned(16))) float ck[3 * KP];      // in_scale | in_shift | bias
    const int K = p.K, N = p.N;
    for (int i = threadIdx.x; i < KT * NU * 64; i += blockDim.x) wb1[i] = gate_chunk(p.B1, K, N, N, 1, i / (64 * NU), (i >> 6) % NU, i & 63);
    for (int i = threadIdx.x; i < KT * KU * 64; i += blockDim.x) wq[i] = gate_chunk(p.Q, K, K, K, 1, i / (64 * KU), (i >> 6) % KU, i & 63);
    gate_fill_row(ck, p.xs, K, KP, 1.f); gate_fill_row(ck + KP, p.xb, K, KP, 0.f); gate_fill_row(ck + 2 * KP, p.bias, K, KP, 0.f);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, px = lane & 15, rg = lane >> 4;
    const int64_t ntiles = (p.M + 15) >> 4;
    const float aslope = act_slope(p.act), xslope = act_slope(p.xact), xhi = act_hi(p.xact);
    const bool has_xf = p.xs != nullptr || p.xact != MNY_ACT_NONE;
    const int64_t tstep = (int64_t)gridDim.x * 4;
    uint2 ng[NT], nx[KT];
    uint32_t nbits;
    auto request = [&](int64_t tile) {
        const int64_t mm = tile * 16 + px;
        const int64_t mr = mm < p.M ? mm : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) ng[t] = (16 * t + 4 * rg < N) ? *reinterpret_cast<const uint2*>(p.g + mr * N + 16 * t + 4 * rg) : make_uint2(0u, 0u);
#pragma unroll
        for (int t = 0; t < KT; ++t) nx[t] = (16 * t + 4 * rg < K) ? *reinterpret_cast<const uint2*>(p.x + mr * K + 16 * t + 4 * rg) : make_uint2(0u, 0u);
        nbits = p.mask[(tile < ntiles ? tile : 0) * 64 + lane];
    };
    request((int64_t)blockIdx.x * 4 + wave);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += tstep) {
        int lo = lane, cq = 4 * rg;
        asm volatile("" : "+v"(lo), "+v"(cq));
        const int64_t m = tile * 16 + px;
        const bool valid = m < p.M;
        uint2 rgv[NT], rxv[KT];
#pragma unroll
        for (int t = 0; t < NT; ++t) rgv[t] = ng[t];
#pragma unroll
        for (int t = 0; t < KT; ++t) rxv[t] = nx[t];
        const uint32_t bits = nbits;
        request(tile + tstep < ntiles ? tile + tstep : tile);
        auto dzval = [&](int t) {
            const float4 G = gate_widen(rgv[t]);
            return make_float4(((bits >> (4 * t + 0)) & 1u) ? G.x : G.x * aslope, ((bits >> (4 * t + 1)) & 1u) ? G.y : G.y * aslope,
                               ((bits >> (4 * t + 2)) & 1u) ? G.z : G.z * aslope, ((bits >> (4 * t + 3)) & 1u) ? G.w : G.w * aslope);
        };
        auto bval = [&](int t) {
            float4 v = gate_widen(rxv[t]);
            if (has_xf) {
                v = gate_fma4(v, gate_ld4(ck + 16 * t + cq), gate_ld4(ck + KP + 16 * t + cq));
                v = make_float4(fminf(fmaxf(v.x, xslope * v.x), xhi), fminf(fmaxf(v.y, xslope * v.y), xhi), fminf(fmaxf(v.z, xslope * v.z), xhi),
                                fminf(fmaxf(v.w, xslope * v.w), xhi));
                if (16 * t + 4 * rg >= K) v = f4zero();
            }
            return v;
        };
        gate_f4 acc[KT], acq[KT];
        gate_product<KT, NU>(acc, wb1 + lo, [&](int u) { return gate_frag(dzval(2 * u), 2 * u + 1 < NT ? dzval(2 * u + 1 < NT ? 2 * u + 1 : 0) : f4zero()); });
        gate_product<KT, KU>(acq, wq + lo, [&](int u) { return gate_frag(bval(2 * u), 2 * u + 1 < KT ? bval(2 * u + 1 < KT ? 2 * u + 1 : 0) : f4zero()); });
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int c0 = 16 * t + 4 * rg;
            const float4 bi = gate_ld4(ck + 2 * KP + 16 * t + cq);
            float4 o = gate_f(acc[t] + acq[t]);
            o.x += bi.x; o.y += bi.y; o.z += bi.z; o.w += bi.w;
            const bool ok = valid && c0 < K;
            if (p.addend != nullptr && ok) add4(o, ld4(p.addend + m * K + c0));
            if (ok) st4(p.dx + m * K + c0, o);
        }
    }
}

// shapes of the wave form: K = 16 / 24 / 32, N <= 80 (five 16-channel tiles: the per-lane sums of stage A fit 2 waves per SIMD; 8 tiles spilled), enough pixels
bool pwe_ok(int64_t M, int K, int N) {
    static const bool off = getenv("MNY_NO_PWE") != nullptr && atoi(getenv("MNY_NO_PWE")) != 0;
    return !off && M >= 131072 && (K == 16 || K == 24 || K == 32) && N > K && N <= 80 && (N & 7) == 0;
}
static inline int pwe_bw(int K) { return K <= 16 ? 4 : 8; }
int pwe_grid(int64_t M, int K) {
    const int64_t want = cdiv(cdiv(M, 16), pwe_bw(K));
    return (int)(want < 512 ? want : 512);
}
size_t pwe_ws_floats(int64_t M, int K, int N) {       // partial rows + reduced row + B1 + Q + bias + one mask word per lane and 16-pixel tile
    return (size_t)(pwe_grid(M, K) + 1) * bnw_stride(N, K) + (size_t)N * K + (size_t)K * K + 64 + 64 + (size_t)cdiv(M, 16) * 64 + 64;
}
template <int KT, int BW>
static int pwe_launch_a(const PweArgs& a, int NT, int grid, hipStream_t st) {
#define MNY_PWE_A(N_) do { constexpr size_t lds = PweLds<KT, N_, BW>::total; \
        if (lds > 64 * 1024 && !allow_lds((const void*)pwe_sums_kernel<KT, N_, BW>, lds)) { set_error("pw_bnbwd (wave form): hipFuncSetAttribute failed"); return MNY_EHIP; } \
        hipLaunchKernelGGL((pwe_sums_kernel<KT, N_, BW>), dim3(grid), dim3(64 * BW), lds, st, a); } while (0)
    if (NT <= 4) MNY_PWE_A(4); else MNY_PWE_A(5);
#undef MNY_PWE_A
    return check_launch("pwe_sums_kernel");
}
template <int KT>
static int pwe_launch_b(const PweArgs& a, int NT, int grid, hipStream_t st) {
    if (NT <= 4) hipLaunchKernelGGL((pwe_dgrad_kernel<KT, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pwe_dgrad_kernel<KT, 5>), dim3(grid), dim3(256), 0, st, a);
    return check_launch("pwe_dgrad_kernel");
}
// the whole unit: stage A -> combine + fp64 finalize (pw_bnbwd_finalize_launch, pwgemm.hip) -> stage B.  Same arguments as mny_pw_bnbwd_bf16.
int pwe_launch(const void* g, const void* y, const float* scale, const float* shift, int act, const float* mean, const float* invstd, const float* gamma,
               const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w, const void* addend, void* dx, float* dw,
               float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, hipStream_t st) {
    const int grid = pwe_grid(M, K);
    const int64_t stride = bnw_stride(Nc, K);
    float* red = ws + (size_t)grid * stride;
    float* B1 = red + stride;
    float* Q = B1 + (size_t)Nc * K;
    float* bias = Q + (size_t)K * K;
    uint32_t* mask = reinterpret_cast<uint32_t*>(bias + 64 + ((64 - ((bias + 64 - ws) & 15)) & 15));      // 64-byte aligned
    PweArgs a{(const bf16_t*)g, (const bf16_t*)y, scale, shift, act, mean, invstd, (const bf16_t*)x, in_scale, in_shift, in_act, ws, mask, M, K, Nc,
              B1, Q, bias, (const bf16_t*)addend, (bf16_t*)dx};
    const int NT = (Nc + 15) / 16;
    int rc = K <= 16 ? pwe_launch_a<1, 4>(a, NT, grid, st) : pwe_launch_a<2, 8>(a, NT, grid, st);
    if (rc) return rc;
    rc = pw_bnbwd_finalize_launch(ws, grid, red, w, gamma, mean, invstd, M, Nc, K, dw, dgamma, dbeta, B1, Q, bias, st);
    if (rc || !dx) return rc;
    const int64_t want = cdiv(cdiv(M, 16), 4);
    const int grid2 = (int)(want < 1024 ? want : 1024);
    return K <= 16 ? pwe_launch_b<1>(a, NT, grid2, st) : pwe_launch_b<2>(a, NT, grid2, st);
}

static bool gate_shape_ok(int64_t M, int C, int R) {
    return M > 0 && ((C == 40 && R == 10) || (C == 112 && R == 28) || (C == 160 && R == 40));
}
// forward: 4 waves per workgroup, two workgroups' worth of tiles per CU before the grid-stride loop takes over
static int gate_grid(int64_t M) {
    const int64_t want = cdiv(cdiv(M, 16), 4);
    return (int)(want < 512 ? want : 512);
}
// backward: 8 waves per workgroup; every workgroup leaves a partial row AND a partial weight gradient, so one resident round is the cap
static int gate_bwd_grid(int64_t M) {
    const int64_t want = cdiv(cdiv(M, 16), kGateBW);
    return (int)(want < 256 ? want : 256);
}

template <int PASS>
static int gate_fwd_launch(const GateFwdArgs& a, int C, int R, hipStream_t st) {
    const dim3 grid(gate_grid(a.M)), block(256);
    if (C == 40) hipLaunchKernelGGL((gate_fwd_kernel<40, 10, PASS>), grid, block, 0, st, a);
    else if (C == 112) hipLaunchKernelGGL((gate_fwd_kernel<112, 28, PASS>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gate_fwd_kernel<160, 40, PASS>), grid, block, 0, st, a);
    return check_launch("gate_fwd_kernel");
}

template <int C, int R, int PASS, bool RED3>
static int gate_bwd_launch_one(const GateBwdArgs& a, hipStream_t st) {
    constexpr size_t lds = GateBwdLds<C, R, PASS>::total;
    static_assert(lds <= 160 * 1024, "gate backward: LDS budget");
    if (lds > 64 * 1024 && !allow_lds((const void*)gate_bwd_kernel<C, R, PASS, RED3>, lds)) { set_error("gate_bwd: hipFuncSetAttribute failed"); return MNY_EHIP; }
    hipLaunchKernelGGL((gate_bwd_kernel<C, R, PASS, RED3>), dim3(gate_bwd_grid(a.M)), dim3(64 * kGateBW), lds, st, a);
    return check_launch("gate_bwd_kernel");
}
template <int PASS, bool RED3>
static int gate_bwd_launch(const GateBwdArgs& a, int C, int R, hipStream_t st) {
    if (C == 40) return gate_bwd_launch_one<40, 10, PASS, RED3>(a, st);
    if (C == 112) return gate_bwd_launch_one<112, 28, PASS, RED3>(a, st);
    return gate_bwd_launch_one<160, 40, PASS, false>(a, st);      // (C = 160 has no register room for the 80 extra accumulators: red3 refused below)
}

}  // namespace mny

using namespace mny;

extern "C" int mny_gate_supported(int64_t M, int C, int R) { return gate_shape_ok(M, C, R) ? 1 : 0; }
extern "C" int mny_gate_parts(int64_t M) { return M > 0 ? gate_grid(M) : MNY_EINVAL; }
extern "C" int mny_gate_bwd_parts(int64_t M) { return M > 0 ? gate_bwd_grid(M) : MNY_EINVAL; }
extern "C" int mny_gate_bwd_red3_supported(int C, int R) { return ((C == 40 && R == 10) || (C == 112 && R == 28)) ? 1 : 0; }
extern "C" size_t mny_gate_wq_bytes(int C, int R) {
    if (C <= 0 || R <= 0) return 0;
    const int CT = (C + 15) / 16, CU = (CT + 1) / 2, RT = (R + 15) / 16, RU = (RT + 1) / 2;
    return (size_t)2 * (RT * CU + CT * RU) * 64 * 16;
}

extern "C" int mny_gate_cut_batch_bf16(const mny_gate_cut_job* jobs, int njobs, void* stream) {
    MNY_REQUIRE(jobs && njobs > 0, "gate_cut_batch: bad arguments");
    hipLaunchKernelGGL(gate_cut_kernel, dim3(16, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return check_launch("gate_cut_kernel");
}

extern "C" int mny_gate_stats1_bf16(const void* y3, const float* s3, const float* b3, const void* wq, float* stats, int64_t M, int C, int R, void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && wq && stats, "gate_stats1: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_stats1: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    GateFwdArgs a{};
    a.y3 = (const bf16_t*)y3; a.s3 = s3; a.b3 = b3; a.wq = (const uint4*)wq; a.parts = stats; a.M = M;
    return gate_fwd_launch<1>(a, C, R, (hipStream_t)stream);
}

extern "C" int mny_gate_stats2_bf16(const void* y3, const float* s3, const float* b3, const void* wq, const float* sc1, const float* sh1,
                                    float* stats, int64_t M, int C, int R, void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && wq && sc1 && sh1 && stats, "gate_stats2: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_stats2: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    GateFwdArgs a{};
    a.y3 = (const bf16_t*)y3; a.s3 = s3; a.b3 = b3; a.wq = (const uint4*)wq; a.sc1 = sc1; a.sh1 = sh1; a.parts = stats; a.M = M;
    return gate_fwd_launch<2>(a, C, R, (hipStream_t)stream);
}

extern "C" int mny_gate_fwd_bf16(const void* y3, const float* s3, const float* b3, const void* wq, const float* sc1, const float* sh1,
                                 const float* sc2, const float* sh2, const void* add_x, const float* add_scale,
                                 const float* add_shift, int add_act, void* out, int64_t M, int C, int R, void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && wq && sc1 && sh1 && sc2 && sh2 && out, "gate_fwd: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_fwd: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    MNY_REQUIRE(!add_scale == !add_shift, "gate_fwd: scale and shift of the residual operand come together");
    GateFwdArgs a{};
    a.y3 = (const bf16_t*)y3; a.s3 = s3; a.b3 = b3; a.wq = (const uint4*)wq; a.sc1 = sc1; a.sh1 = sh1; a.sc2 = sc2; a.sh2 = sh2;
    a.add_x = (const bf16_t*)add_x; a.add_scale = add_scale; a.add_shift = add_shift; a.add_act = add_act;
    a.out = (bf16_t*)out; a.M = M;
    return gate_fwd_launch<3>(a, C, R, (hipStream_t)stream);
}

static GateBwdArgs gate_bwd_args(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* bn1, const float* bn2, int64_t M) {
    GateBwdArgs a{};
    a.y3 = (const bf16_t*)y3; a.s3 = s3; a.b3 = b3; a.dout = (const bf16_t*)dout; a.wq = (const uint4*)wq; a.M = M;
    (void)bn1; (void)bn2;
    return a;
}

extern "C" int mny_gate_bwd1_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1, const float* sh1,
                                  const float* sc2, const float* sh2, const float* mean2, const float* invstd2, float* red2, int64_t M, int C, int R,
                                  void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && dout && wq && sc1 && sh1 && sc2 && sh2 && mean2 && invstd2 && red2, "gate_bwd1: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_bwd1: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    GateBwdArgs a = gate_bwd_args(y3, s3, b3, dout, wq, nullptr, nullptr, M);
    a.sc1 = sc1; a.sh1 = sh1; a.sc2 = sc2; a.sh2 = sh2; a.mean2 = mean2; a.invstd2 = invstd2; a.parts = red2;
    return gate_bwd_launch<1, false>(a, C, R, (hipStream_t)stream);
}

extern "C" int mny_gate_bwd2_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1, const float* sh1,
                                  const float* mean1, const float* invstd1, const float* sc2, const float* sh2, const float* coef2, float* red1,
                                  float* dw2_parts, int64_t M, int C, int R, void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && dout && wq && sc1 && sh1 && mean1 && invstd1 && sc2 && sh2 && coef2 && red1 && dw2_parts, "gate_bwd2: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_bwd2: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    GateBwdArgs a = gate_bwd_args(y3, s3, b3, dout, wq, nullptr, nullptr, M);
    a.sc1 = sc1; a.sh1 = sh1; a.mean1 = mean1; a.invstd1 = invstd1; a.sc2 = sc2; a.sh2 = sh2; a.coef2 = coef2; a.parts = red1; a.dwp = dw2_parts;
    return gate_bwd_launch<2, false>(a, C, R, (hipStream_t)stream);
}

extern "C" int mny_gate_bwd3_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1, const float* sh1,
                                  const float* sc2, const float* sh2, const float* coef2, const float* coef1, const float* mean3, const float* invstd3,
                                  void* dt, float* dw1_parts, float* red3, int64_t M, int C, int R, void* stream) {
    MNY_REQUIRE(y3 && s3 && b3 && dout && wq && sc1 && sh1 && sc2 && sh2 && coef2 && coef1 && dt && dw1_parts, "gate_bwd3: null pointer");
    MNY_REQUIRE(gate_shape_ok(M, C, R), "gate_bwd3: M=%lld C=%d R=%d not supported", (long long)M, C, R);
    MNY_REQUIRE(red3 == nullptr || (mean3 && invstd3 && mny_gate_bwd_red3_supported(C, R)), "gate_bwd3: red3 needs mean3 / invstd3 and a shape mny_gate_bwd_red3_supported() accepts");
    GateBwdArgs a = gate_bwd_args(y3, s3, b3, dout, wq, nullptr, nullptr, M);
    a.sc1 = sc1; a.sh1 = sh1; a.sc2 = sc2; a.sh2 = sh2; a.coef2 = coef2; a.coef1 = coef1; a.mean3 = mean3; a.invstd3 = invstd3;
    a.dt = (bf16_t*)dt; a.dwp = dw1_parts; a.red3 = red3;
    return red3 ? gate_bwd_launch<3, true>(a, C, R, (hipStream_t)stream) : gate_bwd_launch<3, false>(a, C, R, (hipStream_t)stream);
}

extern "C" int mny_pj_bwd_supported_bf16(int64_t M, int Ki, int No, int d_act) {
    static const bool off = getenv("MNY_NO_PJBWD") != nullptr;
    return (!off && pj16_shape_ok(M, Ki, No) && d_act != MNY_ACT_HSIGMOID) ? 1 : 0;
}
extern "C" int mny_pj_bwd_parts_bf16(int64_t M, int Ki, int No) { return pj16_shape_ok(M, Ki, No) ? pj16_grid(M) : MNY_EINVAL; }

extern "C" int mny_pj_bwd_bf16(const void* g, const void* y, const float* coef, const void* d, const float* d_scale, const float* d_shift,
                               const float* d_mean, const float* d_invstd, int d_act, const float* w, void* gd, float* dw, float* dw_ws, float* red,
                               int64_t M, int Ki, int No, void* stream) {
    MNY_REQUIRE(g && y && coef && d && d_scale && d_shift && d_mean && d_invstd && w && gd && dw_ws && red, "pj_bwd_bf16: null pointer");
    MNY_REQUIRE(pj16_shape_ok(M, Ki, No) && d_act != MNY_ACT_HSIGMOID, "pj_bwd_bf16: M=%lld Ki=%d No=%d act %d not supported", (long long)M, Ki, No, d_act);
    Pj16Args a{(const bf16_t*)g, (const bf16_t*)y, coef, (const bf16_t*)d, d_scale, d_shift, d_mean, d_invstd, d_act, w, (bf16_t*)gd, dw_ws, red, M};
    const int grid = pj16_grid(M);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (Ki == 16) rc = pj16_launch<16, 16>(a, grid, st);
    else if (Ki == 64) rc = pj16_launch<64, 24>(a, grid, st);
    else if (Ki == 72 && No == 24) rc = pj16_launch<72, 24>(a, grid, st);
    else if (Ki == 72) rc = pj16_launch<72, 40>(a, grid, st);
    else rc = pj16_launch<120, 40>(a, grid, st);
    if (rc || !dw) return rc;                          // dw == NULL: partial rows only (combined later by mny_reduce_batch)
    return launch_reduce_parts(dw_ws, grid, No * Ki, dw, st);
}
