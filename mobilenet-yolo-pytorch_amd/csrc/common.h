// Shared helpers for libmnyolo (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/mnyolo.h"

namespace mny {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
bool allow_lds(const void* kernel, size_t bytes);     // opt a kernel in to > 64 KB of dynamic LDS, once per (kernel, device)

#define MNY_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            mny::set_error(__VA_ARGS__);       \
            return MNY_EINVAL;                 \
        }                                      \
    } while (0)

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kMaxParts = 1024;  // upper bound on stat/partial rows (4 blocks x 256 CUs)

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Branch-free activations.  relu6 / leaky(0.1) / relu / identity are all  min(max(z, slope*z), hi)  with
// (slope, hi) = (0,6) / (0.1,inf) / (0,inf) / (1,inf); `act` is wave-uniform, so slope/hi live in SGPRs and the
// per-element cost is v_mul + v_max + v_min (a per-element switch made the HBM-bound stencils VALU-bound).
__device__ __forceinline__ float act_slope(int act) { return act == MNY_ACT_NONE ? 1.f : (act == MNY_ACT_LEAKY ? 0.1f : 0.f); }
__device__ __forceinline__ float act_hi(int act) { return act == MNY_ACT_RELU6 ? 6.f : INFINITY; }
// h-swish / h-sigmoid multiply by the constant 1/6 where the reference divides by 6 (models/mobilenetv3.py:16,22): `x / 6.f` compiles to the
// correctly-rounded IEEE sequence (v_div_scale, v_rcp, four fma, v_div_fmas, v_div_fixup: 11 instructions per ELEMENT) — in a GEMM's
// A-operand view that doubled the kernel (bf16 M 65 536 x K 672 x N 160: 27 us plain, 52 us behind an h-swish view, round 5); the
// product differs from the quotient by at most one fp32 ulp.
__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act >= MNY_ACT_HSWISH) {
        const float h = fminf(fmaxf(z + 3.f, 0.f), 6.f) * (1.f / 6.f);
        return act == MNY_ACT_HSWISH ? z * h : h;
    }
    return fminf(fmaxf(z, act_slope(act) * z), act_hi(act));
}
// derivative at pre-activation z (torch's subgradient choices: relu6 = hardtanh: 1 on the open interval (0,6);
// leaky_relu: z > 0 ? 1 : slope)
__device__ __forceinline__ float act_bwd(float z, int act) {
    if (act == MNY_ACT_HSWISH) return z <= -3.f ? 0.f : (z >= 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
    if (act == MNY_ACT_HSIGMOID) return (z > -3.f && z < 3.f) ? (1.f / 6.f) : 0.f;
    return (z > 0.f ? 1.f : act_slope(act)) * (z < act_hi(act) ? 1.f : 0.f);
}

__device__ __forceinline__ float4 xform4(float4 v, float4 sc, float4 sh, int act) {
    float4 r;
    r.x = act_fwd(fmaf(v.x, sc.x, sh.x), act);
    r.y = act_fwd(fmaf(v.y, sc.y, sh.y), act);
    r.z = act_fwd(fmaf(v.z, sc.z, sh.z), act);
    r.w = act_fwd(fmaf(v.w, sc.w, sh.w), act);
    return r;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// bf16 STORAGE type of the `*_bf16` entry points: activations / activation gradients live in HBM as bf16, every kernel
// widens on load and rounds (RNE, v_cvt_pk_bf16_f32) on store; arithmetic, statistics and weights stay fp32.
// ld4/st4/ld1/st1 are overloaded on the pointer type so a kernel templated on T reads the same for both.
struct bf16_t { uint16_t v; };
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    const f2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf2_t));
}
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}
// streaming (non-temporal) 16-B / 8-B stores for outputs no workgroup of the same kernel reads back: measured on MI355X
// (tools/probe/dw_probe.hip) the write path is the slow side of HBM (write-only 4.6 TB/s vs read-only 6.4 TB/s) and `nt` stores
// lift a 1:1 read/write stream by 2-15 %
typedef float mny_f4v __attribute__((ext_vector_type(4)));
typedef unsigned mny_u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
    const mny_f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<mny_f4v*>(p));
}
__device__ __forceinline__ void st4_stream(bf16_t* p, float4 v) {
    const mny_u2v t = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    __builtin_nontemporal_store(t, reinterpret_cast<mny_u2v*>(p));
}
// uniform base + 32-bit BYTE offset: the form the compiler turns into `global_load/store v, v_off, s[base:base+1]`
template <typename T> __device__ __forceinline__ T* at_bytes(T* base, unsigned byte_off) {
    return (T*)((char*)base + byte_off);
}
template <typename T> __device__ __forceinline__ const T* at_bytes(const T* base, unsigned byte_off) {
    return (const T*)((const char*)base + byte_off);
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((uint32_t)p->v << 16); }
__device__ __forceinline__ void st1(bf16_t* p, float v) { p->v = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
// the value a consumer will read back after a store of v as T (BN statistics are taken over THESE values)
template <typename T> __device__ __forceinline__ float stored(float v);
template <> __device__ __forceinline__ float stored<float>(float v) { return v; }
template <> __device__ __forceinline__ float stored<bf16_t>(float v) { return __uint_as_float(pack_bf16x2(v, 0.f) << 16); }
template <typename T> __device__ __forceinline__ float4 stored4(float4 v) {
    return make_float4(stored<T>(v.x), stored<T>(v.y), stored<T>(v.z), stored<T>(v.w));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4one() { return make_float4(1.f, 1.f, 1.f, 1.f); }
__device__ __forceinline__ void fma4(float4& acc, float4 a, float4 b) {
    acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y);
    acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
}
__device__ __forceinline__ void add4(float4& acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }

// packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): a float4 as two register pairs — the stencil kernels are VALU-heavy
// enough (3x3 taps x 4 channels, BN-apply / activation per tap) that halving the fma / mul count shows up in step time
typedef float v2f __attribute__((ext_vector_type(2)));
struct F4P { v2f lo, hi; };
__device__ __forceinline__ F4P f4p(float4 v) { F4P r; r.lo = v2f{v.x, v.y}; r.hi = v2f{v.z, v.w}; return r; }
__device__ __forceinline__ F4P f4p0() { F4P r; r.lo = v2f{0.f, 0.f}; r.hi = v2f{0.f, 0.f}; return r; }
__device__ __forceinline__ float4 f4u(F4P v) { return make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y); }
__device__ __forceinline__ void pfma(F4P& acc, F4P a, F4P b) {
    acc.lo = __builtin_elementwise_fma(a.lo, b.lo, acc.lo);
    acc.hi = __builtin_elementwise_fma(a.hi, b.hi, acc.hi);
}

// ---- LDS reads the compiler cannot see ------------------------------------------------------------------------------
// hipcc treats every LDS read after a `global_load_lds` as possibly aliasing the DMA's LDS write and puts `s_waitcnt vmcnt(0)`
// in front of it (SIInsertWaitcnts; no alias-scope information survives from HIP source).  In a multi-stage ring that wait
// sits right after the NEXT stages have been issued, i.e. it drains the whole queue every k-step and the ring degenerates to
// "issue, wait for everything, compute".  The fragment reads of the pipelined loops are therefore raw `ds_read_b128`
// (volatile inline asm) followed by an explicit `s_waitcnt lgkmcnt(0)` that carries the destination registers as in/out
// operands, so every use is ordered after the wait; the counted `s_waitcnt vmcnt(N)` + `s_barrier` at the top of the loop
// are what guarantee that the stage being read has landed.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ v4f_t lds_read_f4(const float* p) {
    v4f_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_off(p)) : "memory");
    return v;
}
__device__ __forceinline__ v4u_t lds_read_u4(const float* p) {
    v4u_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_off(p)) : "memory");
    return v;
}
// ds_read_b128 at byte address `base` + the immediate OFF (< 64 KB): no address arithmetic per read
template <int OFF>
__device__ __forceinline__ v4f_t lds_read_f4_at(unsigned base) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    v4f_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF) : "memory");
    return v;
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
#define MNY_LGKM_WAIT(first) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(first))
#define MNY_LGKM_DEP(x) asm volatile("" : "+v"(x))

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Channel-group thread layout shared by the NHWC stencil / elementwise kernels:
// a thread owns 4 consecutive channels; `cgb` channel groups per block (<=256), `ppb` pixels per block.
struct CgLayout {
    int cg_total;   // C/4
    int chunks;     // channel chunks (gridDim.y)
    int cgb;        // channel groups per block
    int ppb;        // pixels (or strips) per block
    int threads;    // cgb*ppb
};
inline CgLayout make_cg_layout(int C) {
    CgLayout L;
    L.cg_total = C / 4;
    L.chunks = (int)cdiv(L.cg_total, 256);
    L.cgb = (int)cdiv(L.cg_total, L.chunks);
    L.ppb = 256 / L.cgb; if (L.ppb < 1) L.ppb = 1;
    L.threads = L.cgb * L.ppb;
    return L;
}


// Stencil kernels (a thread = 4 channels x one column) re-read one halo column on each side of a block's column span, and
// neighbouring blocks sit on different XCDs / L2s: with <= 2 columns per block (C >= 512) the depthwise kernels fetched 2-2.7x
// their input.  Channels are independent, so wide layers are split into channel chunks of <= `max_cgb` groups (grid.y) and a
// block spans >= 256/max_cgb columns: halo ratio (ppb + 2) / ppb.  Measured (same box, ms/step dw_bnbwd + dw_fwd + s2 kernels):
// 256: 13.0, 64: 12.1, 32: 12.8, 16: 13.3 -> 64 (narrower chunks lose more in per-pixel contiguity than they save in halo).
// MNY_STENCIL_CGB overrides max_cgb (profiling).
inline CgLayout make_stencil_layout(int C, int max_cgb = 64) {
    static const int env = getenv("MNY_STENCIL_CGB") ? atoi(getenv("MNY_STENCIL_CGB")) : 0;
    if (env > 0) max_cgb = env;
    CgLayout L;
    L.cg_total = C / 4;
    L.chunks = (int)cdiv(L.cg_total, max_cgb);
    L.cgb = (int)cdiv(L.cg_total, L.chunks);
    L.ppb = 256 / L.cgb; if (L.ppb < 1) L.ppb = 1;
    L.threads = L.cgb * L.ppb;
    return L;
}

// partial rows [parts][n] -> out[n]: 32 outputs x 8 part-slices per block, fp64 combine, fixed order
__global__ void reduce_parts_f64_kernel(const float* __restrict__ parts, int nparts, int n, float* __restrict__ out);
int launch_reduce_parts(const float* parts, int nparts, int n, float* out, hipStream_t st);

// tile form of the fused depthwise-unit backward (dwtile.hip): 5x5 stride 1 (and 3x3 behind MNY_DWT3=1); same arguments as mny_dw_bnbwd[_red]
bool dwt_use(int K, int bf, int red, int C);      // which form runs the unit
int dwt_parts(int N, int H, int W, int C, int K, int bf);
bool dwt_fwd_use(int K, int stride, int bf, int C);    // tile form of the depthwise forward (bf16 storage, stride 1)
int dwt_fwd_parts(int N, int H, int W, int C, int K);
int dwt_fwd_launch(int bf, const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w, void* y, float* stats, int N, int H,
                   int W, int C, int K, void* stream);
int dwt_launch(int bf, const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef, const void* x,
               const float* in_scale, const float* in_shift, int in_act, const float* w, const void* addend, void* dx, float* dw, float* ws, int N, int H,
               int W, int C, int K, void* stream, const float* in_mean, const float* in_invstd, float* in_red);

// thin pointwise conv forward / plain data gradient, bf16 storage, a wave per 16 pixels on the matrix cores (gate.hip)
bool pwt_ok(int64_t M, int K, int N, int in_act, bool has_bias);
int pwt_parts(int64_t M);
int pwt_launch(const void* x, const float* xs, const float* xb, int xact, const void* w, const void* addend, void* y, float* stats, int64_t M, int K, int N,
               hipStream_t st, const void* rY = nullptr, const float* r_scale = nullptr, const float* r_shift = nullptr, const float* r_mean = nullptr,
               const float* r_invstd = nullptr, int r_act = 0);      // rY != NULL: the data-gradient + BN-backward-sums form (stats = the partial rows)

// thin expand unit backward on bf16 storage, wave form (gate.hip pwe_sums_kernel / pwe_dgrad_kernel); pw_bnbwd_finalize_launch: pwgemm.hip
bool pwe_ok(int64_t M, int K, int N);
size_t pwe_ws_floats(int64_t M, int K, int N);
int pwe_launch(const void* g, const void* y, const float* scale, const float* shift, int act, const float* mean, const float* invstd, const float* gamma,
               const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w, const void* addend, void* dx, float* dw,
               float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, hipStream_t st);

// short-reduction pointwise conv on the vector ALU (pwthin.hip); the entry points of pwgemm.hip route K = 8/16/24/32 problems here
bool pw_thin_ok(int bf, int red, int64_t M, int K, int N);
int pw_thin_parts(int64_t M, int K, int N, int red);
int pw_thin_launch(int bf, const void* A, const float* in_scale, const float* in_shift, int in_act, const void* W, const float* bias,
                   const void* addend, void* C, float* stats, int64_t M, int K, int N, int red, const void* rY, const float* r_scale,
                   const float* r_shift, const float* r_mean, const float* r_invstd, int r_act, hipStream_t st);

// short reduction feeding a wide output (K = 52..96, N >= K): barrier-free matrix-core kernel (pwwide.hip); fp32 storage only
bool pw_wide_ok(int64_t M, int K, int N, bool red);       // red: the data-gradient + BN-backward-sums form
int pw_wide_parts(int64_t M, int K, int N, bool red);       // partial rows of the whole 32-row tiles
int pw_wide_launch(const float* A, const float* in_scale, const float* in_shift, int in_act, const float* W, float* C, float* stats,
                   int64_t M, int K, int N, const float* rY, const float* r_scale, const float* r_shift, const float* r_mean,
                   const float* r_invstd, int r_act, const float* addend, hipStream_t st);

// mny_pw_lr_fix on the barrier-free kernel (K = 64 / 96, whole 32-row tiles): dx = (in_scale o x + in_shift) Q + bias + addend, + the sums
bool pw_wide_fix_ok(int64_t M, int K);
int pw_wide_fix_parts(int64_t M, int K);
int pw_wide_fix_launch(const float* A, const float* in_scale, const float* in_shift, const float* Q, const float* bias, const float* addend, float* C,
                       float* stats, int64_t M, int K, const float* rY, const float* r_scale, const float* r_shift, const float* r_mean,
                       const float* r_invstd, int r_act, hipStream_t st);

// weight gradient with one narrow side (64 / 96 channels): barrier-free stream kernel (pwwgs.hip); fp32 storage, no bias gradient
bool pw_wgs_ok(int64_t M, int K, int N);
int pw_wgs_splits(int64_t M, int K, int N);
int pw_wgs_launch(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* partial,
                  int64_t M, int K, int N, hipStream_t st);

// fused BN-backward of a thin expand unit (pwgemm.hip): partial row layout  P1[N*K] | Gram[K*K] | s1[N] | s2[N] | s3[K]  and its fp64 finalize
// (-> dW, dgamma, dbeta and the data-gradient operands B1[K][N] = ca o W^T, Q[K][K] = W^T diag(cb) W, bias[K] = cc . W)
__host__ __device__ inline int64_t bnw_stride(int N, int K) { return (int64_t)N * K + (int64_t)K * K + 2 * N + K; }
int pw_bnbwd_finalize_launch(const float* partials, int nparts, float* red, const float* w, const float* gamma, const float* mean,
                             const float* invstd, int64_t M, int Nc, int K, float* dw, float* dgamma, float* dbeta, float* B1, float* Q,
                             float* bias, hipStream_t st);

}  // namespace mny
