// The six-product bf16 form of an fp32 product (shared by pwgemm.hip and pwwide.hip).
#pragma once
#include "common.h"

namespace mny {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// X6 (fp32 operands only): the products run on the bf16 matrix cores.  Every fp32 operand is cut into three bf16 pieces by
// truncation (top 8 significant bits, next 8, last 8: hi + mid + lo == the fp32 value EXACTLY), a bf16 x bf16 product is exact in
// fp32, and six of the nine partial products are accumulated (lo*mid, mid*lo, lo*lo <= 2^-24 of the product are dropped): the
// result differs from an fp32 FMA chain by about one fp32 rounding per product.  One v_mfma_f32_32x32x16_bf16 (32 cycles) covers a
// whole 16-deep stage that takes eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 6 x 32 = 192 matrix-pipe cycles instead of 512
// per accumulator and stage; the cuts cost ~36 VALU instructions per 8-value fragment.
// Non-finite operands: the cut is exact for every FINITE fp32 value (denormals included; the bf16 pipe may flush denormal pieces, an
// error below 1e-38 per product).  +-inf gives mid = inf - inf = NaN, and passing the infinity through in `hi` alone would not help:
// hi(inf) * mid(w) is inf * 0 = NaN for every bf16-representable w.  So an inf or NaN operand poisons exactly the outputs it
// poisons in an fp32 FMA chain, but always as NaN — the SIGN of an infinity is lost.  Pinned by tests/test_gpu_kernels.py
// (test_six_product_form_on_non_finite_...: same non-finite set as MNY_X6=0, finite outputs equal); MNY_X6=0 is the strict-fp32 mode.
__device__ __forceinline__ void x6_split(v4f_t x0, v4f_t x1, bf16x8_t& h, bf16x8_t& m, bf16x8_t& l) {
    float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    // The operands are pinned as ROUNDED fp32 values: where x is the result of a multiplication in the caller (the h-swish view's
    // `z * clamp * (1/6)`), -ffp-contract=fast would fold that multiply into the first subtraction below (fma(t, 1/6, -hi(x))) and cut the
    // UNROUNDED product — hi + mid + lo would no longer equal the value the other kernels see (round 5: the in-kernel and the pre-cut
    // forms differed in the last bit behind h-swish views once the division by 6 had become a multiplication).
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x[e]));
    float r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {                 // pairs: the subtractions are v_pk_add_f32
        const v2f a = v2f{x[e], x[e + 1]};
        const v2f ah = v2f{__uint_as_float(__float_as_uint(x[e]) & 0xffff0000u), __uint_as_float(__float_as_uint(x[e + 1]) & 0xffff0000u)};
        const v2f b = a - ah;                        // exact: the low 16 mantissa bits
        const v2f bh = v2f{__uint_as_float(__float_as_uint(b.x) & 0xffff0000u), __uint_as_float(__float_as_uint(b.y) & 0xffff0000u)};
        const v2f c = b - bh;
        r1[e] = b.x; r1[e + 1] = b.y; r2[e] = c.x; r2[e + 1] = c.y;
    }
    v4u_t hu, mu, lu;
    // v_perm_b32: (odd element's high half << 16) | even element's high half == two truncated bf16 values
    hu.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u); hu.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
    hu.z = __builtin_amdgcn_perm(__float_as_uint(x[5]), __float_as_uint(x[4]), 0x07060302u); hu.w = __builtin_amdgcn_perm(__float_as_uint(x[7]), __float_as_uint(x[6]), 0x07060302u);
    mu.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u); mu.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
    mu.z = __builtin_amdgcn_perm(__float_as_uint(r1[5]), __float_as_uint(r1[4]), 0x07060302u); mu.w = __builtin_amdgcn_perm(__float_as_uint(r1[7]), __float_as_uint(r1[6]), 0x07060302u);
    lu.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u); lu.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
    lu.z = __builtin_amdgcn_perm(__float_as_uint(r2[5]), __float_as_uint(r2[4]), 0x07060302u); lu.w = __builtin_amdgcn_perm(__float_as_uint(r2[7]), __float_as_uint(r2[6]), 0x07060302u);
    h = __builtin_bit_cast(bf16x8_t, hu); m = __builtin_bit_cast(bf16x8_t, mu); l = __builtin_bit_cast(bf16x8_t, lu);
}

}  // namespace mny
