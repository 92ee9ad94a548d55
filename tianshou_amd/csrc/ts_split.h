// ts_split.h -- fp32 operands as three bf16 pieces for the bf16 matrix cores (see ts_ppo_step3.h for the derivation):
// x = x0 + x1 + x2 with xk = bf16_rne of the running remainder (exact), a product sum = six bf16 MFMAs with fp32
// accumulation (a0 b0 + a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0, smallest first); dropped terms <= 2^-26 |a b|.
#pragma once
#include <hip/hip_runtime.h>

namespace tsplit {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

struct Pk3 { unsigned p0, p1, p2; };

// (a, b) -> three packed registers (low half a's piece, high half b's)
__device__ __forceinline__ Pk3 split_pair(float a, float b) {
    Pk3 o;
    o.p0 = cvt_pk(a, b);
    const float ra = a - __uint_as_float(o.p0 << 16), rb = b - __uint_as_float(o.p0 & 0xffff0000u);       // exact
    o.p1 = cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(o.p1 << 16), sb = rb - __uint_as_float(o.p1 & 0xffff0000u);     // exact
    o.p2 = cvt_pk(sa, sb);
    return o;
}

struct P3 { u32x4 p[3]; };       // the 8 k-slots a lane contributes to one K = 16 chunk, three pieces

__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mma6(const P3& a, const P3& b, f32x16 c) {
    c = mfma_bf(a.p[2], b.p[0], c);
    c = mfma_bf(a.p[1], b.p[1], c);
    c = mfma_bf(a.p[0], b.p[2], c);
    c = mfma_bf(a.p[1], b.p[0], c);
    c = mfma_bf(a.p[0], b.p[1], c);
    c = mfma_bf(a.p[0], b.p[0], c);
    return c;
}

}  // namespace tsplit
