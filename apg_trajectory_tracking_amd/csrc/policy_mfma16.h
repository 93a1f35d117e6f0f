// policy_mfma16.h - the policy layers on the 16-bit matrix pipe at fp32 accuracy.
//
// v_mfma_f32_32x32x16_f16 multiplies 8 x more reduction elements per
// instruction than v_mfma_f32_32x32x2_f32 in half the cycles.  Every fp32
// operand x is split into TWO fp16 terms, x = x_h + x_l + O(2^-22 |x|)
// (x_h = fp16(x), x_l = fp16(x - x_h): 11 + 11 significant bits), and a product
// sum is evaluated as three matrix instructions into the same fp32
// accumulator:  W_l x_h + W_h x_l + W_h x_h  (the fourth, W_l x_l, is 2^-22 of
// the result).  tools/mfma_split_probe.hip: one 64 -> 64 layer with tanh is
// exact to 2.4e-7 of the output range - the same as the fp32 instruction -
// and takes 1.89 us against 4.94 us (all 2 048 waves of a 65 536 batch).
// Range: fp16 holds |x| < 65 504 and the low term of |x| < 2^-3 goes
// subnormal (absolute error <= 2^-25): fine for bounded operands - weights,
// tanh / relu activations, reference windows in metres; cotangents must be
// scaled per trajectory first (scaled_split).
//
// Operand convention (as policy_mfma.h: one wave = 32 trajectories, lane l
// works for trajectory l & 31, half-wave hi = l >> 5):
//   A operand: lane l supplies the 8 weights W[row l & 31][k-slots 8 hi + j]
//   B operand: lane l supplies the 8 values  x[k-slots 8 hi + j][column l & 31]
//   C / D    : register i of lane l is row r(i) + 4 hi, column l & 31.
// Which input a k-slot means is free as long as A and B agree: for a layer fed
// by accumulator registers, k-block kb (16 slots) takes registers
// 8 (kb & 1) .. + 7 of row block kb >> 1 - input index kin(kb, j, hi) - so
// layers still chain without shuffles; the packed weights carry the order.
#pragma once
#include "policy_mfma.h"

namespace apg {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// input index of slot j of k-block kb for half-wave hi (64-wide layers)
__host__ __device__ constexpr int kin(int kb, int j, int hi) {
  return 32 * (kb >> 1) + rrow(8 * (kb & 1) + j) + 4 * hi;
}

// two fp32 values -> their packed fp16 high terms and packed fp16 low terms
__device__ __forceinline__ void split_pair(float a, float b, unsigned &h, unsigned &l) {
  const h16x2 vh = {(_Float16)a, (_Float16)b};  // v_cvt_pk_f16_f32, round to nearest
  h = __builtin_bit_cast(unsigned, vh);
  // the remainders a - a_h, b - b_h (exact) as ONE mixed-precision fma each:
  // v_fma_mix_f32 reads the fp16 term out of the packed register (the compiler
  // does not form it: it converts back and subtracts, two instructions)
  float ra, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]"
      : "=v"(ra) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=v"(rb) : "v"(h), "v"(b));
  const h16x2 vl = {(_Float16)ra, (_Float16)rb};
  l = __builtin_bit_cast(unsigned, vl);
}

struct Op16 {  // one operand of a k-block: 8 high terms, 8 low terms
  u32x4 h, l;
};

// the 8 values v[0..7] of this lane's k-slots as a B operand
__device__ __forceinline__ Op16 split8(const float (&v)[8]) {
  Op16 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h, l;
    split_pair(v[2 * q], v[2 * q + 1], h, l);
    o.h[q] = h, o.l[q] = l;
  }
  return o;
}

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a),
                                                __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// acc += W x over one k-block: the three significant products, small first
__device__ __forceinline__ f32x16 mma3(const Op16 &w, const Op16 &x, f32x16 acc) {
  acc = mfma16(w.l, x.h, acc);
  acc = mfma16(w.h, x.l, acc);
  return mfma16(w.h, x.h, acc);
}

// A-operand blocks in LDS: block n = 64 lanes x 16 B of high terms followed by
// 64 x 16 B of low terms (2 KB), the blocks start `base` bytes into LDS.
// ds_read_b128 carries a 16-bit byte offset and the tables span > 64 KB: two
// opaque per-lane bases keep every read "base VGPR + immediate" (cf. LdsView).
constexpr int kBlock16 = 2048;
struct LdsView16 {
  const char *b0, *b1;
  __device__ __forceinline__ LdsView16(const void *lds, int lane) {
    unsigned o0 = (unsigned)lane * 16u, o1 = (unsigned)lane * 16u + 61440u;
    asm volatile("" : "+v"(o0), "+v"(o1));
    b0 = static_cast<const char *>(lds) + o0;
    b1 = static_cast<const char *>(lds) + o1;
  }
  __device__ __forceinline__ u32x4 ld(int byte) const {
    return byte < 61440 ? *reinterpret_cast<const u32x4 *>(b0 + byte)
                        : *reinterpret_cast<const u32x4 *>(b1 + (byte - 61440));
  }
  // block n of the table that starts at byte `base`
  __device__ __forceinline__ Op16 A(int base, int n) const {
    Op16 o;
    o.h = ld(base + n * kBlock16);
    o.l = ld(base + n * kBlock16 + 1024);
    return o;
  }
};

// Cotangents have no bounded range: each trajectory's vector is scaled by a
// power of two (exact) that brings its largest entry into [0.5, 1) before the
// split, and the product is scaled back by the consumer (ldexp with the
// returned exponent).  `amax`: this lane's max |v| over ITS rows; both
// half-waves of a trajectory agree on the exponent.
__device__ __forceinline__ int scale_exponent(float amax) {
  amax = fmaxf(amax, other_half(amax));
  return amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) : 0;
}

// the four k-blocks of a 64-wide cotangent in accumulator layout, scaled by
// 2^-e (returned) and split; consumers multiply their result by 2^e
__device__ __forceinline__ int scaled_split64(const f32x16 (&in)[2], Op16 (&x)[4]) {
  float amax = 0.f;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(in[rb][i]));
  const int e = scale_exponent(amax);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = __builtin_amdgcn_ldexpf(in[kb >> 1][8 * (kb & 1) + j], -e);
    x[kb] = split8(v);
  }
  return e;
}

// out[rb] = W^T-blocks [rb][kb] from `n0` times the split cotangent (scaled)
__device__ __forceinline__ void dense64T_16(f32x16 (&out)[2], const Op16 (&x)[4],
                                            const LdsView16 &L, int base, int n0) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    out[0] = mma3(L.A(base, n0 + kb), x[kb], out[0]);
    out[1] = mma3(L.A(base, n0 + 4 + kb), x[kb], out[1]);
  }
}

// 64-wide layer on accumulator-layout inputs: out[rb] += W[rb] . in, blocks
// [rb][kb] from block index `n0`; `f` is applied to every input first (tanh of
// the previous layer, or the identity) and may store it.
template <typename F>
__device__ __forceinline__ void dense64_16(f32x16 (&out)[2], const f32x16 (&in)[2],
                                           const LdsView16 &L, int base, int n0, F f) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f(kb >> 1, 8 * (kb & 1) + j, in[kb >> 1][8 * (kb & 1) + j]);
    const Op16 x = split8(v);
    out[0] = mma3(L.A(base, n0 + kb), x, out[0]);
    out[1] = mma3(L.A(base, n0 + 4 + kb), x, out[1]);
  }
}

}  // namespace apg
