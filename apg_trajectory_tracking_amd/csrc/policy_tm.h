// policy_tm.h - trajectory-major weight products inside the reverse sweeps
// (mlp_concurrent.hip / mlp_rollout.hip: the concurrent and the autoregressive step).
//
// dW[m][k] = sum_n delta[m][n] x[k][n] is a matrix product whose reduction index
// is the trajectory.  The sweeps hold trajectories in the lane; the matrix
// instruction with its operands swapped yields the cotangent with the
// trajectory in the REGISTERS (16 per lane, the feature in the lane), x comes
// from the forward sweep's planes as four 16-byte loads per lane, and every wave
// adds the 32 x 32 blocks of its own 32 trajectories into 32-bit FIXED-POINT
// accumulators in LDS (ds_add_u32: order-free, bit-reproducible).  This header
// holds what the three kernels share: the block loads and splits, the
// fixed-point additions and flushes, the exponent exchange, the transpositions
// by identity products.  (rounds 4 and 5.)
#pragma once
#include "policy_mfma16.h"

namespace apg {

constexpr int kTmThreads = 512;   // one workgroup = 8 waves = 256 trajectories

__device__ __forceinline__ float wave_fmax(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
  return v;
}

constexpr int kFix = 22, kFixConv = 19;
// The unit is folded into the operand scales (2^11 x 2^11, conv 2^10 x 2^9:
// |operand| <= 2 048, far inside fp16), so a block element leaves the matrix
// pipe already in accumulator units: the conversion is a rounding, no scaling.
constexpr int kPreD = 11, kPreX = kFix - kPreD, kPreDc = 10, kPreXc = kFixConv - kPreDc;
// Maxima are taken on the BIT PATTERNS of |v| (unsigned): finite values order as
// they do as floats, inf and every NaN lie above them - a non-finite cotangent or
// x is seen (a float max would drop a NaN, the integer conversion turn it into
// 0) and the workgroup's gradient blocks are written as NaN from there on.
constexpr unsigned kInfBits = 0x7f800000u;
__device__ __forceinline__ unsigned umax_abs(unsigned m, float v) {
  const unsigned b = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
  return b > m ? b : m;
}
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, s, 64);
    v = o > v ? o : v;
  }
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// exponent e with 2^e above the value whose bits are `a` (0 for zero; a
// non-finite value sets `bad`)
__device__ __forceinline__ int bits_exp(unsigned a, bool &bad, bool nonneg_floor) {
  if (a >= kInfBits) {
    bad = true;
    return 0;
  }
  const int e = a ? __builtin_amdgcn_frexp_expf(__builtin_bit_cast(float, a)) : 0;
  return nonneg_floor && e < 0 ? 0 : e;
}

typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef int i32x4_ __attribute__((ext_vector_type(4)));

#if !defined(APG_EXPERIMENT_BUILD) && defined(APG_AR_KNOCKOUT)
#error "experiment macro in a product build (variants: tools/build_policy_variant.sh)"
#endif
#ifndef APG_AR_KNOCKOUT
#define APG_AR_KNOCKOUT 0   // timing experiments (recurrent sweeps): 1 no global atomics,
                            // 2 no weight-block products, 4 no LDS adds, 8 no conv-weight
                            // products, 16 no workgroup barriers, 32 the trajectory-major
                            // block loads read one cache-resident 2 KB window, 128 the same
                            // loads lane-linear (the block's bytes, coalesced), 64 the
                            // per-lane state / action / reference loads read cached planes
#endif

// 32 planes from `soff` (scalar: first plane x pitch + the wave's first
// trajectory), one per lane & 31, trajectory-major: v[4 g + c] = trajectory
// c + 8 g + 4 hi of the wave - the trajectory set of accumulator register 4 g + c
struct TBlock {
  u32x4 q[4];
  __device__ __forceinline__ void load(const Planes &X, unsigned voff, unsigned soff) {
    if (APG_AR_KNOCKOUT & 32) {
      voff = (threadIdx.x & 31u) * 64u + ((threadIdx.x & 32u) ? 16u : 0u);
      soff = 0u;
    }
    if (APG_AR_KNOCKOUT & 128) {   // lane-linear: 4 KB from the block's first plane on -
      // the same bytes per block, 8 cache lines per instruction instead of 64 (what a
      // tiled plane layout would give the sweep)
      voff = voff == kDead ? kDead : (threadIdx.x & 63u) * 16u;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        q[g] = __builtin_amdgcn_raw_buffer_load_b128(X.rsrc, (int)voff, (int)(soff + 1024 * g),
                                                     APG_PLANES_LD_AUX);
      return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      q[g] = __builtin_amdgcn_raw_buffer_load_b128(X.rsrc, (int)voff, (int)(soff + 32 * g),
                                                   APG_PLANES_LD_AUX);
  }
  __device__ __forceinline__ void get(float (&v)[16]) const {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_ f = __builtin_bit_cast(f32x4_, q[g]);   // (whole vector: see wgrad_block)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[4 * g + c] = f[c];
    }
  }
};

// the two k-blocks (8 trajectories per half-wave each) of 16 trajectory-major
// values x 2^-e
__device__ __forceinline__ void split16(const float (&v)[16], int e, Op16 (&o)[2]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    float w8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w8[j] = __builtin_amdgcn_ldexpf(v[8 * kk + j], -e);
    o[kk] = split8(w8);
  }
}

#if !defined(APG_EXPERIMENT_BUILD) && (defined(APG_TM_KNOCKOUT) || defined(APG_TM_PLAIN_PARTIALS))
#error "experiment macro in a product build (variants: tools/build_policy_variant.sh)"
#endif
#ifndef APG_TM_KNOCKOUT
#define APG_TM_KNOCKOUT 0   // timing experiments: 1 float atomics (ds_add_f32) on the same data
#endif
// v (in units of the accumulator's scale) into the fixed-point accumulator at p
__device__ __forceinline__ void lds_add(char *p, float v, int fix = kFix) {
  if (APG_TM_KNOCKOUT & 1) {
    __hip_atomic_fetch_add(reinterpret_cast<float *>(p), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  (void)fix;   // (v arrives in units of 2^-fix: see kPreD)
  const int q = (int)__builtin_rintf(v);
  __hip_atomic_fetch_add(reinterpret_cast<int *>(p), q, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}

// two-limb addition (conv block; states_in's blocks, whose exponent is a bound as
// well): v = hi 2^-fix + lo 2^-2 fix + O(2^-2 fix - 1)
__device__ __forceinline__ void lds_add2(char *hi, char *lo, float v, int fix = kFixConv) {
  const float s_ = v, qh = __builtin_rintf(s_);   // (v in units of 2^-fix already)
  __hip_atomic_fetch_add(reinterpret_cast<int *>(hi), (int)qh, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
  __hip_atomic_fetch_add(reinterpret_cast<int *>(lo),
                         (int)__builtin_rintf(__builtin_amdgcn_ldexpf(s_ - qh, fix)),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// acc (a 32 x 32 block in accumulator layout, scaled operands) into the LDS block
__device__ __forceinline__ void add_block(char *blk_lane, const f32x16 &acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) lds_add(blk_lane + i * 256, acc[i]);
}

// workgroup: `floats` fixed-point accumulators at `off` -> dst as floats x 2^e
// (e = the product of the operand scales), optionally zeroed for the next user
__device__ __forceinline__ void flush_region(char *lds, int off, int floats, float *dst, int e,
                                             bool rezero, bool bad, int fix = kFix) {
  const i32x4_ z = {0, 0, 0, 0};
  for (int idx = threadIdx.x; idx < floats / 4; idx += kTmThreads) {
    i32x4_ *p = reinterpret_cast<i32x4_ *>(lds + off) + idx;
    const i32x4_ q = *p;
    f32x4_ v;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      v[c] = bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)q[c], e - fix);
#ifdef APG_TM_PLAIN_PARTIALS     /* experiment builds: cached stores of the partials -
                                    second stage -1.1 us, this kernel +2.3: not shipped */
    reinterpret_cast<f32x4_ *>(dst)[idx] = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_ *>(dst) + idx);
#endif
    if (rezero) *p = z;
  }
}

__device__ __forceinline__ void zero_region(char *lds, int off, int bytes) {
  const f32x4_ z = {0.f, 0.f, 0.f, 0.f};
  for (int idx = threadIdx.x; idx < bytes / 16; idx += kTmThreads)
    reinterpret_cast<f32x4_ *>(lds + off)[idx] = z;
}

// the workgroup's exponent from the eight waves' slots
__device__ __forceinline__ int wg_exp(const unsigned (&slots)[8], bool &bad) {
  unsigned a = slots[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) a = slots[w] > a ? slots[w] : a;
  return bits_exp(a, bad, false);
}

// A trajectory-major block of an UNBOUNDED plane group with every value clamped
// to the group's scale 2^f (live columns are inside it by construction: no-op).
// The columns beyond the batch are somebody else's - another step's or another
// trajectory's - values: finite, but not bound by THIS workgroup's maxima; they
// meet a zero cotangent and must not become an overflowed fp16 operand on the
// way.  (tanh planes need nothing: every column is in [-1, 1].)
__device__ __forceinline__ void get_clamped(const TBlock &t, float (&v)[16], int f) {
  t.get(v);
  const float lim = __builtin_amdgcn_ldexpf(1.f, f);
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], -lim, lim);
}

// the per-trajectory exponents `ex` (lane = trajectory) in accumulator layout of
// a swapped product: E[i] = ex of trajectory r(i) + 4 hi - one matrix
// instruction, D[trajectory][feature] = ex[trajectory] x 1 (k-slot 0 only;
// exponents are small integers: exact in fp16)
__device__ __forceinline__ void texp(int ex, int hi, int (&E)[16]) {
  const _Float16 hx = (_Float16)(float)ex;
  u32x4 a = {0u, 0u, 0u, 0u}, o = {0u, 0u, 0u, 0u};
  a[0] = hi ? 0u : (unsigned)__builtin_bit_cast(unsigned short, hx);
  o[0] = hi ? 0u : 0x3c00u;
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  z = mfma16(a, o, z);
#pragma unroll
  for (int i = 0; i < 16; ++i) E[i] = (int)z[i];
}

// The workgroup's accumulators in global memory through a buffer resource: the
// per-thread part of an address is ONE VGPR (4 threadIdx.x), the block offset a
// scalar - no 64-bit address pairs per flush site.  One element += v, no return
// value (buffer_atomic_add_f32: executed at the L2); this thread owns the element
// for the whole sweep, the steps add in order.
__device__ __forceinline__ void gadd(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff,
                                     float v) {
  if (APG_AR_KNOCKOUT & 1) return;
  __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, (int)voff, (int)soff, 0);
}

// workgroup: N fixed-point accumulators at `off` -> += the floats from byte `dst`
// of the partial buffer (x 2^(e - fix)), re-zeroed for their next user
template <int N>
__device__ __forceinline__ void flush_add(char *lds, int off, __amdgpu_buffer_rsrc_t r,
                                          unsigned dst, int e, bool bad, int fix = kFix) {
  static_assert(N % kTmThreads == 0 || N < kTmThreads, "whole rounds of the workgroup");
  if (N < kTmThreads && (int)threadIdx.x >= N) return;   // (whole waves: N = 64, 256)
  int *p = reinterpret_cast<int *>(lds + off) + threadIdx.x;
#pragma unroll
  for (int m = 0; m < (N < kTmThreads ? 1 : N / kTmThreads); ++m) {
    const int q = p[m * kTmThreads];
    p[m * kTmThreads] = 0;
    gadd(r, threadIdx.x * 4u, dst + (unsigned)m * kTmThreads * 4u,
         bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)q, e - fix));
  }
}

// B operand of the identity product that brings a trajectory-major block (an A
// operand as split16 made it: k-slot j of k-block kk, half hi = trajectory
// (j & 3) + 8 (j >> 2) + 16 kk + 4 hi) into accumulator layout, trajectory in
// the lane: slot (kk, hi, j) of column n is 1 where that trajectory IS n.  From
// the lane index: one half-word of the lane's eight is set.
__device__ __forceinline__ void ident_operands(int lane_o, u32x4 (&I)[2]) {
  const int n = lane_o & 31, hi = lane_o >> 5, m = n - 4 * hi;
  const bool valid = m >= 0 && (m & 4) == 0;
  const int kk = (m >> 4) & 1, q = ((m >> 3) & 1) * 2 + ((m >> 1) & 1);
  const unsigned one = (m & 1) ? 0x3c000000u : 0x3c00u;
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) I[k2][q2] = (valid && kk == k2 && q == q2) ? one : 0u;
}

// 32 planes x the wave's 32 trajectories from their trajectory-major split (two
// k-blocks, scaled by 2^s) to accumulator layout (feature r(i) + 4 hi of the
// lane's trajectory, x 2^s): four matrix instructions instead of 32 loads per
// lane of bytes the wave has just read in the other orientation
__device__ __forceinline__ f32x16 to_feature_major(const Op16 (&bx)[2], const u32x4 (&I)[2]) {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    z = mfma16(bx[kk].l, I[kk], z);
    z = mfma16(bx[kk].h, I[kk], z);
  }
  return z;
}

}  // namespace apg
