// policy_mfma.h - building blocks shared by the in-kernel policies (mlp_rollout.hip, mlp_concurrent.hip,
// lstm.hip, mlp_wing.hip): one wave = 32 trajectories, the layers on 32 x 32
// matrix-core tiles.
//
// For D = A B + C with A = weights [32 outputs x k], B = activations
// [k x 32 trajectories], lane l works for trajectory l & 31 (half-wave
// hi = l >> 5) and the accumulator C / D puts row r(i) + 4 hi, column l & 31
// into register i, r(i) = (i & 3) + 8 (i >> 2).  An accumulator register of a
// layer's output is therefore directly an input element of the next layer for
// the same trajectory - layers chain with no shuffles.  Rounds 1-2 multiplied
// with v_mfma_f32_32x32x2_f32; since round 3 the operands are split into fp16
// terms for v_mfma_f32_32x32x16_f16 (policy_mfma16.h, which also defines the
// operand order).  This header keeps what both share: the accumulator layout,
// LDS / plane access, tanh.
#pragma once
#include "apg_device.h"

namespace apg {

constexpr unsigned kDead = 0xFFFFFFFCu;  // buffer offset beyond any tensor

// cache policy of the plane stores / loads (tuning knobs: experiment builds)
#if !defined(APG_EXPERIMENT_BUILD) &&                                          \
    (defined(APG_PLANES_ST_AUX) || defined(APG_PLANES_LD_AUX))
#error "experiment macro in a product build (variants: -DAPG_EXPERIMENT_BUILD, tools/build_policy_variant.sh)"
#endif
#ifndef APG_PLANES_ST_AUX
#define APG_PLANES_ST_AUX 2
#endif
#ifndef APG_PLANES_LD_AUX
#define APG_PLANES_LD_AUX 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int rrow(int i) { return (i & 3) + 8 * (i >> 2); }
// tanh(x) = 1 - 2 / (1 + e^(2x)): one v_exp and one v_rcp, no select.  Exact
// limits (e -> inf: 1, e -> 0: -1); absolute error <= 2e-7 everywhere (near 0
// the subtraction cancels, so the RELATIVE error of tiny outputs is larger -
// irrelevant for activations that are compared, and used, by absolute value).
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // 2 / ln 2
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

// relu of a matrix-pipe result in ONE instruction: the signed-integer maximum of
// the bit pattern and 0 (negative floats are negative integers).  fmaxf and
// fmed3 canonicalise their input first (a second v_max per value); inline
// assembly is no option: the compiler's hazard recognizer does not see a
// matrix-pipe result read by an asm statement and leaves out the wait states
// (measured: stale accumulators).
__device__ __forceinline__ float relu1(float v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

__device__ __forceinline__ float sigmoidf_(float x) {
  return 1.0f / (1.0f + expf(-x));
}

// the value of the lane 32 away.  (round 6: v_permlane32_swap + a select on the
// VALU instead of this ds_bpermute measured no faster - LSTM sweeps 107-109 /
// 116-121 us against 105-106 / 117 on the same box, tools/ab_gate_wgrad.sh.)
__device__ __forceinline__ float other_half(float v) {
  return __shfl_xor(v, 32, 64);
}

// LDS reads with compile-time table offsets.  ds_read has a 16-bit immediate
// byte offset, the tables may span > 64 KB: opaque per-lane bases (zero,
// half-wave index, lane, lane + kSplit) keep every access "base VGPR +
// immediate"; without them the compiler materialises one address VGPR per
// distinct offset, hoists all of them out of the step loop and spills them.
constexpr int kSplit = 255 * 64;
struct LdsView {
  const float *lds;
  int o_0, o_hi, o_l0, o_l1;
  __device__ __forceinline__ LdsView(const float *l, int lane) : lds(l) {
    o_0 = 0, o_hi = lane >> 5, o_l0 = lane, o_l1 = lane + kSplit;
    asm volatile("" : "+v"(o_0), "+v"(o_hi), "+v"(o_l0), "+v"(o_l1));
  }
  // wave-uniform entry [off]
  __device__ __forceinline__ float U(int off) const { return lds[o_0 + off]; }
  // table entry [off + hi] (off even: [..][2] tables)
  __device__ __forceinline__ float T(int off) const { return lds[o_hi + off]; }
  // A operand [off + lane]
  __device__ __forceinline__ float A(int off) const {
    return off < kSplit ? lds[o_l0 + off] : lds[o_l1 + (off - kSplit)];
  }
};

// Plane-addressed global memory through a buffer resource: the per-lane part
// of the address is ONE 32-bit VGPR, the plane offset a scalar - no 64-bit
// address pairs per access (the kernels touch hundreds of planes per step).
// Lanes that must not touch memory get the offset kDead: their loads return
// 0 and their stores are dropped by the range check - no branches.
struct Planes {
  __amdgpu_buffer_rsrc_t rsrc;
  __device__ __forceinline__ Planes(const void *base, unsigned planes, unsigned pitch)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                               (int)(planes * pitch), 0x00020000)) {}
  // voff: per-lane byte offset (VGPR), soff: plane * pitch (scalar)
  __device__ __forceinline__ float ld(unsigned voff, unsigned soff) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                         rsrc, (int)voff, (int)soff, APG_PLANES_LD_AUX));
  }
  __device__ __forceinline__ unsigned ldu(unsigned voff, unsigned soff) const {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff,
                                                APG_PLANES_LD_AUX);
  }
  __device__ __forceinline__ void st(unsigned voff, unsigned soff, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc,
                                          (int)voff, (int)soff, APG_PLANES_ST_AUX);
  }
  __device__ __forceinline__ void stu(unsigned voff, unsigned soff, unsigned v) const {
    __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)voff, (int)soff,
                                          APG_PLANES_ST_AUX);
  }
};

// The plane pitch as the loop body sees it: opaque per iteration, so that the
// `plane * pitch` scalar offsets of a step are computed where they are used
// (one s_mul each on the idle SALU) instead of being hoisted out of the step
// loop into several hundred live SGPRs.
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Workgroup copy of the packed operand tables into LDS: direct-to-LDS DMA, one
// wave instruction per KiB (64 lanes x 16 B), every wave issues its share
// back to back and waits once.  (Rounds 1-3 copied through registers in a loop
// the compiler does not unroll - 16 dependent load -> ds_write round trips per
// thread before the first instruction of the sweep: 2.5-3 us per launch,
// profiles/r03_gemm_stream_bf16x3.txt.)
// the same without the wait and the barrier: the caller has more to do first
__device__ __forceinline__ void fill_lds_issue(float *lds, const float *src, int floats) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0,
                                                   (unsigned)floats * 4u, 0x00020000);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  for (int c = wave * 256; c < floats; c += waves * 256)
    if (c + lane * 4 < floats)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + c), 16,
                                               (c + lane * 4) * 4, 0, 0, 0);
}

__device__ __forceinline__ void fill_lds(float *lds, const float *src, int floats) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0,
                                                   (unsigned)floats * 4u, 0x00020000);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  for (int c = wave * 256; c < floats; c += waves * 256)
    if (c + lane * 4 < floats)   // (a last, partial KiB: the other lanes stay out)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + c), 16,
                                               (c + lane * 4) * 4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Minibatch rows read THROUGH THE INDEX by the kernel that consumes them (round 5;
// TrainBase.run_epoch's batch selection, scripts/train_base.py:191-194, without a
// gather pass): the workgroup's 256 trajectories' rows of one [N][ld] data-set
// tensor land in LDS as [256][P] (P odd: a lane per trajectory then reads without
// bank conflicts) by direct-to-LDS loads - one wave instruction moves 64
// consecutive floats of that image, i.e. pieces of at most two source rows: a few
// cache lines per instruction, where a lane-per-trajectory load touches 64.
// rows: the workgroup's 256 source row numbers (LDS); R <= P columns are read,
// the pad columns get zeros (offset kDead).
template <int P>
__device__ __forceinline__ void gather_rows_issue(float *dst, const int *rows,
                                                  const float *base, unsigned bytes, int ld,
                                                  int R) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes,
                                                   0x00020000);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int waves = blockDim.x >> 6;
  for (int n = wave; n < 4 * P; n += waves) {      // 256 P / 64 instructions
    const int e = n * 64 + lane, t = e / P, j = e - t * P;
    const unsigned voff =
        j < R ? ((unsigned)rows[t] * (unsigned)ld + (unsigned)j) * 4u : kDead;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + n * 64), 4, (int)voff, 0, 0,
                                             0);
  }
}

}  // namespace apg
