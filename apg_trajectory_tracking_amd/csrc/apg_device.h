// apg_device.h - shared device helpers for the APG rollout kernels (gfx950).
//
// Data layout (see include/apg.h): one trajectory per lane.  In the native
// SoA layout (batch fastest) every wave-wide access below is one contiguous
// 256-byte transaction; in the reference's AoS layout a lane owns a row and
// reads it with 16-byte loads where the row length allows it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "apg.h"

// Tuning knobs and instrumentation (request schedules, stamps, occupancy
// overrides ...) are compile-time macros so that tools/build_variant.py can
// A/B them.  A PRODUCT build must not carry any of them: one stray -D in the
// environment would ship a different kernel.  Variant builds say so
// explicitly (-DAPG_EXPERIMENT_BUILD); forks that produce wrong results on
// purpose live in tools/patches/, not here.
#if !defined(APG_EXPERIMENT_BUILD) &&                                          \
    (defined(APG_STAMP) || defined(APG_QX) || defined(APG_EXP_NO_STORES) ||    \
     defined(APG_TRIG_1WORD) || defined(APG_SW_TRIG) || defined(APG_MLP_EXP) || \
     defined(APG_ROWS_BLOCK) || defined(APG_ROWS_ACT_PRE) ||                   \
     defined(APG_ROWS_REF_PER_STEP) || defined(APG_ROWS_REF_LOOK) ||           \
     defined(APG_ROWS_ST_AUX) || defined(APG_ROWS_REF_TOP) ||                  \
     defined(APG_ROWS_LD_AUX) || defined(APG_ROWS_STORE_AT_END) || defined(APG_ROWS_STORE_FLUSH_AT) ||                                               \
     defined(APG_REG_ACT_PRE) || defined(APG_REG_REF_PER_STEP) ||              \
     defined(APG_REG_REF_LOOK) || defined(APG_GEMM_ST_MAX) ||                  \
     defined(APG_GEMM_STREAM) || defined(APG_WING_WAVES) ||                    \
     defined(APG_WING_GROUP_PREFETCH) || defined(APG_WING_LITERALS) ||        \
     defined(APG_WING_PK) || defined(APG_WING_PK_KMODE))
#error "experiment macro in a product build (variants: -DAPG_EXPERIMENT_BUILD, tools/build_variant.py)"
#endif

namespace apg {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);

// ---- state that is cached per DEVICE (host) --------------------------------
// hipFuncSetAttribute and the CU count belong to the current device; a process
// may drive more than one GPU, so "done once" / "looked up once" is kept per
// device id (an unknown device is never cached).
constexpr int kMaxDevices = 64;
int device_slot();            // current device id in [0, kMaxDevices), or -1
int device_cu_count();        // CUs of the current device (256 if unknown)
struct PerDeviceOnce {
  bool done[kMaxDevices] = {};
  bool test() const {
    const int d = device_slot();
    return d >= 0 && done[d];
  }
  void set() {
    const int d = device_slot();
    if (d >= 0) done[d] = true;
  }
};

// ---- per-lane loads / stores ----------------------------------------------
// "state-like" tensors: [B][S] (AoS) or [S][B] (SoA)
template <int LAYOUT, int S>
__device__ __forceinline__ void load_state(const float *__restrict__ p, int B,
                                           int b, float (&out)[S]) {
  if constexpr (LAYOUT == APG_LAYOUT_SOA) {
#pragma unroll
    for (int i = 0; i < S; ++i) out[i] = p[(size_t)i * B + b];
  } else if constexpr (S % 4 == 0) {
    const float4 *q = reinterpret_cast<const float4 *>(p + (size_t)b * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i) {
      float4 v = q[i];
      out[4 * i + 0] = v.x, out[4 * i + 1] = v.y;
      out[4 * i + 2] = v.z, out[4 * i + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) out[i] = p[(size_t)b * S + i];
  }
}

template <int LAYOUT, int S>
__device__ __forceinline__ void store_state(float *__restrict__ p, int B, int b,
                                            const float (&v)[S]) {
  if constexpr (LAYOUT == APG_LAYOUT_SOA) {
#pragma unroll
    for (int i = 0; i < S; ++i)
      __builtin_nontemporal_store(v[i], p + (size_t)i * B + b);
  } else if constexpr (S % 4 == 0) {
    float4 *q = reinterpret_cast<float4 *>(p + (size_t)b * S);
#pragma unroll
    for (int i = 0; i < S / 4; ++i)
      q[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) p[(size_t)b * S + i] = v[i];
  }
}

// "sequence-like" tensors: [B][H][C] (AoS) or [H][C][B] (SoA); loads the N
// components starting at column c0 of row (b, k).
template <int LAYOUT, int N>
__device__ __forceinline__ void load_seq(const float *__restrict__ p, int B,
                                         int H, int C, int b, int k, int c0,
                                         float (&out)[N]) {
  if constexpr (LAYOUT == APG_LAYOUT_SOA) {
    const float *q = p + ((size_t)k * C + c0) * B + b;
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = q[(size_t)i * B];
  } else {
    const float *q = p + ((size_t)b * H + k) * C + c0;
    if (N % 4 == 0 && (C % 4) == 0 && (c0 % 4) == 0) {
      const float4 *q4 = reinterpret_cast<const float4 *>(q);
#pragma unroll
      for (int i = 0; i < N / 4; ++i) {
        float4 v = q4[i];
        out[4 * i + 0] = v.x, out[4 * i + 1] = v.y;
        out[4 * i + 2] = v.z, out[4 * i + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) out[i] = q[i];
    }
  }
}

template <int LAYOUT, int N>
__device__ __forceinline__ void store_seq(float *__restrict__ p, int B, int H,
                                          int C, int b, int k, int c0,
                                          const float (&v)[N]) {
  if constexpr (LAYOUT == APG_LAYOUT_SOA) {
    float *q = p + ((size_t)k * C + c0) * B + b;
#pragma unroll
    for (int i = 0; i < N; ++i) __builtin_nontemporal_store(v[i], q + (size_t)i * B);
  } else {
    float *q = p + ((size_t)b * H + k) * C + c0;
    if (N % 4 == 0 && (C % 4) == 0 && (c0 % 4) == 0) {
      float4 *q4 = reinterpret_cast<float4 *>(q);
#pragma unroll
      for (int i = 0; i < N / 4; ++i)
        q4[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) q[i] = v[i];
    }
  }
}

// ---- buffer-addressed SoA planes --------------------------------------------
// A SoA tensor is a stack of planes of B floats.  Addressing it through a
// buffer resource keeps the per-lane part of the address in ONE 32-bit VGPR
// (4*b) and the plane offset in an SGPR: no 64-bit VALU address arithmetic
// and no address register pairs per access.  The descriptor is built from
// kernel arguments only, so it is provably wave-uniform (no waterfall loop).
// Requires planes * B * 4 < 2^31 (checked on the host; larger tensors take
// the flat-address path).
struct SoaPlanes {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;  // 4 * trajectory index
  int pitch; // 4 * B
  __device__ __forceinline__ SoaPlanes(const void *base, int planes, int B, int b)
      : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0,
                                               planes * B * 4, 0x00020000)),
        voff(b * 4), pitch(B * 4) {}
  __device__ __forceinline__ float ld(int plane) const {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, plane * pitch, 0));
  }
  // outputs are written once and never re-read by this launch: `nt`
  // (aux bit 1) streams them out during the kernel instead of leaving dirty
  // L2 lines for the end-of-kernel write-back (tools/hbm_probe.hip: 8.6 ->
  // 6.9 us per launch for the rollout-shaped stream at B = 65 536)
  __device__ __forceinline__ void st(int plane, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc,
                                          voff, plane * pitch, 2);
  }
};

// ---- trigonometry ----------------------------------------------------------
// Branch-free sin/cos pair: 3-term Cody-Waite reduction by pi/2 (exact
// products through FMA) + degree-9 / degree-10 polynomials on [-pi/4, pi/4].
// Max error 1.6 ulp for |x| <= 1e5 (checked against float64 on 1e7 points,
// see DESIGN.md); beyond ~1e6 rad accuracy degrades gracefully, NaN/Inf give
// NaN.  ~30 VALU ops and no control flow, so the three attitude angles of a
// step interleave in the pipeline (libm's sincosf costs ~45 issued ops per
// call plus branches that serialise them).  Host-callable as well: tests/
// compiles this arithmetic for the CPU (tests/host_math).
__host__ __device__ __forceinline__ void sincos_fast(float x, float *sn, float *cs) {
  const float kf = rintf(x * 0.6366197466850281f);
  const int k = (int)kf;
  float r = fmaf(-kf, 1.5707963705062866f, x);
  r = fmaf(-kf, -4.371138828673793e-08f, r);
  r = fmaf(-kf, -1.7151245100058819e-15f, r);
  const float t = r * r;
  float ps = fmaf(t, 2.6658919978217455e-06f, -0.0001983463589567691f);
  ps = fmaf(t, ps, 0.008333319798111916f);
  ps = fmaf(t, ps, -0.1666666716337204f);
  const float s = fmaf(r * t, ps, r);
  float pc = fmaf(t, -4.336599204179947e-07f, 2.494495674909558e-05f);
  pc = fmaf(t, pc, -0.0013889188412576914f);
  pc = fmaf(t, pc, 0.0416666679084301f);
  const float c = fmaf(t * t, pc, fmaf(t, -0.5f, 1.0f));
  const bool swap = (k & 1) != 0;
  const float so = swap ? c : s, co = swap ? s : c;
  *sn = __builtin_bit_cast(
      float, __builtin_bit_cast(unsigned, so) ^ ((unsigned)(k & 2) << 30));
  *cs = __builtin_bit_cast(
      float, __builtin_bit_cast(unsigned, co) ^ ((unsigned)((k + 1) & 2) << 30));
}

// Hardware pair: v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS and
// reduce it themselves (valid for |u| <= 256), so the work left to the VALU
// is x / 2pi to better than one fp32 rounding: the product with 1/2pi as a
// two-word constant, split as fract(hi) + lo so that many revolutions cost no
// precision (5 plain + 2 quarter-rate instructions instead of ~30; absolute
// error of the results ~1.5e-7).  A single rounded product (APG_TRIG_1WORD,
// 3 instructions fewer per angle) leaves a phase error of 1.2e-7 |x| rad:
// fine for attitudes within a few revolutions, but it fails the 1e-4 parity
// bar at |x| ~ 1e4 rad (tests/test_gpu_parity.py::test_quad_large_angles_and_
// rates) and measured no faster on the bench kernel (profiles/r02_ab_trig.json).
// Host builds (tests only) use libm.
__host__ __device__ __forceinline__ void sincos_hw(float x, float *sn, float *cs) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float kHi = 0.15915494f;          // fl(1 / 2pi)
  const float u = x * kHi;
#if defined(APG_TRIG_1WORD)
  const float f = __builtin_amdgcn_fractf(u);
#else
  const float kLo = 6.4206382e-09f;       // 1 / 2pi - kHi
  float e = fmaf(x, kHi, -u);             // exact low word of the product
  e = fmaf(x, kLo, e);
  const float f = __builtin_amdgcn_fractf(u) + e;
#endif
  *sn = __builtin_amdgcn_sinf(f);
  *cs = __builtin_amdgcn_cosf(f);
#else
  *sn = (float)sin((double)x);
  *cs = (float)cos((double)x);
#endif
}

// 1/x: hardware v_rcp_f32 (1 ulp) + one Newton step (~0.5 ulp) in 3 VALU ops
// instead of the ~10-op IEEE division sequence.  (Host builds - tests only -
// start the same Newton step from the correctly rounded quotient.)
__host__ __device__ __forceinline__ float rcp_nr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(x);
#else
  const float r = 1.0f / x;
#endif
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// rcp_nr that stays usable at x = 0 (+-FLT_MAX instead of inf / NaN: the
// Newton step of rcp_nr turns 1 / 0 into 0 * inf).  Identical bits for every
// x whose reciprocal is finite; used where a mask multiplies the result
// (0 * FLT_MAX = 0, 0 * inf = NaN).  Two v_med3_f32 more than rcp_nr.
__host__ __device__ __forceinline__ float rcp_nr_finite(float x) {
  const float kMax = 3.402823466e+38f;
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(x), -kMax, kMax);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return __builtin_amdgcn_fmed3f(r, -kMax, kMax);
#else
  float r = fminf(fmaxf(1.0f / x, -kMax), kMax);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return fminf(fmaxf(r, -kMax), kMax);
#endif
}

// hardware v_sqrt_f32 (1 ulp); sqrtf on the host
__host__ __device__ __forceinline__ float sqrt_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}

// Two values per lane (the fixed-wing kernels' two-trajectories-per-lane
// form, wing_math.h): component-wise forms of the helpers above, in THIS
// namespace so that overload resolution sees them next to the scalar ones.
typedef float fx2 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ fx2 rcp_nr(fx2 x) {
  return (fx2){rcp_nr(x.x), rcp_nr(x.y)};
}
__host__ __device__ __forceinline__ fx2 rcp_nr_finite(fx2 x) {
  return (fx2){rcp_nr_finite(x.x), rcp_nr_finite(x.y)};
}
__host__ __device__ __forceinline__ fx2 sqrt_fast(fx2 x) {
  return (fx2){sqrt_fast(x.x), sqrt_fast(x.y)};
}

// ---- wave64 reduction -------------------------------------------------------
// Butterfly over the 64 lanes; every lane ends with the full sum, in an order
// that depends only on the lane index (deterministic).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// One loss partial per WAVE, indexed by the global wave number.  `count` =
// number of partials the buffer holds (apg_loss_partials_count(B) =
// ceil(B / 64)): with workgroups of more than one wave (-DAPG_ROLLOUT_BLOCK=128
// / 256 builds) the fully dead tail waves of the last workgroup stay
// convergent for the reductions and reach this point - they must not write.
__device__ __forceinline__ void write_wave_partial(float *partials, float lane_loss,
                                                   int count = 0x7fffffff) {
  float s = wave_sum(lane_loss);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (wave < count) partials[wave] = s;
  }
}

// Deferred reduction (ApgDeferredLoss): executed by ONE wave of the launch,
// split in two so that it costs that wave no stall: `head` issues the loads
// of the earlier launch's partials BEFORE the wave's own input loads (they
// return first, in order), `tail` sums them at the very end of the kernel,
// long after they have landed.  Lane l owns partials l, l+64, ...; the
// summation order is fixed by the indices (deterministic).
struct PrevPartials {
  float v[16];
  double acc;
};

__device__ __forceinline__ void reduce_prev_head(const ApgDeferredLoss &d,
                                                 PrevPartials &pp) {
  const int lane = threadIdx.x & (kWave - 1);
  pp.acc = 0.0;
  // everything beyond the first 1024 partials (batches > 65 536) is summed
  // right away, 16 independent loads per lane and pass
  for (int base = 16 * kWave; base < d.prev_count; base += 16 * kWave) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = base + j * kWave + lane;
      t[j] = i < d.prev_count ? d.prev_partials[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) pp.acc += (double)t[j];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = j * kWave + lane;
    pp.v[j] = i < d.prev_count ? d.prev_partials[i] : 0.f;
  }
}

__device__ __forceinline__ void reduce_prev_tail(const ApgDeferredLoss &d,
                                                 const PrevPartials &pp) {
  double acc = pp.acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc += (double)pp.v[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) d.prev_loss[0] = (float)acc;
}

__device__ __forceinline__ void reduce_prev_partials(const ApgDeferredLoss &d) {
  PrevPartials pp;
  reduce_prev_head(d, pp);
  reduce_prev_tail(d, pp);
}

// Second stage: fixed-order sum of the per-wave partials (one workgroup).
int launch_reduce_partials(const float *partials, int n, float *loss,
                           hipStream_t stream);

}  // namespace apg
