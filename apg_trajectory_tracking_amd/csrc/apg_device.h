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

namespace apg {

constexpr int kWave = 64;  // CDNA wavefront

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);

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
    for (int i = 0; i < S; ++i) p[(size_t)i * B + b] = v[i];
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
    for (int i = 0; i < N; ++i) q[(size_t)i * B] = v[i];
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

// ---- wave64 reduction -------------------------------------------------------
// Butterfly over the 64 lanes; every lane ends with the full sum, in an order
// that depends only on the lane index (deterministic).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// One loss partial per wave, indexed by the global wave number.
__device__ __forceinline__ void write_wave_partial(float *partials, float lane_loss) {
  float s = wave_sum(lane_loss);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    partials[wave] = s;
  }
}

// Second stage: fixed-order sum of the per-wave partials (one workgroup).
int launch_reduce_partials(const float *partials, int n, float *loss,
                           hipStream_t stream);

}  // namespace apg
