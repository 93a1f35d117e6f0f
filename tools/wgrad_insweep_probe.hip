// wgrad_insweep_probe.hip - what would it cost to accumulate a layer's weight
// gradient INSIDE the reverse sweep instead of writing the cotangent planes for
// a second pass ("accumulate the weight gradients inside the
// reverse sweep")?
//
// dW[m][k] = sum over trajectories n of delta[m][n] x[k][n].  In the sweeps a
// wave holds delta and x in MFMA accumulator layout (rows in registers, the
// trajectory in the lane); the matrix instruction reduces over k-slots, so both
// operands have to be transposed through LDS: every lane writes its values as
// fp16 high / low terms to [term][row][trajectory] and reads back 16 B = the 8
// trajectories 8 hi + j of ITS row - the k-slots of v_mfma_f32_32x32x16_f16.
// delta has no bounded range: it is scaled by a power of two per wave (largest
// entry into [0.5, 1)), the product goes to a zeroed temporary and is added to
// the accumulator with the inverse scale.
//
// One 64 x 64 layer, B = 65 536 trajectories, 4 waves per workgroup, one
// workgroup per CU (the sweeps' tables leave 32-48 KB of LDS):
//   mode 0  what is shipped: store the 64 cotangent rows as planes (the product
//           pass then reads them and the 64 activation rows again)
//   mode 1  "private": every wave accumulates the whole 64 x 64 gradient of its
//           own 32 trajectories (64 accumulator registers per layer), staging in
//           wave-private LDS (20 KB per wave), no barrier
//   mode 2  "shared": wave w owns block (w >> 1, w & 1) of the gradient and
//           multiplies all four waves' staged operands (16 accumulator registers
//           per layer), two barriers per layer
// Prints the time per layer for all 65 536 trajectories and the error of the
// accumulated gradient against a double-precision host product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/wgrad_insweep_probe tools/wgrad_insweep_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                       \
    }                                                                \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int rrow(int i) { return (i & 3) + 8 * (i >> 2); }

constexpr int kThreads = 256, kWaves = kThreads / 64;
constexpr int kRowB = 80;                 // bytes per staged row: 32 fp16 + pad (conflict-free b128)
constexpr int kTermB = 64 * kRowB;        // one term of one operand
constexpr int kStageB = 4 * kTermB;       // delta h, delta l, x h, x l = 20 480 B
constexpr int kLdsB = 147456;             // 144 KB: one workgroup per CU, as in the sweeps

struct Args {
  const float *d, *x;   // [64][B]
  float *planes;        // mode 0: [64][B]
  float *partial;       // modes 1, 2: [workgroup][wave][64 x 64 or block]
  int B, iters;
};

__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// wave-uniform exponent that brings the wave's largest |v| into [0.5, 1)
__device__ __forceinline__ int wave_exponent(const f32x16 (&v)[2]) {
  float amax = 0.f;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(v[rb][i]));
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s));
  return amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) : 0;
}

// this lane's 32 values of `v` (scaled by 2^-e) as fp16 terms into the stage
__device__ __forceinline__ void stage(char *st, const f32x16 (&v)[2], int e, int lane) {
  char *p = st + (lane >> 5) * 4 * kRowB + (lane & 31) * 2;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float s = __builtin_amdgcn_ldexpf(v[rb][i], -e);
      const _Float16 h = (_Float16)s;
      const _Float16 l = (_Float16)(s - (float)h);
      *reinterpret_cast<_Float16 *>(p + (32 * rb + rrow(i)) * kRowB) = h;
      *reinterpret_cast<_Float16 *>(p + kTermB + (32 * rb + rrow(i)) * kRowB) = l;
    }
}

// tmp = block (mb, kb) of the staged operands' product over the 32 trajectories
__device__ __forceinline__ f32x16 block_product(const char *st, int mb, int kb, int lane) {
  const char *pd = st + (32 * mb + (lane & 31)) * kRowB + (lane >> 5) * 16;
  const char *px = st + 2 * kTermB + (32 * kb + (lane & 31)) * kRowB + (lane >> 5) * 16;
  f32x16 t;
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const u32x4 dh = *reinterpret_cast<const u32x4 *>(pd + 32 * kh);
    const u32x4 dl = *reinterpret_cast<const u32x4 *>(pd + kTermB + 32 * kh);
    const u32x4 xh = *reinterpret_cast<const u32x4 *>(px + 32 * kh);
    const u32x4 xl = *reinterpret_cast<const u32x4 *>(px + kTermB + 32 * kh);
    t = mfma16(dl, xh, t);
    t = mfma16(dh, xl, t);
    t = mfma16(dh, xh, t);
  }
  return t;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void probe_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, hi = lane >> 5, wave = threadIdx.x >> 6;
  const int b = (blockIdx.x * kWaves + wave) * 32 + (lane & 31);
  f32x16 d[2], x[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      d[rb][i] = A.d[(size_t)(rb * 32 + rrow(i) + 4 * hi) * A.B + b];
      x[rb][i] = A.x[(size_t)(rb * 32 + rrow(i) + 4 * hi) * A.B + b];
    }
  char *mine = lds + wave * kStageB;
  int *exps = reinterpret_cast<int *>(lds + kWaves * kStageB);
  f32x16 acc[MODE == 1 ? 4 : 1];
#pragma unroll
  for (int q = 0; q < (MODE == 1 ? 4 : 1); ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
#pragma unroll 1
  for (int it = 0; it < A.iters; ++it) {
    const float c = 1.f + (float)it * 0.015625f;   // a different cotangent every layer
    const float sg = (it & 1) ? -1.f : 1.f;        // ... and a different activation
    f32x16 dc[2], xc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) dc[rb] = d[rb] * c, xc[rb] = x[rb] * sg;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          __builtin_nontemporal_store(dc[rb][i],
                                      A.planes + (size_t)(rb * 32 + rrow(i) + 4 * hi) * A.B + b);
    } else {
      const int e = wave_exponent(dc);
      stage(mine, dc, e, lane);
      stage(mine + 2 * kTermB, xc, 0, lane);
      if constexpr (MODE == 1) {
        __builtin_amdgcn_wave_barrier();
        const float back = __builtin_amdgcn_ldexpf(1.f, e);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16 t = block_product(mine, q >> 1, q & 1, lane);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[q][i] = fmaf(t[i], back, acc[q][i]);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        if (lane == 0) exps[wave] = e;
        __syncthreads();
#pragma unroll
        for (int src = 0; src < kWaves; ++src) {
          const f32x16 t = block_product(lds + src * kStageB, wave >> 1, wave & 1, lane);
          const float back = __builtin_amdgcn_ldexpf(1.f, exps[src]);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[0][i] = fmaf(t[i], back, acc[0][i]);
        }
        __syncthreads();
      }
    }
  }
  if constexpr (MODE == 1) {
    float *out = A.partial + (size_t)(blockIdx.x * kWaves + wave) * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        out[(32 * (q >> 1) + rrow(i) + 4 * hi) * 64 + 32 * (q & 1) + (lane & 31)] = acc[q][i];
  } else if constexpr (MODE == 2) {
    float *out = A.partial + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      out[(32 * (wave >> 1) + rrow(i) + 4 * hi) * 64 + 32 * (wave & 1) + (lane & 31)] = acc[0][i];
  }
}

template <int MODE>
static float run(const Args &A, int blocks, int reps) {
  CK(hipFuncSetAttribute((const void *)probe_kernel<MODE>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe_kernel<MODE>, dim3(blocks), dim3(kThreads), kLdsB, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(probe_kernel<MODE>, dim3(blocks), dim3(kThreads), kLdsB, 0, A);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int B = 65536, blocks = B / (32 * kWaves);
  std::vector<float> D((size_t)64 * B), X((size_t)64 * B);
  srand(5);
  // cotangents of a mean loss: ~1 / B, three decades of spread between trajectories
  std::vector<float> mag(B);
  for (auto &m : mag) m = powf(10.f, -3.f * (rand() / (float)RAND_MAX)) / B;
  for (int r = 0; r < 64; ++r)
    for (int n = 0; n < B; ++n) {
      D[(size_t)r * B + n] = (rand() / (float)RAND_MAX - 0.5f) * 2.f * mag[n];
      X[(size_t)r * B + n] = tanhf((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    }
  std::vector<double> ref(4096, 0.0);
  for (int m = 0; m < 64; ++m)
    for (int k = 0; k < 64; ++k) {
      double s = 0;
      const float *dm = &D[(size_t)m * B], *xk = &X[(size_t)k * B];
      for (int n = 0; n < B; ++n) s += (double)dm[n] * (double)xk[n];
      ref[m * 64 + k] = s;
    }
  // what an fp32 accumulation of the same sum is off by, for scale
  double f32_err = 0, scale = 0;
  for (int m = 0; m < 64; ++m)
    for (int k = 0; k < 64; ++k) {
      float s = 0;
      const float *dm = &D[(size_t)m * B], *xk = &X[(size_t)k * B];
      for (int n = 0; n < B; ++n) s = fmaf(dm[n], xk[n], s);
      f32_err = fmax(f32_err, fabs(s - ref[m * 64 + k]));
      scale = fmax(scale, fabs(ref[m * 64 + k]));
    }
  Args A;
  float *dd, *dx, *dp, *dpart;
  CK(hipMalloc(&dd, D.size() * 4));
  CK(hipMalloc(&dx, X.size() * 4));
  CK(hipMalloc(&dp, D.size() * 4));
  CK(hipMalloc(&dpart, (size_t)blocks * kWaves * 4096 * 4));
  CK(hipMemcpy(dd, D.data(), D.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  A.d = dd, A.x = dx, A.planes = dp, A.partial = dpart, A.B = B;
  const char *names[3] = {"store 64 cotangent planes (shipped)", "in-sweep, private 64x64 per wave",
                          "in-sweep, one 32x32 block per wave, shared staging"};
  std::vector<float> P((size_t)blocks * kWaves * 4096);
  for (int mode = 0; mode < 3; ++mode) {
    double worst = 0;
    if (mode) {
      A.iters = 1;
      CK(hipMemset(dpart, 0, P.size() * 4));
      if (mode == 1) run<1>(A, blocks, 1);
      if (mode == 2) run<2>(A, blocks, 1);
      CK(hipMemcpy(P.data(), dpart, P.size() * 4, hipMemcpyDeviceToHost));
      const int parts = mode == 1 ? blocks * kWaves : blocks;
      for (int e = 0; e < 4096; ++e) {
        double s = 0;
        for (int p = 0; p < parts; ++p) s += P[(size_t)p * 4096 + e];
        worst = fmax(worst, fabs(s - ref[e]));
      }
    }
    A.iters = 200;
    float ms = 0;
    if (mode == 0) ms = run<0>(A, blocks, 5);
    if (mode == 1) ms = run<1>(A, blocks, 5);
    if (mode == 2) ms = run<2>(A, blocks, 5);
    printf("{\"mode\": \"%s\", \"us_per_layer_all_trajectories\": %.3f, "
           "\"max_abs_err\": %.3e, \"rel_to_max\": %.3e, \"fp32_fma_loop_rel_to_max\": %.3e}\n",
           names[mode], ms * 1e3 / 200, worst, worst / scale, f32_err / scale);
  }
  return 0;
}
