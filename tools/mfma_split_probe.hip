// mfma_split_probe.hip - what would the policy layers of the fused training
// kernels gain from the bf16 matrix pipe?  ("the way to ...")
//
// One 64 -> 64 dense layer with tanh on its inputs, chained ITER times, in the
// shipped formulation (one wave = 32 trajectories, activations live in MFMA
// accumulator layout, weights as A operands from LDS, 8 waves per workgroup,
// one workgroup per CU = two waves per SIMD):
//   mode 0  v_mfma_f32_32x32x2_f32, exact fp32 (shipped)
//   mode 1  v_mfma_f32_32x32x16_bf16 on operands split into two bf16 terms,
//           three products (a_h w_h + a_l w_h + a_h w_l): ~2^-16 per product
//   mode 2  activations split into THREE bf16 terms, weights into two, six
//           products: activations exact to 2^-24, weights to 2^-17
//   mode 3  v_mfma_f32_32x32x16_f16 on operands split into two fp16 terms
//           (11 + 11 mantissa bits), three products: ~2^-22 per product for
//           operands inside fp16's range (activations after tanh, weights)
// The split costs VALU work per activation (convert, widen, subtract, convert);
// the probe measures the whole layer, not the matrix pipe alone, and checks
// every mode's result of ONE layer against a double-precision host product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_split_probe tools/mfma_split_probe.hip
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__host__ __device__ constexpr int rrow(int i) { return (i & 3) + 8 * (i >> 2); }
// fp32 path: input index fed by accumulator register c & 15 of row block c >> 4
__host__ __device__ constexpr int kchain(int c, int hi) {
  return (c >> 4) * 32 + rrow(c & 15) + 4 * hi;
}
// bf16 path: input index of slot j (0..7) of k-block kb (0..3) for half `hi`
__host__ __device__ constexpr int kin(int kb, int j, int hi) {
  return 32 * (kb >> 1) + rrow(8 * (kb & 1) + j) + 4 * hi;
}

__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
}

constexpr int kThreads = 512;
constexpr int kF32Tab = 2 * 32 * 64;      // floats
constexpr int kBfTab = 2 * 4 * 64 * 4;    // u32 per term (8 bf16 = 4 words per lane)

struct Args {
  const float *tab32;     // [2][32][64]
  const unsigned *tabh;   // [2][4][64][4]
  const unsigned *tabl;
  const unsigned *tabh16, *tabl16;   // the same split in fp16
  const float *x;         // [64][B]
  float *y;               // [64][B]
  int B, iters;
};

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned pack2h(float a, float b) {
  const f16x2 v = {(_Float16)a, (_Float16)b};   // v_cvt_pk_f16_f32 / v_cvt_pkrtz? -> check ISA
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lo_h2f(unsigned p) {
  return (float)__builtin_bit_cast(f16x2, p)[0];
}
__device__ __forceinline__ float hi_h2f(unsigned p) {
  return (float)__builtin_bit_cast(f16x2, p)[1];
}
__device__ __forceinline__ float lo_f32(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f32(unsigned p) {
  return __builtin_bit_cast(float, p & 0xffff0000u);
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void layer_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  float *t32 = reinterpret_cast<float *>(lds);
  unsigned *th = lds + kF32Tab, *tl = th + kBfTab;
  for (int i = threadIdx.x; i < kF32Tab; i += kThreads) t32[i] = A.tab32[i];
  for (int i = threadIdx.x; i < kBfTab; i += kThreads)
    th[i] = MODE == 3 ? A.tabh16[i] : A.tabh[i], tl[i] = MODE == 3 ? A.tabl16[i] : A.tabl[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  f32x16 x[2], y[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      x[rb][i] = A.x[(size_t)(rb * 32 + rrow(i) + 4 * hi) * A.B + b];
#pragma unroll 1
  for (int it = 0; it < A.iters; ++it) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) y[rb][i] = 0.f;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float bv = tanh_fast(x[c >> 4][c & 15]);
        y[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t32[(0 * 32 + c) * 64 + lane], bv, y[0], 0, 0, 0);
        y[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(t32[(1 * 32 + c) * 64 + lane], bv, y[1], 0, 0, 0);
      }
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        u32x4 ph, pm;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float t0 = tanh_fast(x[kb >> 1][8 * (kb & 1) + 2 * q]);
          const float t1 = tanh_fast(x[kb >> 1][8 * (kb & 1) + 2 * q + 1]);
          const unsigned h = pack2h(t0, t1);
          ph[q] = h, pm[q] = pack2h(t0 - lo_h2f(h), t1 - hi_h2f(h));
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, ph), bm = __builtin_bit_cast(f16x8, pm);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const u32x4 wh4 = *reinterpret_cast<const u32x4 *>(th + ((rb * 4 + kb) * 64 + lane) * 4);
          const u32x4 wl4 = *reinterpret_cast<const u32x4 *>(tl + ((rb * 4 + kb) * 64 + lane) * 4);
          const f16x8 wh = __builtin_bit_cast(f16x8, wh4), wl = __builtin_bit_cast(f16x8, wl4);
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, y[rb], 0, 0, 0);
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bm, y[rb], 0, 0, 0);
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, y[rb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        u32x4 ph, pl, pm;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float t0 = tanh_fast(x[kb >> 1][8 * (kb & 1) + 2 * q]);
          const float t1 = tanh_fast(x[kb >> 1][8 * (kb & 1) + 2 * q + 1]);
          const unsigned h = pack2(t0, t1);
          const float r0 = t0 - lo_f32(h), r1 = t1 - hi_f32(h);
          const unsigned m = pack2(r0, r1);
          ph[q] = h, pm[q] = m;
          if constexpr (MODE == 2) pl[q] = pack2(r0 - lo_f32(m), r1 - hi_f32(m));
        }
        const bf16x8 bh = __builtin_bit_cast(bf16x8, ph), bm = __builtin_bit_cast(bf16x8, pm);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          const u32x4 wh4 = *reinterpret_cast<const u32x4 *>(th + ((rb * 4 + kb) * 64 + lane) * 4);
          const u32x4 wl4 = *reinterpret_cast<const u32x4 *>(tl + ((rb * 4 + kb) * 64 + lane) * 4);
          const bf16x8 wh = __builtin_bit_cast(bf16x8, wh4), wl = __builtin_bit_cast(bf16x8, wl4);
          // small terms first
          if constexpr (MODE == 2) {
            const bf16x8 bl = __builtin_bit_cast(bf16x8, pl);
            y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bl, y[rb], 0, 0, 0);
            y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bm, y[rb], 0, 0, 0);
          }
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, bh, y[rb], 0, 0, 0);
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bm, y[rb], 0, 0, 0);
          y[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, bh, y[rb], 0, 0, 0);
        }
      }
    }
    x[0] = y[0], x[1] = y[1];
  }
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      A.y[(size_t)(rb * 32 + rrow(i) + 4 * hi) * A.B + b] = x[rb][i];
}

static unsigned short bf16_rne(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
static float bf16_f32(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int MODE>
static float run(const Args &A, int blocks, int reps) {
  const size_t lds = (kF32Tab + 2 * kBfTab) * 4;
  CK(hipFuncSetAttribute((const void *)layer_kernel<MODE>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(layer_kernel<MODE>, dim3(blocks), dim3(kThreads), lds, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(layer_kernel<MODE>, dim3(blocks), dim3(kThreads), lds, 0, A);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int blocks = 256, B = blocks * 256;
  std::vector<float> W(64 * 64), X((size_t)64 * B);
  srand(3);
  for (auto &w : W) w = (rand() / (float)RAND_MAX - 0.5f) * 0.5f;
  for (auto &x : X) x = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
  std::vector<float> t32(kF32Tab);
  std::vector<unsigned> th(kBfTab), tl(kBfTab), th16(kBfTab), tl16(kBfTab);
  for (int rb = 0; rb < 2; ++rb)
    for (int c = 0; c < 32; ++c)
      for (int l = 0; l < 64; ++l)
        t32[(rb * 32 + c) * 64 + l] = W[(rb * 32 + (l & 31)) * 64 + kchain(c, l >> 5)];
  for (int rb = 0; rb < 2; ++rb)
    for (int kb = 0; kb < 4; ++kb)
      for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) {
          unsigned wh = 0, wl = 0;
          for (int e = 0; e < 2; ++e) {
            const float w = W[(rb * 32 + (l & 31)) * 64 + kin(kb, 2 * q + e, l >> 5)];
            const unsigned short h = bf16_rne(w), lo = bf16_rne(w - bf16_f32(h));
            wh |= (unsigned)h << (16 * e), wl |= (unsigned)lo << (16 * e);
          }
          th[((rb * 4 + kb) * 64 + l) * 4 + q] = wh, tl[((rb * 4 + kb) * 64 + l) * 4 + q] = wl;
          unsigned xh = 0, xl = 0;
          for (int e = 0; e < 2; ++e) {
            const float w = W[(rb * 32 + (l & 31)) * 64 + kin(kb, 2 * q + e, l >> 5)];
            const _Float16 h = (_Float16)w, lo = (_Float16)(w - (float)h);
            unsigned short hb, lb;
            memcpy(&hb, &h, 2), memcpy(&lb, &lo, 2);
            xh |= (unsigned)hb << (16 * e), xl |= (unsigned)lb << (16 * e);
          }
          th16[((rb * 4 + kb) * 64 + l) * 4 + q] = xh, tl16[((rb * 4 + kb) * 64 + l) * 4 + q] = xl;
        }
  Args A;
  float *d32, *dx, *dy;
  unsigned *dh, *dl, *dh16, *dl16;
  CK(hipMalloc(&dh16, kBfTab * 4));
  CK(hipMalloc(&dl16, kBfTab * 4));
  CK(hipMemcpy(dh16, th16.data(), kBfTab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dl16, tl16.data(), kBfTab * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d32, kF32Tab * 4));
  CK(hipMalloc(&dh, kBfTab * 4));
  CK(hipMalloc(&dl, kBfTab * 4));
  CK(hipMalloc(&dx, X.size() * 4));
  CK(hipMalloc(&dy, X.size() * 4));
  CK(hipMemcpy(d32, t32.data(), kF32Tab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dh, th.data(), kBfTab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dl, tl.data(), kBfTab * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  A.tab32 = d32, A.tabh = dh, A.tabl = dl, A.tabh16 = dh16, A.tabl16 = dl16, A.x = dx, A.y = dy, A.B = B;
  // accuracy of ONE layer against the double-precision product (first 256 columns)
  std::vector<double> ref((size_t)64 * 256);
  for (int m = 0; m < 64; ++m)
    for (int b = 0; b < 256; ++b) {
      double acc = 0;
      for (int k = 0; k < 64; ++k) acc += (double)W[m * 64 + k] * tanh((double)X[(size_t)k * B + b]);
      ref[(size_t)m * 256 + b] = acc;
    }
  std::vector<float> Y(X.size());
  const char *names[4] = {"fp32 32x32x2", "bf16 2-term, 3 products", "bf16 3x2-term, 6 products",
                          "fp16 2-term, 3 products"};
  for (int mode = 0; mode < 4; ++mode) {
    A.iters = 1;
    if (mode == 0) run<0>(A, blocks, 1);
    if (mode == 1) run<1>(A, blocks, 1);
    if (mode == 2) run<2>(A, blocks, 1);
    if (mode == 3) run<3>(A, blocks, 1);
    CK(hipMemcpy(Y.data(), dy, Y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int m = 0; m < 64; ++m)
      for (int b = 0; b < 256; ++b) {
        worst = fmax(worst, fabs(Y[(size_t)m * B + b] - ref[(size_t)m * 256 + b]));
        scale = fmax(scale, fabs(ref[(size_t)m * 256 + b]));
      }
    A.iters = 400;
    float ms = 0;
    if (mode == 0) ms = run<0>(A, blocks, 5);
    if (mode == 1) ms = run<1>(A, blocks, 5);
    if (mode == 2) ms = run<2>(A, blocks, 5);
    if (mode == 3) ms = run<3>(A, blocks, 5);
    printf("{\"mode\": \"%s\", \"max_abs_err_one_layer\": %.3e, \"rel_to_max\": %.3e, "
           "\"us_per_layer_all_waves\": %.3f, \"cycles_per_layer_per_simd_at_2.4GHz\": %.0f}\n",
           names[mode], worst, worst / scale, ms * 1e3 / 400, ms * 1e-3 / 400 * 2.4e9);
  }
  return 0;
}
