// transposition_probe.hip - VERDICT r5 next #2(b): what does it cost to bring a
// layer's cotangent from the orientation the CHAIN needs (accumulator layout:
// trajectory in the lane, feature in the registers) into the orientation the
// WEIGHT PRODUCT needs (feature in the lane, trajectories in the k-slots of the
// A operand), inside a reverse sweep like mlp_rollout_bwd_tm_kernel /
// mlp_concurrent_bwd_tm_kernel (csrc/mlp_rollout.hip, mlp_concurrent.hip)?
//
// One 64 x 64 tanh layer over B = 65 536 trajectories, workgroups of 8 waves
// (32 trajectories per wave), one workgroup per CU, the kernels' own building
// blocks (csrc/policy_tm.h, policy_mfma16.h: fp16-split operands,
// v_mfma_f32_32x32x16_f16, per-trajectory power-of-two scales).  Per layer every
// variant does the same common work
//   * split the incoming cotangent d (scaled_split64),
//   * the feature-major chain  d' = (W^T d) (1 - x^2)  (24 matrix instructions,
//     x brought into accumulator layout by an identity product as the kernels do),
//   * the four 32 x 32 weight blocks  dW += d x^T  (24 matrix instructions) added
//     into the workgroup's fixed-point accumulators in LDS exactly as the kernels
//     do (policy_tm.h add_block: rint + ds_add_u32; workgroup exponent exchanged
//     through LDS behind ONE barrier per layer, two alternating regions, the other
//     region flushed into the workgroup's global partials by atomic adds),
// and differs ONLY in how the A operands of the weight blocks are made:
//   0 swapped    what is shipped: the chain a second time with the operands
//                swapped (trajectory-major result), tanh' and split again
//   1 identity   the split cotangent times an identity B operand (4 matrix
//                instructions per 32 features), rescale, split
//   2 lds_tr     the split cotangent written to LDS as packed fp16 (ds_write_b64,
//                8-byte units swizzled) and read back with ds_read_b64_tr_b16:
//                the result IS the A operand up to the per-trajectory scales, which
//                a packed fp16 multiply per dword applies (exact: powers of two)
//   3 permlane   the same exchange in registers: six lane-bit <-> register-bit
//                stages (v_perm_b32 + DPP, DPP with bank masks, v_permlane16_swap,
//                v_permlane32_swap)
// Prints one JSON line per variant: us per layer (all 65 536 trajectories), the
// error of the accumulated 64 x 64 gradient after two chained layers against a
// double-precision host evaluation; tools/transposition_probe_counts.py adds the
// instruction counts of each variant's loop body from the disassembly.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude \
//     -Iapg_trajectory_tracking_amd/csrc -o tools/exp/transposition_probe tools/transposition_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "policy_tm.h"

using namespace apg;

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);   \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

// (the library's host-side symbols the headers declare)
namespace apg {
void set_error(const char *, ...) {}
int check_launch(const char *) { return 0; }
int device_slot() { return 0; }
int device_cu_count() { return 256; }
}  // namespace apg

constexpr int kThr = 512, kWavesWG = kThr / 64;
constexpr int kTab = 8 * kBlock16;                 // W^T as 8 A-operand blocks
constexpr int kAcc = kTab;                         // two regions of four 4 KB blocks
constexpr int kMeta = kAcc + 2 * 16384;            // the waves' maxima [2][8]
constexpr int kStage = kMeta + 256;                // per-wave staging (variant 2)
constexpr int kSub = 1152, kTerm = 4 * kSub, kStageWave = 2 * kTerm;   // 9 216 B per wave
constexpr int kLdsB = 147456;                      // 144 KB: one workgroup per CU

struct Args {
  const float *d0;     // [64][B]   incoming cotangent
  const float *x;      // [64][B]   the layer's input activations (tanh values)
  const float *W;      // [64][64]  W[m][k]
  float *partial;      // [workgroups][4 blocks][16 registers][64 lanes]
  int B, iters;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// ---- variant 2: LDS staging, 8-byte units [term][16-feature subtile][trajectory][chunk]
__device__ __forceinline__ void stage_write(char *st, const Op16 (&x)[4], int n, int hi) {
  const int sw = (n >> 2) & 3;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    char *p = st + kb * kSub + n * 32;      // subtile 2 (kb >> 1) + (kb & 1) = kb
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int s = hi + 2 * half;           // features 4 hi + 8 half .. + 3 of the subtile
      const u32x2 vh = {x[kb].h[2 * half], x[kb].h[2 * half + 1]};
      const u32x2 vl = {x[kb].l[2 * half], x[kb].l[2 * half + 1]};
      *reinterpret_cast<u32x2 *>(p + ((s ^ sw) * 8)) = vh;
      *reinterpret_cast<u32x2 *>(p + kTerm + ((s ^ sw) * 8)) = vl;
    }
  }
}
__device__ __forceinline__ u32x2 tr_read(const char *p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_ptr;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4_ptr)(unsigned)(size_t)p);
  return __builtin_bit_cast(u32x2, v);
}
// A operands [mb][kk] of the staged cotangent: lane = feature 32 mb + (lane & 31)
__device__ __forceinline__ void stage_read(const char *st, int lane, Op16 (&ad)[2][2]) {
  const int m = lane & 31, hi = lane >> 5, g16 = (m >> 4) & 1, p = lane & 15;
  const int j = p >> 2, s = p & 3;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int t = 8 * (2 * kk + r) + 4 * hi + j;      // trajectory row of this lane's unit
        const char *q = st + (2 * mb + g16) * kSub + t * 32 + ((s ^ ((t >> 2) & 3)) * 8);
        const u32x2 vh = tr_read(q), vl = tr_read(q + kTerm);
        ad[mb][kk].h[2 * r] = vh[0], ad[mb][kk].h[2 * r + 1] = vh[1];
        ad[mb][kk].l[2 * r] = vl[0], ad[mb][kk].l[2 * r + 1] = vl[1];
      }
}

// ---- variant 3: six exchange stages on the 16 dwords of one term
template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dpp_merge(unsigned old, unsigned src) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK, false);
}
__device__ __forceinline__ void butterfly(unsigned (&u)[16], int lane) {
  // S1  half-word <-> lane bit 0
  const unsigned sel = (lane & 1) ? 0x03020706u : 0x05040100u;
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u[d], 0xB1, 0xf, 0xf, true);
    u[d] = __builtin_amdgcn_perm(t, u[d], sel);
  }
  // S2  dword bit 0 <-> lane bit 1
  const bool l1 = (lane >> 1) & 1;
#pragma unroll
  for (int d = 0; d < 16; d += 2) {
    const unsigned a = u[d], b = u[d + 1];
    const unsigned ta = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, 0x4E, 0xf, 0xf, true);
    const unsigned tb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xf, 0xf, true);
    u[d] = l1 ? tb : a;
    u[d + 1] = l1 ? b : ta;
  }
  // S3  dword bit 1 <-> lane bit 3 (row_shr:8 / row_shl:8 under bank masks)
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if (d & 2) continue;
    const unsigned a = u[d], b = u[d + 2];
    u[d] = dpp_merge<0x118, 0xC>(a, b);
    u[d + 2] = dpp_merge<0x108, 0x3>(b, a);
  }
  // S4  dword bit 2 <-> lane bit 4
#pragma unroll
  for (int d = 0; d < 16; ++d) {
    if (d & 4) continue;
    const auto r = __builtin_amdgcn_permlane16_swap(u[d], u[d + 4], false, false);
    u[d] = r[0], u[d + 4] = r[1];
  }
  // S5  dword bit 3 <-> lane bit 2 (row_shr:4 / row_shl:4 under bank masks)
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const unsigned a = u[d], b = u[d + 8];
    u[d] = dpp_merge<0x114, 0xA>(a, b);
    u[d + 8] = dpp_merge<0x104, 0x5>(b, a);
  }
  // S6  dword bit 3 <-> lane bit 5
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const auto r = __builtin_amdgcn_permlane32_swap(u[d], u[d + 8], false, false);
    u[d] = r[0], u[d + 8] = r[1];
  }
}
// afterwards lane & 31 = f0 + 2 f1 + 4 f5 + 8 f3 + 16 f4 of block f2 (dword bit 3);
// dword bits (2, 1, 0) + half-word = (n4, n3, n1, n0), lane bit 5 = n2
__host__ __device__ constexpr int v3_feature(int block, int a) {
  return (a & 1) + 2 * ((a >> 1) & 1) + 4 * block + 8 * ((a >> 3) & 1) + 16 * ((a >> 4) & 1) +
         32 * ((a >> 2) & 1);
}

template <int V>
__global__ __launch_bounds__(kThr) void probe_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t0 = (blockIdx.x * kWavesWG + wave) * 32;    // the wave's first trajectory
  const int B = A.B;
  // W^T as A-operand blocks [nb][kb]: row = input feature 32 nb + row, slot -> output feature
  for (int blk = wave; blk < 8; blk += kWavesWG) {
    const int nb = blk >> 2, kb = blk & 3;
    float w8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w8[j] = A.W[kin(kb, j, hi) * 64 + 32 * nb + row];
    const Op16 o = split8(w8);
    *reinterpret_cast<u32x4 *>(lds + blk * kBlock16 + lane * 16) = o.h;
    *reinterpret_cast<u32x4 *>(lds + blk * kBlock16 + 1024 + lane * 16) = o.l;
  }
  zero_region(lds, kAcc, 2 * 16384 + 256);
  const __amdgpu_buffer_rsrc_t part = __builtin_amdgcn_make_buffer_rsrc(
      A.partial + (size_t)blockIdx.x * 4096, 0, 16384, 0x00020000);
  for (int i = threadIdx.x; i < 4096; i += kThr) A.partial[(size_t)blockIdx.x * 4096 + i] = 0.f;
  __syncthreads();
  const LdsView16 L16(lds, lane);
  // incoming cotangent, accumulator layout
  f32x16 d[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      d[rb][i] = A.d0[(size_t)(32 * rb + rrow(i) + 4 * hi) * B + t0 + row];
  // the layer's x, trajectory-major: lane = plane 32 nb + row, 16 trajectories from 4 hi
  float xT[2][16];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        xT[nb][4 * g + c] = A.x[(size_t)(32 * nb + row) * B + t0 + c + 8 * g + 4 * hi];
  float dT[2][16];   // variant 0 carries the trajectory-major cotangent
  if constexpr (V == 0) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          dT[mb][4 * g + c] = A.d0[(size_t)(32 * mb + row) * B + t0 + c + 8 * g + 4 * hi];
  }
  u32x4 ident[2];
  ident_operands(lane, ident);
  char *stage = lds + kStage + wave * kStageWave;
  unsigned (*slots)[8] = reinterpret_cast<unsigned (*)[8]>(lds + kMeta);
  int rg = kAcc, ro = kAcc + 16384, e_prev = 0;

#pragma unroll 1
  for (int it = 0; it < A.iters; ++it) {
    // ---- common: split of the incoming cotangent (per-trajectory scale 2^ex)
    Op16 x[4];
    const int ex = scaled_split64(d, x);
    // the workgroup's exponent: the waves' maxima through LDS, one barrier per
    // layer; behind it the OTHER region (the previous layer's blocks) is flushed
    unsigned am = 0u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) am = umax_abs(am, d[rb][i]);
    am = wave_umax(am);
    if (lane == 0) slots[it & 1][wave] = am;
    __syncthreads();
    bool bad = false;
    const int ew = wg_exp(slots[it & 1], bad);
    if (it > 0) flush_add<4096>(lds, ro, part, 0, e_prev, bad);
    e_prev = ew;
    // ---- the A operands of the weight blocks
    Op16 ad[2][2];
    int E[16];
    if constexpr (V == 0) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) split16(dT[mb], ew - kPreD, ad[mb]);
    } else {
      texp(ex, hi, E);
    }
    if constexpr (V == 1) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const Op16 pr[2] = {x[2 * mb], x[2 * mb + 1]};
        const f32x16 tz = to_feature_major(pr, ident);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_ldexpf(tz[i], E[i] - ew + kPreD);
        split16(v, 0, ad[mb]);
      }
    } else if constexpr (V == 2) {
      stage_write(stage, x, row, hi);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stage_read(stage, lane, ad);
      __builtin_amdgcn_wave_barrier();
    } else if constexpr (V == 3) {
      unsigned uh[16], ul[16];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) uh[4 * kb + q] = x[kb].h[q], ul[4 * kb + q] = x[kb].l[q];
      butterfly(uh, lane);
      butterfly(ul, lane);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ad[mb][kk].h[q] = uh[8 * mb + 4 * kk + q], ad[mb][kk].l[q] = ul[8 * mb + 4 * kk + q];
    }
    if constexpr (V >= 2) {
      // per-trajectory scales 2^(E - ew + kPreD) of the k-slots, packed fp16 pairs
      typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
      hv2 sc[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const hv2 t = {(_Float16)__builtin_amdgcn_ldexpf(1.f, E[8 * kk + 2 * q] - ew + kPreD),
                         (_Float16)__builtin_amdgcn_ldexpf(1.f, E[8 * kk + 2 * q + 1] - ew + kPreD)};
          sc[kk][q] = t;
        }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (element copies first: __builtin_bit_cast of a vector ELEMENT
            // expression reads element 0 of the vector)
            const unsigned wh = ad[mb][kk].h[q], wl = ad[mb][kk].l[q];
            ad[mb][kk].h[q] = __builtin_bit_cast(unsigned, __builtin_bit_cast(hv2, wh) * sc[kk][q]);
            ad[mb][kk].l[q] = __builtin_bit_cast(unsigned, __builtin_bit_cast(hv2, wl) * sc[kk][q]);
          }
    }
    // ---- per x block: weight blocks, chain
    f32x16 nx[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      Op16 bx[2];      // x as the chain needs it (tanh'), scaled 2^kPreX
      split16(xT[nb], -kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        add_block(lds + rg + (2 * mb + nb) * 4096 + lane * 4, acc);
      }
      f32x16 tt;
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = 0.f, nx[nb][i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const Op16 w = L16.A(0, 4 * nb + kb);
        if constexpr (V == 0) tt = mma3(x[kb], w, tt);
        nx[nb] = mma3(w, x[kb], nx[nb]);
      }
      if constexpr (V == 0) {
        int E0[16];
        texp(ex, hi, E0);
#pragma unroll
        for (int i = 0; i < 16; ++i)
          dT[nb][i] = __builtin_amdgcn_ldexpf(tt[i], E0[i]) * (1.f - xT[nb][i] * xT[nb][i]);
      }
      const f32x16 hf = to_feature_major(bx, ident);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float hx = __builtin_amdgcn_ldexpf(hf[i], -kPreX);
        nx[nb][i] = __builtin_amdgcn_ldexpf(nx[nb][i], ex) * (1.f - hx * hx);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    d[0] = nx[0], d[1] = nx[1];
    { const int r_ = rg; rg = ro, ro = r_; }
  }
  __syncthreads();
  bool bad = false;
  flush_add<4096>(lds, ro, part, 0, e_prev, bad);
}

template <int V>
static float run(const Args &A, int blocks, int reps) {
  CK(hipFuncSetAttribute((const void *)probe_kernel<V>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsB));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe_kernel<V>, dim3(blocks), dim3(kThr), kLdsB, 0, A);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(probe_kernel<V>, dim3(blocks), dim3(kThr), kLdsB, 0, A);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static float dispatch(int v, const Args &A, int blocks, int reps) {
  switch (v) {
    case 0: return run<0>(A, blocks, reps);
    case 1: return run<1>(A, blocks, reps);
    case 2: return run<2>(A, blocks, reps);
    default: return run<3>(A, blocks, reps);
  }
}

int main() {
  const int B = 65536, blocks = B / (32 * kWavesWG);
  std::vector<float> D((size_t)64 * B), X((size_t)64 * B), W(4096);
  srand(5);
  std::vector<float> mag(B);
  for (auto &m : mag) m = powf(10.f, -3.f * (rand() / (float)RAND_MAX)) / B;
  for (int r = 0; r < 64; ++r)
    for (int n = 0; n < B; ++n) {
      D[(size_t)r * B + n] = (rand() / (float)RAND_MAX - 0.5f) * 2.f * mag[n];
      X[(size_t)r * B + n] = tanhf((rand() / (float)RAND_MAX - 0.5f) * 3.f);
    }
  for (auto &w : W) w = (rand() / (float)RAND_MAX - 0.5f) * 0.7f;
  // host, double: two chained layers, dW = d0 x^T + d1 x^T, d1 = (W^T d0)(1 - x^2)
  std::vector<double> ref(4096, 0.0);
  {
    std::vector<double> d1((size_t)64 * 4096);
    for (int n0 = 0; n0 < B; n0 += 4096) {
      for (int k = 0; k < 64; ++k)
        for (int n = 0; n < 4096; ++n) {
          double s = 0;
          for (int m = 0; m < 64; ++m) s += (double)W[m * 64 + k] * (double)D[(size_t)m * B + n0 + n];
          const double xv = X[(size_t)k * B + n0 + n];
          d1[(size_t)k * 4096 + n] = s * (1.0 - xv * xv);
        }
      for (int m = 0; m < 64; ++m)
        for (int k = 0; k < 64; ++k) {
          double s = 0;
          for (int n = 0; n < 4096; ++n)
            s += ((double)D[(size_t)m * B + n0 + n] + d1[(size_t)m * 4096 + n]) *
                 (double)X[(size_t)k * B + n0 + n];
          ref[m * 64 + k] += s;
        }
    }
  }
  double scale = 0;
  for (double v : ref) scale = fmax(scale, fabs(v));
  Args A;
  float *dd, *dx, *dw, *dpart;
  CK(hipMalloc(&dd, D.size() * 4));
  CK(hipMalloc(&dx, X.size() * 4));
  CK(hipMalloc(&dw, W.size() * 4));
  CK(hipMalloc(&dpart, (size_t)blocks * 4096 * 4));
  CK(hipMemcpy(dd, D.data(), D.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  A.d0 = dd, A.x = dx, A.W = dw, A.partial = dpart, A.B = B;
  const char *names[4] = {"swapped product (shipped)", "identity product", "LDS + ds_read_b64_tr_b16",
                          "permlane / DPP exchange stages"};
  std::vector<float> P((size_t)blocks * 4096);
  for (int v = 0; v < 4; ++v) {
    A.iters = 2;
    dispatch(v, A, blocks, 1);
    CK(hipMemcpy(P.data(), dpart, P.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int e = 0; e < 4096; ++e) {
      double s = 0;
      for (int p = 0; p < blocks; ++p) s += P[(size_t)p * 4096 + e];
      // accumulator layout: block 2 mb + nb, register i, lane (hi, col)
      const int blk = e >> 10, i = (e >> 6) & 15, ln = e & 63;
      const int a = rrow(i) + 4 * (ln >> 5), k = 32 * (blk & 1) + (ln & 31);
      int m = 32 * (blk >> 1) + a;
      if (v == 3) m = v3_feature(blk >> 1, a);   // variant 3's own row order
      worst = fmax(worst, fabs(s - ref[m * 64 + k]));
    }
    A.iters = 200;
    const float ms = dispatch(v, A, blocks, 5);
    printf("{\"variant\": %d, \"name\": \"%s\", \"us_per_layer_all_trajectories\": %.3f, "
           "\"rel_err_two_layers\": %.3e}\n", v, names[v], ms * 1e3 / 200, worst / scale);
  }
  return 0;
}
