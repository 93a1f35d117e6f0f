// linear_wgrad.hip - the weight gradient of a PyTorch-side policy layer.
//
// Where the policy is NOT inside a kernel (any torch.nn policy on the
// row-layout path, TrainDrone.train_controller_packed; the per-step fallbacks
// of the recurrent modes; horizons / widths the fused kernels are not built
// for), autograd computes a Linear layer's weight gradient as
//   dW = dY^T X,  db = sum_b dY        dY [B, M], X [B, N] row-major
// (scripts/train_base.py:198-209 -> loss.backward()).  With B = 65 536 rows
// and M, N <= 256 this is a tall-skinny reduction: rocBLAS takes 190-210 us
// per layer for it (profiles/r03_packed_step_timeline.txt), four times what
// the whole fused rollout costs.  It is a stream over 4 (M + N) B bytes with a
// tiny output, so: split-K over all waves, every wave multiplies its rows'
// [M x 2] x [2 x N] slivers on v_mfma_f32_32x32x2_f32 (exact fp32: A operand
// lane l supplies dY[row pair member l >> 5][m = l & 31] - 128 contiguous
// bytes per half-wave, no transpose, no LDS), accumulators stay in registers,
// the four waves of a workgroup are summed through LDS and a second kernel
// adds the workgroups' partials in a fixed order (no float atomics).
#include "apg_device.h"

namespace apg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 64 x 64 outputs per workgroup column: 208 registers per wave leave room for
// two workgroups per CU (a 64 x 128 tile - 352 registers, one wave per SIMD -
// measured 2-4 x slower per byte, profiles/r03_linear_wgrad.jsonl)
constexpr int kTM = 2, kTN = 2;
constexpr int kThreads = 256;
constexpr int kUnroll = 4;            // row pairs in flight per wave

struct WgradArgs {
  const float *dY, *X;
  float *part;          // [split][tiles][kTM*32][kTN*32 + 1]
  long long B;
  int M, N, split;
};

__global__ __launch_bounds__(kThreads) void linear_wgrad_kernel(WgradArgs A) {
  constexpr int kW = kTN * 32 + 1;    // partial row pitch; last column = row sums
  __shared__ float red[kTM * 32 * kW];
  const int lane = threadIdx.x & 63, hi = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.z * kTM * 32, n0 = blockIdx.y * kTN * 32;
  const auto rY = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A.dY), 0, (unsigned)(A.B * A.M * 4), 0x00020000);
  const auto rX = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A.X), 0, (unsigned)(A.B * A.N * 4), 0x00020000);
  constexpr unsigned kDeadOff = 0xfffffff0u;
  unsigned offY[kTM], offX[kTN];     // byte offset of this lane's column in a row
#pragma unroll
  for (int mb = 0; mb < kTM; ++mb) {
    const int m = m0 + mb * 32 + c;
    offY[mb] = m < A.M ? (unsigned)m * 4u : kDeadOff;
  }
#pragma unroll
  for (int nb = 0; nb < kTN; ++nb) {
    const int n = n0 + nb * 32 + c;
    offX[nb] = n < A.N ? (unsigned)n * 4u : kDeadOff;
  }
  f32x16 acc[kTM][kTN];
  float rsum[kTM];
#pragma unroll
  for (int mb = 0; mb < kTM; ++mb) {
    rsum[mb] = 0.f;
#pragma unroll
    for (int nb = 0; nb < kTN; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;
  }
  // row pairs are dealt to the waves of the split round-robin in groups of
  // kUnroll: at any time the grid reads one contiguous stretch of both tensors
  const long long pairs = (A.B + 1) / 2;
  const long long nw = (long long)A.split * 4;
  const unsigned pitchY = (unsigned)A.M * 4u, pitchX = (unsigned)A.N * 4u;
  // two register sets: the loads of the next group are in flight while this
  // one multiplies (one wave per SIMD and workgroup: latency, not issue, bound)
  auto load = [&](long long g, float (&a)[kUnroll][kTM], float (&b)[kUnroll][kTN]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const long long row = (g * kUnroll + u) * 2 + hi;   // rows past B: out of range
      const bool live = row < A.B;
      const unsigned rb = (unsigned)row;
#pragma unroll
      for (int mb = 0; mb < kTM; ++mb) {
        const unsigned off = (live && offY[mb] != kDeadOff) ? rb * pitchY + offY[mb] : kDeadOff;
        a[u][mb] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rY, (int)off, 0, 0));
      }
#pragma unroll
      for (int nb = 0; nb < kTN; ++nb) {
        const unsigned off = (live && offX[nb] != kDeadOff) ? rb * pitchX + offX[nb] : kDeadOff;
        b[u][nb] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rX, (int)off, 0, 0));
      }
    }
  };
  auto multiply = [&](const float (&a)[kUnroll][kTM], const float (&b)[kUnroll][kTN]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
#pragma unroll
      for (int mb = 0; mb < kTM; ++mb) {
        rsum[mb] += a[u][mb];
#pragma unroll
        for (int nb = 0; nb < kTN; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mb], b[u][nb],
                                                             acc[mb][nb], 0, 0, 0);
      }
  };
  float a0[kUnroll][kTM], b0[kUnroll][kTN], a1[kUnroll][kTM], b1[kUnroll][kTN];
  long long g = (long long)blockIdx.x * 4 + wave;
  load(g, a0, b0);          // (a group past the end loads nothing: all offsets dead)
  while (g * kUnroll < pairs) {
    load(g + nw, a1, b1);
    multiply(a0, b0);
    g += nw;
    if (g * kUnroll >= pairs) break;
    load(g + nw, a0, b0);
    multiply(a1, b1);
    g += nw;
  }
  // sum the four waves through LDS: C / D register i of lane l is row
  // (i & 3) + 8 (i >> 2) + 4 hi, column l & 31
#pragma unroll
  for (int mb = 0; mb < kTM; ++mb) rsum[mb] += __shfl_xor(rsum[mb], 32, 64);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mb = 0; mb < kTM; ++mb) {
#pragma unroll
        for (int nb = 0; nb < kTN; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            const int idx = row * kW + nb * 32 + c;
            if (w == 0) red[idx] = acc[mb][nb][i];
            else red[idx] += acc[mb][nb][i];
          }
        if (hi == 0) {  // row sums: lane c holds the sum of column m = mb*32 + c of dY
          const int idx = (mb * 32 + c) * kW + kTN * 32;
          if (w == 0) red[idx] = rsum[mb];
          else red[idx] += rsum[mb];
        }
      }
    }
    __syncthreads();
  }
  const int tiles = gridDim.y * gridDim.z, tile = blockIdx.z * gridDim.y + blockIdx.y;
  float *out = A.part + ((size_t)blockIdx.x * tiles + tile) * (kTM * 32 * kW);
  for (int i = threadIdx.x; i < kTM * 32 * kW; i += kThreads) out[i] = red[i];
}

struct WreduceArgs {
  const float *part;
  float *dW, *db;
  int M, N, split, tiles_n, tiles;
};

// dW[m][n] (and db[m]) = sum over the split: 32 outputs per workgroup, 32
// slices of the split each summed in order by one thread, then the 32 slice
// sums in order (fixed order, double accumulation)
__global__ __launch_bounds__(1024) void linear_wgrad_reduce_kernel(WreduceArgs R) {
  __shared__ double sh[32][33];
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + x;
  const int cols = R.N + 1, kW = kTN * 32 + 1;
  const bool ok = idx < R.M * cols;
  const int m = ok ? idx / cols : 0, n = ok ? idx % cols : 0;
  // column N of the output = the row sums, kept by the FIRST tile of each tile row
  const int tm = m / (kTM * 32), tn = n == R.N ? 0 : n / (kTN * 32);
  const int col = n == R.N ? kTN * 32 : n % (kTN * 32);
  const size_t tile = (size_t)tm * R.tiles_n + tn;
  const float *p = R.part + tile * (kTM * 32 * kW) + (size_t)(m % (kTM * 32)) * kW + col;
  const size_t stride = (size_t)R.tiles * (kTM * 32 * kW);
  const int per = (R.split + 31) / 32;
  const int s0 = y * per, s1 = s0 + per < R.split ? s0 + per : R.split;
  double acc[4] = {0, 0, 0, 0};
  int sidx = s0;
  if (ok) {
    for (; sidx + 4 <= s1; sidx += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += (double)p[(size_t)(sidx + u) * stride];
    }
    for (; sidx < s1; ++sidx) acc[0] += (double)p[(size_t)sidx * stride];
  }
  sh[y][x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (y == 0 && ok) {
    double t = 0;
    for (int k = 0; k < 32; ++k) t += sh[k][x];
    if (n == R.N) {
      if (R.db) R.db[m] = (float)t;
    } else {
      R.dW[(size_t)m * R.N + n] = (float)t;
    }
  }
}

int cu_count_() { return device_cu_count(); }

void shape(int M, int N, int &tm, int &tn, int &split) {
  tm = (M + kTM * 32 - 1) / (kTM * 32);
  tn = (N + kTN * 32 - 1) / (kTN * 32);
  // two workgroups per CU over all tiles, at least one per tile
  split = (2 * cu_count_() + tm * tn - 1) / (tm * tn);
  if (split < 1) split = 1;
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

long long apg_linear_wgrad_workspace_floats(int M, int N) {
  if (M < 1 || N < 1) return 0;
  int tm, tn, split;
  shape(M, N, tm, tn, split);
  return (long long)split * tm * tn * (kTM * 32 * (kTN * 32 + 1));
}

int apg_linear_wgrad(const float *dY, const float *X, long long B, int M, int N,
                     float *dW, float *db, float *workspace, apg_stream_t stream) {
  if (B < 0 || M < 1 || N < 1) {
    set_error("linear_wgrad: need B >= 0, M, N >= 1 (got %lld, %d, %d)", B, M, N);
    return APG_ERR_ARG;
  }
  if (!dW || (B > 0 && (!dY || !X || !workspace))) {
    set_error("linear_wgrad: NULL pointer");
    return APG_ERR_ARG;
  }
  if (B * (long long)(M > N ? M : N) * 4 >= 0xfffffff0ll) {
    set_error("linear_wgrad: operands must stay below 4 GiB (32-bit buffer offsets)");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (hipMemsetAsync(dW, 0, (size_t)M * N * sizeof(float), st) != hipSuccess ||
        (db && hipMemsetAsync(db, 0, (size_t)M * sizeof(float), st) != hipSuccess)) {
      set_error("linear_wgrad: hipMemsetAsync failed");
      return APG_ERR_HIP;
    }
    return APG_OK;
  }
  int tm, tn, split;
  shape(M, N, tm, tn, split);
  // no more workgroups along the rows than groups of row pairs
  const long long groups = ((B + 1) / 2 + kUnroll - 1) / kUnroll;
  if ((long long)split * 4 > groups) split = (int)((groups + 3) / 4);
  WgradArgs A;
  A.dY = dY, A.X = X, A.part = workspace, A.B = B, A.M = M, A.N = N, A.split = split;
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(split, tn, tm), dim3(kThreads), 0, st, A);
  WreduceArgs R;
  R.part = workspace, R.dW = dW, R.db = db, R.M = M, R.N = N, R.split = split;
  R.tiles_n = tn, R.tiles = tm * tn;
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((M * (N + 1) + 31) / 32), dim3(1024),
                     0, st, R);
  return check_launch("linear_wgrad");
}

}  // extern "C"
