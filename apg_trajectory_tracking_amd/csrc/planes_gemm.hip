// planes_gemm.hip - "planes x planes" reduction GEMM on the matrix cores.
//
// The weight gradients of the in-kernel LSTM policy (lstm.hip) are
//   C[m][j] = sum_{s < S} sum_{n < N} A[(m*S + s)][n] * B[boff[j] + s*bstride][n]
// with tiny M (<= 32), small J (<= 192) and an enormous reduction length
// (N = H*B = 655 360 at the benchmark batch).  Every operand row is a plane of
// N contiguous floats ("NT" layout: the reduction index is the contiguous
// one).  rocBLAS picks a 16x16 macro-tile without split-K for this shape and
// takes 1.5 ms per call; this kernel streams both operands from HBM once and
// is bound by that stream.
//
// Structure: the S*N reduction range is cut into tiles of 64; workgroups take
// tiles grid-strided.  A tile is staged in LDS as [row][64 (+1 pad)] with
// coalesced dword row-segment loads, then each of the 4 waves multiplies 8 of
// the tile's 32 k-pairs for ALL column blocks with v_mfma_f32_32x32x2_f32
// (exact f32, 16 accumulator registers per 32x32 block).  Accumulators stay in
// registers across tiles; at the end the 4 waves are summed through LDS and
// the workgroup writes one partial C; a second kernel adds the partials in a
// fixed order (deterministic, no float atomics).  An optional extra column of
// ones yields the row sums (bias gradients) for free.
#include "apg_device.h"

namespace apg {
namespace {

constexpr int kKT = 64;        // reduction elements per tile
constexpr int kLd = kKT + 1;   // padded LDS row
constexpr int kMaxNB = 6;      // column blocks of 32 (J + ones <= 192)
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float *A, *Bp;
  const int *boff;
  float *part;  // [gridDim.x][32][NB*32]
  long long N;
  int M, S, J, bstride, with_ones, NB, tiles_per_seg;
};

template <int NB>
__global__ __launch_bounds__(kThreads) void planes_gemm_kernel(GemmArgs G) {
  extern __shared__ float lds[];  // (32 + NB*32) rows of kLd floats
  float *la = lds, *lb = lds + 32 * kLd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Jt = G.J + G.with_ones;
  f32x16 acc[NB];
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[jb][i] = 0.f;

  const long long total_tiles = (long long)G.S * G.tiles_per_seg;
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int s = (int)(tile / G.tiles_per_seg);
    const long long n0 = (tile % G.tiles_per_seg) * kKT;
    // stage: thread t loads column (t & 63) of rows (t >> 6) + 4 i
    const int col = tid & 63;
    const long long n = n0 + col;
    const bool in_n = n < G.N;
#pragma unroll 4
    for (int r = wave; r < 32; r += 4) {
      float v = 0.f;
      if (r < G.M && in_n) v = G.A[((long long)r * G.S + s) * G.N + n];
      la[r * kLd + col] = v;
    }
#pragma unroll 4
    for (int r = wave; r < NB * 32; r += 4) {
      float v = 0.f;
      if (in_n) {
        if (r < G.J)
          v = G.Bp[((long long)G.boff[r] + (long long)s * G.bstride) * G.N + n];
        else if (r < Jt)
          v = 1.0f;  // the ones column: row sums of A
      }
      lb[r * kLd + col] = v;
    }
    __syncthreads();
    // wave w owns k-pairs [8w, 8w+8) of the tile
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      const int kcol = 2 * (wave * 8 + kp) + (lane >> 5);
      const float a = la[(lane & 31) * kLd + kcol];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const float b = lb[(jb * 32 + (lane & 31)) * kLd + kcol];
        acc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[jb], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // sum the 4 waves through LDS (reuse the tile buffers): [wave][32][NB*32]
  // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *red = lds;
  const int W = NB * 32;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          const int cc = jb * 32 + (lane & 31);
          if (w == 0) red[row * W + cc] = acc[jb][i];
          else red[row * W + cc] += acc[jb][i];
        }
    }
    __syncthreads();
  }
  float *out = G.part + (size_t)blockIdx.x * 32 * W;
  for (int i = tid; i < 32 * W; i += kThreads) out[i] = red[i];
}

// C[m][j] = sum over workgroups of part[wg][m][j], fixed order
__global__ __launch_bounds__(256) void planes_gemm_reduce_kernel(
    const float *__restrict__ part, int num_wg, int W, int M, int Jt,
    float *__restrict__ C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * Jt) return;
  const int m = idx / Jt, j = idx % Jt;
  double acc = 0.0;
  for (int w = 0; w < num_wg; ++w) acc += (double)part[((size_t)w * 32 + m) * W + j];
  C[idx] = (float)acc;
}

template <int NB>
int launch(const GemmArgs &G, int num_wg, hipStream_t st) {
  const size_t lds = (size_t)(32 + NB * 32) * kLd * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)planes_gemm_kernel<NB>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm)");
    attr_set = true;
  }
  hipLaunchKernelGGL((planes_gemm_kernel<NB>), dim3(num_wg), dim3(kThreads), lds,
                     st, G);
  return check_launch("planes_gemm");
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_planes_gemm_workspace_floats(int J, int with_ones, int num_wg) {
  const int NB = (J + (with_ones ? 1 : 0) + 31) / 32;
  return num_wg * 32 * NB * 32;
}

int apg_planes_gemm(const float *A, int M, int S, const float *Bp,
                    const int *boff, int J, int bstride, int with_ones,
                    long long N, float *workspace, int num_wg, float *C,
                    apg_stream_t stream) {
  const int Jt = J + (with_ones ? 1 : 0);
  if (!A || !Bp || !boff || !workspace || !C) {
    set_error("apg_planes_gemm: NULL pointer");
    return APG_ERR_ARG;
  }
  if (M < 1 || M > 32 || S < 1 || J < 1 || Jt > kMaxNB * 32 || N < 1 ||
      num_wg < 1) {
    set_error("apg_planes_gemm: need 1 <= M <= 32, J + ones <= %d, S, N, "
              "num_wg >= 1", kMaxNB * 32);
    return APG_ERR_ARG;
  }
  GemmArgs G;
  G.A = A, G.Bp = Bp, G.boff = boff, G.part = workspace;
  G.N = N, G.M = M, G.S = S, G.J = J, G.bstride = bstride;
  G.with_ones = with_ones ? 1 : 0;
  G.NB = (Jt + 31) / 32;
  G.tiles_per_seg = (int)((N + kKT - 1) / kKT);
  hipStream_t st = (hipStream_t)stream;
  int e;
  switch (G.NB) {
    case 1: e = launch<1>(G, num_wg, st); break;
    case 2: e = launch<2>(G, num_wg, st); break;
    case 3: e = launch<3>(G, num_wg, st); break;
    case 4: e = launch<4>(G, num_wg, st); break;
    case 5: e = launch<5>(G, num_wg, st); break;
    default: e = launch<6>(G, num_wg, st); break;
  }
  if (e) return e;
  hipLaunchKernelGGL(planes_gemm_reduce_kernel, dim3((M * Jt + 255) / 256),
                     dim3(256), 0, st, workspace, num_wg, G.NB * 32, M, Jt, C);
  return check_launch("planes_gemm_reduce");
}

}  // extern "C"
