// planes_gemm.hip - "planes x planes" reduction GEMM on the matrix cores.
//
// The weight gradients of the in-kernel policies (lstm.hip, mlp.hip) are
//   C[m][j] = sum_{s < S} sum_{n < N} A[(m*S + s)][n] * B[bplane(j, s)][n]
//   bplane(j, s) = boff[j] + (s / sdiv) * bstride + (s % sdiv) * bstride2
// with small M (<= 64), small J (<= 192) and an enormous reduction length
// (S * N = 655 360 ... 5 242 880 at the benchmark batch).  Every operand row is
// a plane of N contiguous floats ("NT" layout: the reduction index is the
// contiguous one).  rocBLAS picks a 16x16 macro-tile without split-K for this
// shape and takes 1.5 ms per call; this kernel streams both operands from HBM
// once and is bound by that stream.  The two-level segment stride lets the
// conv-weight gradient read the sliding reference windows straight from the
// [2H][9][B] reference tensor (segment = (window position, step)) instead of
// from a materialised [90][H*B] copy.
//
// Structure: the S*N reduction range is cut into tiles of 64; workgroups take
// tiles grid-strided.  A tile is staged in LDS as [row][64 (+1 pad)] with
// coalesced dword row-segment loads, then each of the 4 waves multiplies 8 of
// the tile's 32 k-pairs for ALL row / column blocks with
// v_mfma_f32_32x32x2_f32 (exact f32, 16 accumulator registers per 32x32
// block; an A fragment is reused by every column block, a B fragment by both
// row blocks).  Accumulators stay in registers across tiles; at the end the 4
// waves are summed through LDS and the workgroup writes one partial C; a
// second kernel adds the partials in a fixed order (deterministic, no float
// atomics).  An optional extra column of ones yields the row sums (bias
// gradients) for free.
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
  float *part;  // [gridDim.x][MB*32][NB*32]
  long long N, a_bytes, b_bytes;
  int M, S, J, bstride, sdiv, bstride2, with_ones, tiles_per_seg;
};

template <int MB, int NB>
__global__ __launch_bounds__(kThreads) void planes_gemm_kernel(GemmArgs G) {
  extern __shared__ float lds[];  // (MB*32 + NB*32) rows of kLd floats
  float *la = lds, *lb = lds + MB * 32 * kLd;
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index is wave-uniform; tell the compiler, so that everything
  // derived from it (row numbers, plane offsets) lives in SGPRs
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane;  // column of the tile this lane stages
  const int Jt = G.J + G.with_ones;
  constexpr int RA = MB * 8, RB = NB * 8;  // rows staged per wave (row = wave + 4 i)
  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][jb][i] = 0.f;

  // Branch-free staging through buffer resources: per-lane column offset in a
  // VGPR, the row's plane offset in an SGPR (row numbers are wave-uniform).
  // EVERY row is loaded unconditionally - padding rows and the ones row read
  // plane 0 - and is turned into what it should be by one FMA when it is
  // written to LDS:   real row: v*1 + 0   padding: v*0 + 0   ones: v*0 + 1.
  // No conditional touches a loaded value before that point, so all loads of
  // a tile are in flight together and overlap the previous tile's MFMAs.
  const unsigned plane_bytes = (unsigned)(G.N * 4);
  const auto rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.A), 0, (unsigned)G.a_bytes, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.Bp), 0, (unsigned)G.b_bytes, 0x00020000);

  float va[RA], vb[RB];
  float keep = 1.f;  // 0 for the lanes of a ragged last tile beyond N
  auto fetch = [&](long long tile) {  // issue every load of a tile, no waits
    const int s = (int)(tile / G.tiles_per_seg);
    const int sb = (s / G.sdiv) * G.bstride + (s % G.sdiv) * G.bstride2;
    const long long n = (tile % G.tiles_per_seg) * kKT + col;
    keep = n < G.N ? 1.f : 0.f;
    const int voff = (int)((n < G.N ? n : G.N - 1) * 4);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = wave + 4 * i;
      const int plane = r < G.M ? r * G.S + s : 0;
      va[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                  rA, voff, (unsigned)plane * plane_bytes, 0));
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = wave + 4 * i;
      const int plane = r < G.J ? G.boff[r] + sb : 0;
      vb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                  rB, voff, (unsigned)plane * plane_bytes, 0));
    }
  };

  const long long total_tiles = (long long)G.S * G.tiles_per_seg;
  long long tile = blockIdx.x;
  if (tile < total_tiles) fetch(tile);
  for (; tile < total_tiles; tile += gridDim.x) {
    // registers -> LDS ([row][kLd]: conflict-free for the stores and for the
    // MFMA fragment reads below)
    const float kp_ = keep;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int r = wave + 4 * i;
      la[r * kLd + col] = va[i] * (r < G.M ? kp_ : 0.f);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = wave + 4 * i;
      const float mul = r < G.J ? kp_ : 0.f;
      const float add = (r >= G.J && r < Jt) ? kp_ : 0.f;  // ones column
      lb[r * kLd + col] = fmaf(vb[i], mul, add);
    }
    __syncthreads();
    // software pipeline: the next tile's loads fly while this one multiplies
    if (tile + gridDim.x < total_tiles) fetch(tile + gridDim.x);
    // wave w owns k-pairs [8w, 8w+8) of the tile
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      const int kcol = 2 * (wave * 8 + kp) + (lane >> 5);
      float a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = la[(mb * 32 + (lane & 31)) * kLd + kcol];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const float b = lb[(jb * 32 + (lane & 31)) * kLd + kcol];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b, acc[mb][jb], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // sum the 4 waves through LDS (reuse the tile buffers): [MB*32][NB*32]
  // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *red = lds;
  constexpr int W = NB * 32;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            const int cc = jb * 32 + (lane & 31);
            if (w == 0) red[row * W + cc] = acc[mb][jb][i];
            else red[row * W + cc] += acc[mb][jb][i];
          }
    }
    __syncthreads();
  }
  float *out = G.part + (size_t)blockIdx.x * MB * 32 * W;
  for (int i = tid; i < MB * 32 * W; i += kThreads) out[i] = red[i];
}

// C[m*ldc + j] = sum over workgroups of part[wg][m][j].  64 outputs x 4
// slices of the workgroup range per block: the loads of a slice are coalesced
// over the outputs; every slice is summed in index order in double, the four
// slices are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void planes_gemm_reduce_kernel(
    const float *__restrict__ part, int num_wg, int W, int rows, int M, int Jt,
    float *__restrict__ C, int ldc) {
  __shared__ double sh[4][64];
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + x;
  const bool ok = idx < M * Jt;
  const int m = ok ? idx / Jt : 0, j = ok ? idx % Jt : 0;
  const float *p = part + (size_t)m * W + j;
  const size_t stride = (size_t)rows * W;
  const int per = (num_wg + 3) / 4;
  const int w0 = y * per, w1 = w0 + per < num_wg ? w0 + per : num_wg;
  double acc[4] = {0, 0, 0, 0};
  int w = w0;
  for (; w + 4 <= w1; w += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += (double)p[(size_t)(w + u) * stride];
  }
  for (; w < w1; ++w) acc[0] += (double)p[(size_t)w * stride];
  sh[y][x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (y == 0 && ok)
    C[(size_t)m * ldc + j] = (float)((sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]));
}

template <int MB, int NB>
int launch(const GemmArgs &G, int num_wg, hipStream_t st) {
  const size_t lds = (size_t)(MB * 32 + NB * 32) * kLd * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)planes_gemm_kernel<MB, NB>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm)");
    attr_set = true;
  }
  hipLaunchKernelGGL((planes_gemm_kernel<MB, NB>), dim3(num_wg), dim3(kThreads),
                     lds, st, G);
  return check_launch("planes_gemm");
}

template <int MB>
int launch_nb(const GemmArgs &G, int NB, int num_wg, hipStream_t st) {  // MB == 1
  switch (NB) {
    case 1: return launch<MB, 1>(G, num_wg, st);
    case 2: return launch<MB, 2>(G, num_wg, st);
    case 3: return launch<MB, 3>(G, num_wg, st);
    case 4: return launch<MB, 4>(G, num_wg, st);
    case 5: return launch<MB, 5>(G, num_wg, st);
    default: return launch<MB, 6>(G, num_wg, st);
  }
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_planes_gemm_workspace_floats(int M, int J, int with_ones, int num_wg) {
  const int MB = (M + 31) / 32, NB = (J + (with_ones ? 1 : 0) + 31) / 32;
  return num_wg * MB * 32 * NB * 32;
}

int apg_planes_gemm(const float *A, int M, int S, const float *Bp,
                    const int *boff, int J, int bstride, int sdiv, int bstride2,
                    int with_ones, int b_planes, long long N, float *workspace,
                    int num_wg, float *C, int ldc, apg_stream_t stream) {
  const int Jt = J + (with_ones ? 1 : 0);
  if (!A || !Bp || !boff || !workspace || !C) {
    set_error("apg_planes_gemm: NULL pointer");
    return APG_ERR_ARG;
  }
  const int MB = (M + 31) / 32, NB = (Jt + 31) / 32;
  if (M < 1 || M > 64 || S < 1 || J < 1 || Jt > kMaxNB * 32 || N < 1 ||
      num_wg < 1 || sdiv < 1 || ldc < Jt || (MB == 2 && NB > 4)) {
    set_error("apg_planes_gemm: need 1 <= M <= 64, J + ones <= %d (<= 128 when "
              "M > 32), S, N, num_wg, sdiv >= 1, ldc >= J + ones", kMaxNB * 32);
    return APG_ERR_ARG;
  }
  const long long a_bytes = (long long)M * S * N * 4;
  const long long b_bytes = (long long)b_planes * N * 4;
  if (b_planes < 1 || a_bytes >= (1ll << 32) || b_bytes >= (1ll << 32)) {
    set_error("apg_planes_gemm: operands must be smaller than 4 GiB each "
              "(32-bit buffer offsets); split the batch");
    return APG_ERR_ARG;
  }
  GemmArgs G;
  G.A = A, G.Bp = Bp, G.boff = boff, G.part = workspace;
  G.a_bytes = a_bytes, G.b_bytes = b_bytes;
  G.N = N, G.M = M, G.S = S, G.J = J;
  G.bstride = bstride, G.sdiv = sdiv, G.bstride2 = bstride2;
  G.with_ones = with_ones ? 1 : 0;
  G.tiles_per_seg = (int)((N + kKT - 1) / kKT);
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (MB == 1) {
    e = launch_nb<1>(G, NB, num_wg, st);
  } else {
    switch (NB) {
      case 1: e = launch<2, 1>(G, num_wg, st); break;
      case 2: e = launch<2, 2>(G, num_wg, st); break;
      case 3: e = launch<2, 3>(G, num_wg, st); break;
      default: e = launch<2, 4>(G, num_wg, st); break;
    }
  }
  if (e) return e;
  hipLaunchKernelGGL(planes_gemm_reduce_kernel, dim3((M * Jt + 63) / 64),
                     dim3(256), 0, st, workspace, num_wg, NB * 32, MB * 32, M, Jt,
                     C, ldc);
  return check_launch("planes_gemm_reduce");
}

}  // extern "C"
