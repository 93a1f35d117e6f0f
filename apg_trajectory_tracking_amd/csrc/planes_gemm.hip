// planes_gemm.hip - "planes x planes" reduction GEMM on the matrix cores.
//
// The weight gradients of the in-kernel policies (lstm.hip, mlp.hip) are
//   C[m][j] = sum_{s < S} sum_{n < N} A[(m*S + s)][n] * B[bplane(j, s)][n]
//   bplane(j, s) = bdesc[0][j] + (s / sdiv) * bdesc[1][j] + (s % sdiv) * bdesc[2][j]
// with small M (<= 64), small J (<= 192) and an enormous reduction length
// (S * N = 655 360 ... 5 242 880 at the benchmark batch).  Every operand row is
// a plane of N contiguous floats ("NT" layout: the reduction index is the
// contiguous one).  rocBLAS picks a 16x16 macro-tile without split-K for this
// shape and takes 1.5 ms per call; this kernel streams both operands from HBM
// once and is bound by that stream.  The per-column two-level segment stride
// lets the conv-weight gradient read the sliding reference windows straight
// from the [2H][9][B] reference tensor (segment = (window position, step))
// and the positions before each step from the state planes in the same pass,
// instead of from a materialised [90][H*B] copy.
//
// Structure: the S*N reduction range is cut into tiles of 64 columns;
// workgroups take tiles grid-strided.  A tile ((MB + NB) * 32 operand rows x
// 64 columns) goes global -> LDS by direct-to-LDS DMA, 16 bytes per lane: one
// wave instruction moves a GROUP of 4 rows x 64 columns (1 KiB; 256 contiguous
// bytes per plane), so the big shape needs 12 instructions per wave and tile,
// no staging registers, no ds_write and no VALU.  Inside a group the 16-byte
// chunk of (row r, columns 4c..4c+3) sits in slot r*16 + (c ^ 4r): the source
// address per lane is free, so the swizzle costs nothing and spreads the
// MFMA fragment reads over the banks.  LDS is double-buffered: the DMA of
// tile t+1 is in flight while tile t is multiplied, ONE barrier per tile.
// Each of the 4 waves multiplies 8 of the tile's 32 k-pairs for ALL row /
// column blocks with v_mfma_f32_32x32x2_f32 (exact f32; an A fragment is
// reused by every column block, a B fragment by both row blocks).
// Accumulators stay in registers across tiles; at the end the 4 waves are
// summed through LDS and the workgroup writes one partial C; a second kernel
// adds the partials in a fixed order (deterministic, no float atomics).  The
// optional extra column of row sums (bias gradients) is accumulated on the
// VALU from the A fragments the lanes read anyway.
#include "apg_device.h"

namespace apg {
namespace {

constexpr int kKT = 64;        // reduction elements per tile
constexpr int kGS = 4 * kKT + 4;  // floats per 4-row group in LDS (16 B pad)
constexpr int kMaxNB = 6;      // column blocks of 32 (J + ones <= 192)
constexpr int kThreads = 256;
constexpr unsigned kDeadOff = 0x80000000u;  // beyond any operand (< 2 GiB each)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lds_ptr;

struct GemmArgs {
  const float *A, *Bp;
  const int *bdesc;  // [3][J]: plane offset, stride per s / sdiv, stride per s % sdiv
  float *part;  // [gridDim.x][MB*32][NB*32]
  long long N, a_bytes, b_bytes;
  int M, S, J, sdiv, with_ones, tiles_per_seg;
};

// LDS float index of element (row, col) of a tile buffer
__device__ __forceinline__ int tile_index(int row, int col) {
  const int r = row & 3;
  return (row >> 2) * kGS + (r << 6) + ((((col >> 2) ^ (r << 2))) << 2) + (col & 3);
}

// One workgroup's share of a product: tiles bid, bid + nb, ... of `G`; the
// partial C ([MB*32][NB*32]) goes to `out`.
template <int MB, int NB>
__device__ __forceinline__ void gemm_body(const GemmArgs &G, int bid, int nb,
                                          float *out, float *lds) {
  constexpr int NG = (MB + NB) * 8;  // 4-row groups per tile
  constexpr int GI = NG / 4;         // groups (= DMA instructions) per wave
  constexpr int BUF = NG * kGS;      // floats per tile buffer
  constexpr int W = NB * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[MB][NB];
  float rsum[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    rsum[mb] = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][jb][i] = 0.f;
  }
  // padding rows are never written by the DMA: zero both buffers once
  for (int i = tid; i < 2 * BUF; i += kThreads) lds[i] = 0.f;

  const unsigned plane_bytes = (unsigned)(G.N * 4);
  const auto rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.A), 0, (unsigned)G.a_bytes, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.Bp), 0, (unsigned)G.b_bytes, 0x00020000);
  // this lane's slot of a group: row (lane >> 4), source column chunk cgs
  const int r_in = lane >> 4, cgs = (lane & 15) ^ (r_in << 2);
  unsigned rowoff[GI];  // byte offset of the lane's row + chunk (segment 0)
  unsigned bs1[NB * 2], bs2[NB * 2];  // B rows: bytes per s / sdiv, per s % sdiv
#pragma unroll
  for (int i = 0; i < GI; ++i) {
    const int row = (wave + 4 * i) * 4 + r_in;
    if (i < MB * 2) {
      rowoff[i] = row < G.M ? (unsigned)(row * G.S) * plane_bytes + cgs * 16 : kDeadOff;
    } else {
      const int j = row - MB * 32, jc = j < G.J ? j : 0;
      rowoff[i] = j < G.J ? (unsigned)G.bdesc[jc] * plane_bytes + cgs * 16 : kDeadOff;
      bs1[i - MB * 2] = j < G.J ? (unsigned)G.bdesc[G.J + jc] * plane_bytes : 0u;
      bs2[i - MB * 2] = j < G.J ? (unsigned)G.bdesc[2 * G.J + jc] * plane_bytes : 0u;
    }
  }
  __syncthreads();

  auto issue = [&](long long tile, int p) {  // DMA of one tile into buffer p
    // segment fastest: the workgroups running at the same time then work on a
    // few column tiles across ALL segments, so B rows shared by segments (the
    // sliding windows of the conv product) are fetched from HBM once
    const int s = (int)(tile % G.S);
    const unsigned colb = (unsigned)(tile / G.S) * (kKT * 4);
    const unsigned sa = (unsigned)s * plane_bytes;
    const unsigned s1 = (unsigned)(s / G.sdiv), s2 = (unsigned)(s % G.sdiv);
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const int gi = wave + 4 * i;
      const bool isA = i < MB * 2;
      const int first = isA ? gi * 4 : gi * 4 - MB * 32;  // first row of the group
      if (first < (isA ? G.M : G.J)) {                    // wave-uniform
        if (isA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rA, (lds_ptr)(lds + p * BUF + gi * kGS), 16, (int)(rowoff[i] + colb),
              (int)sa, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rB, (lds_ptr)(lds + p * BUF + gi * kGS), 16,
              (int)(rowoff[i] + colb + s1 * bs1[i < MB * 2 ? 0 : i - MB * 2] +
                    s2 * bs2[i < MB * 2 ? 0 : i - MB * 2]),
              0, 0, 0);
      }
    }
  };

  const int lr = lane & 31, kh = lane >> 5;
  const int lane_base =
      (lr >> 2) * kGS + ((lr & 3) << 6) + ((wave ^ (lr & 3)) << 4) + kh;
  const long long total_tiles = (long long)G.S * G.tiles_per_seg;
  long long tile = bid;
  int p = 0;
  if (tile < total_tiles) issue(tile, 0);
  for (; tile < total_tiles; tile += nb, p ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile landed (all waves); buffer p^1 is free again
    const long long n0 = (tile / G.S) * kKT;
    if (G.N - n0 < kKT) {  // ragged last tile of a segment: zero A beyond N
      const int rem = (int)(G.N - n0);
      for (int e = tid; e < MB * 32 * kKT; e += kThreads)
        if ((e & 63) >= rem) lds[p * BUF + tile_index(e >> 6, e & 63)] = 0.f;
      __syncthreads();
    }
    if (tile + nb < total_tiles) issue(tile + nb, p ^ 1);
    const float *bp = lds + p * BUF + lane_base;
    // wave w owns k-pairs [8w, 8w+8) of the tile
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      float a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        a[mb] = bp[mb * 8 * kGS + 2 * kp];
        rsum[mb] += a[mb];
      }
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const float b = bp[(MB + jb) * 8 * kGS + 2 * kp];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b, acc[mb][jb], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  // sum the 4 waves through LDS (reuse the tile buffers): [MB*32][NB*32]
  // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float *red = lds;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) rsum[mb] += __shfl_xor(rsum[mb], 32, 64);
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
    if (wave == w && G.with_ones && lane < 32) {  // row sums -> column J
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) red[(mb * 32 + lane) * W + G.J] += rsum[mb];
    }
    __syncthreads();
  }
  for (int i = tid; i < MB * 32 * W; i += kThreads) out[i] = red[i];
}

template <int MB, int NB>
__global__ __launch_bounds__(kThreads) void planes_gemm_kernel(GemmArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  gemm_body<MB, NB>(G, blockIdx.x, gridDim.x,
                    G.part + (size_t)blockIdx.x * MB * 32 * NB * 32, lds);
}

// Several products in ONE launch (the weight gradients of a training step):
// every workgroup belongs to one problem (range [wg0[p], wg0[p+1])), all
// problems use the 64 x 128 accumulator shape.  A second launch reduces every
// problem's partials.
constexpr int kMaxGroup = 8;
struct GroupArgs {
  GemmArgs g[kMaxGroup];
  int wg0[kMaxGroup + 1];
  float *C[kMaxGroup], *bias[kMaxGroup];
  int ldc[kMaxGroup];
  int n;
};

__global__ __launch_bounds__(kThreads) void planes_gemm_grouped_kernel(GroupArgs GA) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int p = 0;
  while (p + 1 < GA.n && (int)blockIdx.x >= GA.wg0[p + 1]) ++p;
  gemm_body<2, 4>(GA.g[p], blockIdx.x - GA.wg0[p], GA.wg0[p + 1] - GA.wg0[p],
                  GA.g[0].part + (size_t)blockIdx.x * 64 * 128, lds);
}

// C[m*ldc + j] = sum over workgroups of part[wg][m][j].  64 outputs x 4
// slices of the workgroup range per block: the loads of a slice are coalesced
// over the outputs; every slice is summed in index order in double, the four
// slices are combined in a fixed order (deterministic).
__global__ __launch_bounds__(256) void planes_gemm_reduce_kernel(
    const float *__restrict__ part, int num_wg, int W, int rows, int M, int Jt,
    float *__restrict__ C, int ldc, int J, float *__restrict__ bias_out) {
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
  if (y == 0 && ok) {
    const float v = (float)((sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]));
    if (j == J && bias_out) bias_out[m] = v;  // row sums to their own vector
    else C[(size_t)m * ldc + j] = v;
  }
}

__global__ __launch_bounds__(256) void planes_gemm_grouped_reduce_kernel(GroupArgs GA) {
  __shared__ double sh[4][64];
  const int p = blockIdx.y;
  const GemmArgs &G = GA.g[p];
  const int Jt = G.J + G.with_ones;
  const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + x;
  const bool ok = idx < G.M * Jt;
  const int m = ok ? idx / Jt : 0, j = ok ? idx % Jt : 0;
  const float *part = GA.g[0].part + (size_t)GA.wg0[p] * 64 * 128 + (size_t)m * 128 + j;
  const int num_wg = GA.wg0[p + 1] - GA.wg0[p];
  const int per = (num_wg + 3) / 4;
  const int w0 = y * per, w1 = w0 + per < num_wg ? w0 + per : num_wg;
  double acc = 0;
  for (int w = w0; w < w1; ++w) acc += (double)part[(size_t)w * 64 * 128];
  sh[y][x] = acc;
  __syncthreads();
  if (y == 0 && ok) {
    const float v = (float)((sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x]));
    if (j == G.J && GA.bias[p]) GA.bias[p][m] = v;
    else GA.C[p][(size_t)m * GA.ldc[p] + j] = v;
  }
}

template <int MB, int NB>
int launch(const GemmArgs &G, int num_wg, hipStream_t st) {
  const size_t lds = (size_t)2 * (MB + NB) * 8 * kGS * sizeof(float);
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
                    const int *bdesc, int J, int sdiv, int with_ones,
                    int b_planes, long long N, float *workspace, int num_wg,
                    float *C, int ldc, float *bias_out, apg_stream_t stream) {
  const int Jt = J + (with_ones ? 1 : 0);
  if (!A || !Bp || !bdesc || !workspace || !C) {
    set_error("apg_planes_gemm: NULL pointer");
    return APG_ERR_ARG;
  }
  const int MB = (M + 31) / 32, NB = (Jt + 31) / 32;
  if (M < 1 || M > 64 || S < 1 || J < 1 || Jt > kMaxNB * 32 || N < 1 ||
      num_wg < 1 || sdiv < 1 || ldc < (bias_out ? J : Jt) || (MB == 2 && NB > 4)) {
    set_error("apg_planes_gemm: need 1 <= M <= 64, J + ones <= %d (<= 128 when "
              "M > 32), S, N, num_wg, sdiv >= 1, ldc >= J + ones", kMaxNB * 32);
    return APG_ERR_ARG;
  }
  const long long a_bytes = (long long)M * S * N * 4;
  const long long b_bytes = (long long)b_planes * N * 4;
  if (b_planes < 1 || a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) {
    set_error("apg_planes_gemm: operands must be smaller than 2 GiB each "
              "(32-bit buffer offsets); split the batch");
    return APG_ERR_ARG;
  }
  GemmArgs G;
  G.A = A, G.Bp = Bp, G.bdesc = bdesc, G.part = workspace;
  G.a_bytes = a_bytes, G.b_bytes = b_bytes;
  G.N = N, G.M = M, G.S = S, G.J = J;
  G.sdiv = sdiv;
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
                     C, ldc, J, with_ones ? bias_out : nullptr);
  return check_launch("planes_gemm_reduce");
}

int apg_planes_gemm_grouped(const ApgGemmProblem *problems, int n,
                            float *workspace, int num_wg, apg_stream_t stream) {
  if (!problems || !workspace || n < 1 || n > kMaxGroup || num_wg < n) {
    set_error("apg_planes_gemm_grouped: need 1 <= n <= %d problems, num_wg >= n",
              kMaxGroup);
    return APG_ERR_ARG;
  }
  GroupArgs GA;
  GA.n = n;
  double cost[kMaxGroup], total = 0;
  for (int p = 0; p < n; ++p) {
    const ApgGemmProblem &q = problems[p];
    const int Jt = q.J + (q.with_ones ? 1 : 0);
    if (!q.A || !q.B || !q.bdesc || !q.C || q.M < 1 || q.M > 64 || q.S < 1 ||
        q.J < 1 || Jt > 128 || q.N < 1 || q.sdiv < 1 ||
        q.ldc < (q.bias_out && q.with_ones ? q.J : Jt)) {
      set_error("apg_planes_gemm_grouped: problem %d: need M <= 64, J + ones <= "
                "128, valid pointers and ldc", p);
      return APG_ERR_ARG;
    }
    const long long a_bytes = (long long)q.M * q.S * q.N * 4;
    const long long b_bytes = (long long)q.b_planes * q.N * 4;
    if (q.b_planes < 1 || a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) {
      set_error("apg_planes_gemm_grouped: problem %d: operands must be < 2 GiB", p);
      return APG_ERR_ARG;
    }
    GemmArgs &G = GA.g[p];
    G.A = q.A, G.Bp = q.B, G.bdesc = q.bdesc, G.part = workspace;
    G.a_bytes = a_bytes, G.b_bytes = b_bytes;
    G.N = q.N, G.M = q.M, G.S = q.S, G.J = q.J, G.sdiv = q.sdiv;
    G.with_ones = q.with_ones ? 1 : 0;
    G.tiles_per_seg = (int)((q.N + kKT - 1) / kKT);
    GA.C[p] = q.C, GA.ldc[p] = q.ldc;
    GA.bias[p] = q.with_ones ? q.bias_out : nullptr;
    // measured per workgroup and tile: ~0.6 us fixed + 12.5 ns per operand row
    cost[p] = (double)q.S * G.tiles_per_seg * (q.M + Jt + 48);
    total += cost[p];
  }
  // workgroups in proportion to the time each problem needs
  int given = 0;
  for (int p = 0; p < n; ++p) {
    int w = (int)(cost[p] / total * (num_wg - n)) + 1;
    GA.wg0[p] = given;
    given += w;
  }
  GA.wg0[n] = given;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)2 * (2 + 4) * 8 * kGS * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)planes_gemm_grouped_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm_grouped)");
    attr_set = true;
  }
  hipLaunchKernelGGL(planes_gemm_grouped_kernel, dim3(given), dim3(kThreads), lds, st,
                     GA);
  if (int e = check_launch("planes_gemm_grouped")) return e;
  hipLaunchKernelGGL(planes_gemm_grouped_reduce_kernel, dim3((64 * 128 + 63) / 64, n),
                     dim3(256), 0, st, GA);
  return check_launch("planes_gemm_grouped_reduce");
}

}  // extern "C"
