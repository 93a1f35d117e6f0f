// planes_gemm.hip - "planes x planes" reduction GEMM on the matrix cores.
//
// The weight gradients of the in-kernel policies (lstm.hip, mlp_wing.hip; mlp_planes.hip of the test library) are
//   C[m][j] = sum_{s < S} sum_{n < N} A[(m*S + s)][n] * B[bplane(j, s)][n]
//   bplane(j, s) = bdesc[0][j] + (s / sdiv) * bdesc[1][j] + (s % sdiv) * bdesc[2][j]
// with small M (<= 64), small J (<= 192) and an enormous reduction length
// (S * N = 655 360 ... 5 242 880 at the benchmark batch).  Every operand row is
// a plane of N contiguous floats ("NT" layout: the reduction index is the
// contiguous one).  rocBLAS picks a 16x16 macro-tile without split-K for this
// shape and takes 1.5 ms per call; this kernel streams both operands from HBM
// once and is bound by that stream.  Two kernels:
//  * plain products (S = 1; six of the seven of an autoregressive step):
//    planes_gemm_stream_kernel - global -> registers -> v_mfma_f32_16x16x32_bf16
//    on three-term bf16 splits of the fp32 operands,
//    split-K over all waves, no LDS tile (described at stream_body below);
//  * segmented products (S > 1) with 16 or more B rows, and the grouped launch
//    for short planes: planes_gemm_kernel, the LDS-tile kernel described next
//    (a segmented product with fewer B rows - the 3 position planes of the
//    conv gradient - streams too: use_stream).  Its per-column
//    two-level segment stride lets the conv-weight gradient read the sliding
//    reference windows straight from the [2H][9][B] reference tensor (segment
//    = (window position, step)) and the positions before each step from the
//    state planes in the same pass, instead of from a materialised
//    [90][H*B] copy.
//
// LDS-tile kernel: the S*N reduction range is cut into tiles of 64 columns;
// workgroups take tiles grid-strided.  A tile ((MB + NB) * 32 operand rows x
// 64 columns) goes global -> LDS by direct-to-LDS DMA, 16 bytes per lane: one
// wave instruction moves a GROUP of 4 rows x 64 columns (1 KiB; 256 contiguous
// bytes per plane), so the big shape needs 12 instructions per wave and tile,
// no staging registers, no ds_write and no VALU.  Inside a group the 16-byte
// chunk of (row r, columns 4c..4c+3) sits in slot r*16 + (c ^ 4r): the source
// address per lane is free, so the swizzle costs nothing and spreads the
// MFMA fragment reads over the banks.  LDS is a ring of ST = 2 tile buffers:
// while tile t is multiplied the DMA of tile t+1 is in flight; every wave
// issues the same number of DMA instructions per tile (rows that do not exist
// get an out-of-range offset, which costs an issue slot and no memory
// traffic), so "the oldest tile has landed" is a vmcnt immediate followed by
// ONE barrier per tile (a bare s_barrier, no fence).
// Each of the 4 waves multiplies 8 of the tile's 32 k-pairs for ALL row /
// column blocks with v_mfma_f32_32x32x2_f32 (exact f32; an A fragment is
// reused by every column block, a B fragment by both row blocks).
// Accumulators stay in registers across tiles; at the end the 4 waves are
// summed through LDS and the workgroup writes one partial C; a second kernel
// adds the partials in a fixed order (deterministic, no float atomics) - one
// launch for all products of a training step (apg_planes_gemm_multi).  The
// optional extra column of row sums (bias gradients) is accumulated on the
// VALU from the A fragments the lanes read anyway.
// Operand offsets are unsigned 32-bit buffer offsets: each operand (for B:
// the span of planes the caller passes) must stay below 4 GiB.
#include "apg_device.h"

namespace apg {
namespace {

// Depth of the LDS tile ring.  A third buffer (two tiles in flight during a
// multiply, where 160 KB hold it) measured no faster than two on MI355X - the
// waves that multiply are also the ones whose DMA issue blocks on the CU's
// request queue, and a separate loader wave was slower still - so 2 is built.
#ifndef APG_GEMM_ST_MAX
#define APG_GEMM_ST_MAX 2
#endif
// 1: apg_planes_gemm / apg_planes_gemm_multi run the register-streaming kernel
// (stream_body); 0: the LDS-tile kernel (experiments; the grouped entry point
// for short planes always uses the LDS-tile kernel).
#ifndef APG_GEMM_STREAM
#define APG_GEMM_STREAM 1
#endif

constexpr int kKT = 64;        // reduction elements per tile
constexpr int kGS = 4 * kKT + 4;  // floats per 4-row group in LDS (16 B pad)
constexpr int kMaxNB = 6;      // column blocks of 32 (J + ones <= 192)
constexpr int kThreads = 256;
constexpr unsigned kDeadOff = 0xfffffff0u;  // beyond any operand
constexpr long long kMaxOperandBytes = 0xffffff00ll;  // < both kernels' dead offsets
constexpr int kLdsBytes = 160 * 1024;

// LDS footprint of a tile shape (ring depth: see APG_GEMM_ST_MAX)
template <int MB, int NB>
struct Shape {
  static constexpr int NG = (MB + NB) * 8;  // 4-row groups per tile
  static constexpr int BUF = NG * kGS;      // floats per tile buffer
  static constexpr int ST = (APG_GEMM_ST_MAX >= 3 && 3 * BUF * 4 <= kLdsBytes) ? 3 : 2;
  static constexpr int lds_bytes = ST * BUF * 4;
};

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
  constexpr int NG = Shape<MB, NB>::NG;
  constexpr int GI = NG / 4;         // groups (= DMA instructions) per wave
  constexpr int BUF = Shape<MB, NB>::BUF;
  constexpr int ST = Shape<MB, NB>::ST;
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
  // padding rows only ever receive out-of-range (zero) data: zero the ring once
  for (int i = tid; i < ST * BUF; i += kThreads) lds[i] = 0.f;

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

  // tile t = (segment s, column tile ct), t = ct * S + s: segment fastest, so
  // the workgroups running at the same time work on a few column tiles across
  // ALL segments and B rows shared by segments (the sliding windows of the
  // conv product) are fetched once.  "Once" per L2: workgroup b runs on XCD
  // b % 8, so a segmented product gives XCD x the column tiles ct = 8 c + x -
  // all segments of a column tile then share ONE XCD's L2 instead of pulling
  // the windows into all eight (737 -> 5xx MB fetched for the conv product).
  // The DMA front and the multiply walk their own (s, c) by the stride of the
  // XCD's share of the grid.
  const int X = (G.S > 1 && nb % 8 == 0) ? 8 : 1;
  const int xcd = bid % X, lnb = nb / X;
  const int ds = lnb % G.S, dc = lnb / G.S;
  auto advance = [&](int &s_, int &c_) {
    s_ += ds, c_ += dc;
    if (s_ >= G.S) s_ -= G.S, ++c_;
  };
  auto col_tile = [&](int c_) { return c_ * X + xcd; };
  // Every wave issues GI instructions per tile (rows that do not exist and
  // tiles past the end get an out-of-range offset: zeros, no traffic), so
  // "the oldest tile has landed" is a vmcnt immediate.
  auto issue = [&](int s_, int c_, int q) {  // DMA of one tile into buffer q
    const int ct_ = col_tile(c_);
    const bool live = ct_ < G.tiles_per_seg;
    const unsigned colb = (unsigned)ct_ * (kKT * 4);
    const unsigned sa = (unsigned)s_ * plane_bytes;
    const unsigned s1 = (unsigned)(s_ / G.sdiv), s2 = (unsigned)(s_ % G.sdiv);
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const int gi = wave + 4 * i;
      unsigned off;
      if (i < MB * 2) {
        off = rowoff[i] + colb + sa;
      } else {
        off = rowoff[i] + colb + s1 * bs1[i - MB * 2] + s2 * bs2[i - MB * 2];
      }
      off = (live && rowoff[i] != kDeadOff) ? off : kDeadOff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          i < MB * 2 ? rA : rB, (lds_ptr)(lds + q * BUF + gi * kGS), 16, (int)off, 0,
          0, 0);
    }
  };

  const int lr = lane & 31, kh = lane >> 5;
  const int lane_base =
      (lr >> 2) * kGS + ((lr & 3) << 6) + ((wave ^ (lr & 3)) << 4) + kh;
  int s_m = (bid / X) % G.S, ct_m = (bid / X) / G.S;  // the tile being multiplied
  int s_i = s_m, ct_i = ct_m;                         // the DMA front
#pragma unroll
  for (int q = 0; q < ST - 1; ++q) {
    issue(s_i, ct_i, q);
    advance(s_i, ct_i);
  }
  int p = 0;
  for (; col_tile(ct_m) < G.tiles_per_seg; advance(s_m, ct_m)) {
    // the oldest of the ST-1 tiles in flight has landed (this wave's part) ...
    if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and everybody's; the buffer multiplied last is free.  A bare
    // s_barrier: __syncthreads() carries a fence that makes the compiler wait
    // for ALL outstanding DMAs.  Nothing else needs the fence: the LDS reads
    // of the last multiply have been consumed by its MFMAs.
    __builtin_amdgcn_s_barrier();
    const long long n0 = (long long)col_tile(ct_m) * kKT;
    if (G.N - n0 < kKT) {  // ragged last tile of a segment: zero A AND B beyond
      // N (what lies there is the next plane, or another tensor: 0 x Inf = NaN)
      const int rem = (int)(G.N - n0);
      for (int e = tid; e < (MB + NB) * 32 * kKT; e += kThreads)
        if ((e & 63) >= rem) lds[p * BUF + tile_index(e >> 6, e & 63)] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    issue(s_i, ct_i, p == 0 ? ST - 1 : p - 1);
    advance(s_i, ct_i);
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
    p = p + 1 == ST ? 0 : p + 1;
  }
  // the DMAs issued past the end (zeros) must be done before the ring is reused
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// ---------------------------------------------------------------------------
// The long-plane kernel: operands go global -> registers -> matrix core, no
// LDS tile and no barrier in the main loop.  Lane (i, g) = (l & 15, l >> 4)
// loads 8 CONSECUTIVE floats of row i's plane (two 16-byte loads at column
// 32 c + 8 g): the lane's 8 k-slots of a 16 x 16 x 32 matrix instruction
// (split3 / mfma_bf16 below); 16 rows x 128 contiguous bytes per load pair,
// every operand element is fetched exactly once, and each wave streams its own
// chunks of 32 columns
// (split-K over all waves of the grid) with the next chunk's loads in flight
// while this one multiplies (plain products, and segmented ones with a few B
// rows; the segmented window product re-reads its windows per segment and is
// faster through the LDS tiles).
// One wave holds ALL MB x NB accumulator tiles of
// 16 x 16 (4 registers each); the 4 waves of a workgroup are summed through
// LDS at the end and the partial C goes to the same second stage as above.
// Row sums (bias gradients) are accumulated from the A operands on the VALU.
constexpr unsigned kDeadOff2 = 0xffffff00u;  // + the 16-byte immediate: still out of range
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The stream kernel's matrix instruction.  v_mfma_f32_16x16x4_f32 (rounds 1-2)
// took 8 instructions of 32 cycles per chunk and tile: for the 4 x 7 shape
// 7 168 cycles per chunk against ~11 500 of memory time at the HBM rate, and
// the two only overlap in steady state - the short planes of the grouped
// launch (6-9 chunks per wave) paid their sum.  With v_mfma_f32_16x16x32_bf16
// lane l supplies A[l & 15][k-slots 8 (l >> 4) + j] - exactly the 8 consecutive
// floats the lane has loaded - so a chunk is ONE instruction (~17 cycles) per
// product of terms.  Every fp32 operand is cut into THREE bf16 terms by
// rounding to nearest, x = h + m + l + (<= 2^-24 |x|) with |m| <= 2^-8 |x|,
// |l| <= 2^-16 |x| (bf16 has fp32's exponent, so cotangents need no scaling),
// and a tile takes the six products of weight >= 2^-16 (h h, h m, m h, m m,
// h l, l h): what is dropped (m l, l m, l l, the terms' own remainders) is
// <= 2^-22 of |a||b| per product in the worst case and 2^-24 - the rounding of
// an fp32 multiply - typically (tests/test_host_cpu.py emulates it).
// Measured (profiles/r03_gemm_stream_bf16x3.txt): grouped launch of the
// concurrent step 60 -> 42 us, 4 x 7 product of the autoregressive step
// 113 -> 87 us (5.3 TB/s), full-size parity unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Terms {
  u32x4 t[3];  // high, middle, low terms of 8 values, packed in pairs
};
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// (x0, x1) rounded to bf16 (v_cvt_pk_bf16_f32, to nearest even), packed; `f0`,
// `f1`: the rounded values back in fp32
__device__ __forceinline__ unsigned round_pair(float x0, float x1, float &f0, float &f1) {
  const bf16x2 v = {(__bf16)x0, (__bf16)x1};
  const unsigned p = __builtin_bit_cast(unsigned, v);
  f0 = __builtin_bit_cast(float, p << 16);
  f1 = __builtin_bit_cast(float, p & 0xffff0000u);
  return p;
}
__device__ __forceinline__ Terms split3(u32x4 lo4, u32x4 hi4) {
  Terms o;
  const f32x4 v[2] = {__builtin_bit_cast(f32x4, lo4), __builtin_bit_cast(f32x4, hi4)};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = v[q >> 1][2 * (q & 1)], x1 = v[q >> 1][2 * (q & 1) + 1];
    float h0, h1, m0, m1, l0, l1;
    o.t[0][q] = round_pair(x0, x1, h0, h1);
    const float r0 = x0 - h0, r1 = x1 - h1;           // exact, |r| <= 2^-8 |x|
    o.t[1][q] = round_pair(r0, r1, m0, m1);
    o.t[2][q] = round_pair(r0 - m0, r1 - m1, l0, l1); // exact difference, rounded once more
  }
  return o;
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MB, int NB>
__device__ __forceinline__ void stream_body(const GemmArgs &G, int bid, int nb,
                                            float *out, float *red) {
  constexpr int W = NB * 16 + 1;  // partial row pitch; column J holds the row sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  f32x4 acc[MB][NB];
  float rsum[MB];
#pragma unroll
  for (int rb = 0; rb < MB; ++rb) {
    rsum[rb] = 0.f;
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const unsigned plane_bytes = (unsigned)(G.N * 4);
  const auto rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.A), 0, (unsigned)G.a_bytes, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(G.Bp), 0, (unsigned)G.b_bytes, 0x00020000);
  // Segments (S > 1, small products only: see use_stream): row m of A is the S
  // planes (m S + s), B row j of segment s the plane bplane(j, s); a chunk of 32
  // columns never crosses a segment (chunk index = segment * chunks per
  // segment + chunk of the segment).
  unsigned offA[MB], offB[NB], bs1[NB], bs2[NB];
#pragma unroll
  for (int rb = 0; rb < MB; ++rb) {
    const int row = rb * 16 + i16;
    offA[rb] = row < G.M ? (unsigned)(row * G.S) * plane_bytes + g * 32 : kDeadOff2;
  }
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    const int j = cb * 16 + i16, jc = j < G.J ? j : 0;
    offB[cb] = j < G.J ? (unsigned)G.bdesc[jc] * plane_bytes + g * 32 : kDeadOff2;
    bs1[cb] = G.S > 1 ? (unsigned)G.bdesc[G.J + jc] * plane_bytes : 0u;
    bs2[cb] = G.S > 1 ? (unsigned)G.bdesc[2 * G.J + jc] * plane_bytes : 0u;
  }
  // the waves of the grid take chunks of 32 columns wave-strided: at any time
  // the grid reads one contiguous stretch of every plane
  const int cps1 = (int)((G.N + 31) / 32);   // chunks of one segment
  const int cps = cps1 * G.S;
  const int nw = nb * 4;
  int cc = bid * 4 + wave;
  auto load = [&](u32x4 (&fa)[MB][2], u32x4 (&fb)[NB][2]) {
    const bool live = cc < cps;
    const int s = G.S > 1 ? cc / cps1 : 0, c = cc - s * cps1;
    const unsigned colb = (unsigned)c * 128u, sa = (unsigned)s * plane_bytes;
    const unsigned s1 = (unsigned)(s / G.sdiv), s2 = (unsigned)(s % G.sdiv);
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
      const unsigned off =
          (live && offA[rb] != kDeadOff2) ? offA[rb] + sa + colb : kDeadOff2;
      fa[rb][0] = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)off, 0, 0);
      fa[rb][1] = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)off + 16, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const unsigned off = (live && offB[cb] != kDeadOff2)
                               ? offB[cb] + s1 * bs1[cb] + s2 * bs2[cb] + colb
                               : kDeadOff2;
      fb[cb][0] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)off, 0, 0);
      fb[cb][1] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)off + 16, 0, 0);
    }
  };
  auto multiply = [&](u32x4 (&fa)[MB][2], u32x4 (&fb)[NB][2], int cc_) {
    cc_ = G.S > 1 ? cc_ % cps1 : cc_;   // the chunk of its segment
    const int n0 = cc_ * 32 + g * 8;  // this lane's first column of the chunk
    if ((long long)cc_ * 32 + 32 > G.N) {  // ragged last chunk: zero A AND B
      // beyond N (the columns there belong to the next plane or to another
      // tensor; a non-finite value against a zeroed A would still be NaN)
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (n0 + m >= G.N) {
#pragma unroll
          for (int rb = 0; rb < MB; ++rb) fa[rb][m >> 2][m & 3] = 0u;
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) fb[cb][m >> 2][m & 3] = 0u;
        }
    }
    // row sums from the exact fp32 values, then the chunk's 8 k-slots per lane
    // as three bf16 terms each and six products per tile
    Terms ta[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
      const f32x4 v0 = __builtin_bit_cast(f32x4, fa[rb][0]);
      const f32x4 v1 = __builtin_bit_cast(f32x4, fa[rb][1]);
      rsum[rb] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
      ta[rb] = split3(fa[rb][0], fa[rb][1]);
    }
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      const Terms tb = split3(fb[cb][0], fb[cb][1]);
      // smallest products first; the row blocks take turns, so that an
      // instruction does not wait for the one before it
      constexpr int kTa[6] = {2, 0, 1, 1, 0, 0}, kTb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int rb = 0; rb < MB; ++rb)
          acc[rb][cb] = mfma_bf16(ta[rb].t[kTa[t]], tb.t[kTb[t]], acc[rb][cb]);
    }
  };
  // two register sets: the loads of chunk c+1 are in flight while chunk c
  // multiplies (the compiler places the vmcnt waits)
  u32x4 fa0[MB][2], fb0[NB][2], fa1[MB][2], fb1[NB][2];
  load(fa0, fb0);
  while (cc < cps) {
    const int c0 = cc;
    cc += nw;
    load(fa1, fb1);
    multiply(fa0, fb0, c0);
    if (cc >= cps) break;
    const int c1 = cc;
    cc += nw;
    load(fa0, fb0);
    multiply(fa1, fb1, c1);
  }
  // sum the 4 waves through LDS: every wave writes its accumulators to its own
  // [MB*16][W] region at the same time, then all threads add the four regions
  // in wave order (fixed order: deterministic).  (Four read-modify-write passes
  // over one region, wave after wave, cost ~5 us per launch for 28 tiles.)
  // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
  constexpr int kRegion = MB * 16 * W;
  float *mine = red + wave * kRegion;
#pragma unroll
  for (int rb = 0; rb < MB; ++rb) {
    rsum[rb] += __shfl_xor(rsum[rb], 16, 64);
    rsum[rb] += __shfl_xor(rsum[rb], 32, 64);
  }
#pragma unroll
  for (int rb = 0; rb < MB; ++rb) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        mine[(rb * 16 + 4 * g + r) * W + cb * 16 + i16] = acc[rb][cb][r];
    if (lane < 16) mine[(rb * 16 + lane) * W + NB * 16] = 0.f;
  }
  // row sums -> column J (a dead column of the accumulators, or the extra one):
  // same wave, LDS operations complete in order
  __builtin_amdgcn_wave_barrier();
  if (G.with_ones && lane < 16) {
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) mine[(rb * 16 + lane) * W + G.J] = rsum[rb];
  }
  __syncthreads();
  for (int i = tid; i < kRegion; i += kThreads)
    out[i] = ((red[i] + red[kRegion + i]) + red[2 * kRegion + i]) + red[3 * kRegion + i];
}

template <int MB, int NB>
__global__ __launch_bounds__(kThreads) void planes_gemm_stream_kernel(GemmArgs G) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stream_body<MB, NB>(G, blockIdx.x, gridDim.x,
                      G.part + (size_t)blockIdx.x * MB * 16 * (NB * 16 + 1), lds);
}

// All plain products of a training step in ONE launch: every workgroup belongs
// to one product (range [wg0[p], wg0[p+1]), sized by the bytes the product
// streams) and runs that product's block shape.  For short planes (one column
// per trajectory) seven separate launches are mostly launch ramp and tail.
constexpr int kMaxGroup = 8;
struct StreamGroupArgs {
  GemmArgs g[kMaxGroup];   // .part = the product's own partial region
  int wg0[kMaxGroup + 1];
  int shape[kMaxGroup];    // MB * 16 + NB (blocks of 16)
  int n;
};

__global__ __launch_bounds__(kThreads) void planes_gemm_stream_grouped_kernel(
    StreamGroupArgs GA) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int p = 0;
  while (p + 1 < GA.n && (int)blockIdx.x >= GA.wg0[p + 1]) ++p;
  const GemmArgs &G = GA.g[p];
  const int bid = blockIdx.x - GA.wg0[p], nb = GA.wg0[p + 1] - GA.wg0[p];
#define APG_STREAM_CASE(MB, NB)                                                  \
  case MB * 16 + NB:                                                             \
    stream_body<MB, NB>(G, bid, nb,                                              \
                        G.part + (size_t)bid * MB * 16 * (NB * 16 + 1), lds);    \
    break;
  switch (GA.shape[p]) {
    APG_STREAM_CASE(1, 1) APG_STREAM_CASE(1, 2) APG_STREAM_CASE(1, 4)
    APG_STREAM_CASE(1, 7) APG_STREAM_CASE(1, 8) APG_STREAM_CASE(1, 12)
    APG_STREAM_CASE(2, 1) APG_STREAM_CASE(2, 2) APG_STREAM_CASE(2, 4)
    APG_STREAM_CASE(2, 7) APG_STREAM_CASE(2, 8) APG_STREAM_CASE(2, 12)
    APG_STREAM_CASE(4, 1) APG_STREAM_CASE(4, 2) APG_STREAM_CASE(4, 4)
    APG_STREAM_CASE(4, 7) APG_STREAM_CASE(4, 8)
    default: break;
  }
#undef APG_STREAM_CASE
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

// C[m*ldc + j] = sum over workgroups of part[wg][m][j], for up to kMaxGroup
// products per launch (blockIdx.y).  A block owns 32 outputs and cuts the
// workgroup range into 32 slices: the loads of a slice are coalesced over the
// outputs, every slice is summed in index order in double and the slices are
// combined in index order (deterministic).
struct ReduceItem {
  const float *part;
  float *C, *bias;
  int num_wg, W, rows, M, Jt, J, ldc;
};
struct ReduceArgs {
  ReduceItem it[kMaxGroup];
};

// (16 / 64 / 128 outputs per block measured 0.8-1.3 x / 1.3-1.8 x / 1.7-2.1 x the time)
constexpr int kRedX = 32, kRedY = 1024 / kRedX;  // outputs x slices per block
__global__ __launch_bounds__(1024) void planes_gemm_reduce_kernel(ReduceArgs R) {
  __shared__ double sh[kRedY][kRedX + 1];
  const ReduceItem &q = R.it[blockIdx.y];
  if ((int)blockIdx.x * kRedX >= q.M * q.Jt) return;
  const int x = threadIdx.x % kRedX, y = threadIdx.x / kRedX;
  const int idx = blockIdx.x * kRedX + x;
  const bool ok = idx < q.M * q.Jt;
  const int m = ok ? idx / q.Jt : 0, j = ok ? idx % q.Jt : 0;
  const float *p = q.part + (size_t)m * q.W + j;
  const size_t stride = (size_t)q.rows * q.W;
  const int per = (q.num_wg + kRedY - 1) / kRedY;
  const int w0 = y * per, w1 = w0 + per < q.num_wg ? w0 + per : q.num_wg;
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
    double t = 0;
    for (int k = 0; k < kRedY; ++k) t += sh[k][x];
    const float v = (float)t;
    if (j == q.J && q.bias) q.bias[m] = v;  // row sums to their own vector
    else q.C[(size_t)m * q.ldc + j] = v;
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
  const size_t lds = Shape<MB, NB>::lds_bytes;
  static PerDeviceOnce attr_set;
  if (!attr_set.test()) {
    if (hipFuncSetAttribute((const void *)planes_gemm_kernel<MB, NB>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm)");
    attr_set.set();
  }
  hipLaunchKernelGGL((planes_gemm_kernel<MB, NB>), dim3(num_wg), dim3(kThreads),
                     lds, st, G);
  return check_launch("planes_gemm");
}

// ---- the long-plane (stream) kernel: instantiated block shapes ------------
// row blocks of 16: 1, 2 or 4; column blocks of 16: 1, 2, 4, 7, 8 or 12
int stream_mb(int M) { const int b = (M + 15) / 16; return b <= 1 ? 1 : b <= 2 ? 2 : 4; }
int stream_nb(int J) {
  const int b = (J + 15) / 16;
  return b <= 1 ? 1 : b <= 2 ? 2 : b <= 4 ? 4 : b <= 7 ? 7 : b <= 8 ? 8 : 12;
}
long long stream_partial_floats(int M, int J) {
  return (long long)stream_mb(M) * 16 * (stream_nb(J) * 16 + 1);
}

template <int MB, int NB>
int launch_stream(const GemmArgs &G, int num_wg, hipStream_t st) {
  const size_t lds = (size_t)4 * MB * 16 * (NB * 16 + 1) * sizeof(float);  // a region per wave
  static PerDeviceOnce attr_set;
  if (!attr_set.test() && lds > 48 * 1024) {
    if (hipFuncSetAttribute((const void *)planes_gemm_stream_kernel<MB, NB>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm_stream)");
    attr_set.set();
  }
  hipLaunchKernelGGL((planes_gemm_stream_kernel<MB, NB>), dim3(num_wg),
                     dim3(kThreads), lds, st, G);
  return check_launch("planes_gemm_stream");
}

template <int MB>
int launch_stream_nb(const GemmArgs &G, int NB, int num_wg, hipStream_t st) {
  switch (NB) {
    case 1: return launch_stream<MB, 1>(G, num_wg, st);
    case 2: return launch_stream<MB, 2>(G, num_wg, st);
    case 4: return launch_stream<MB, 4>(G, num_wg, st);
    case 7: return launch_stream<MB, 7>(G, num_wg, st);
    case 8: return launch_stream<MB, 8>(G, num_wg, st);
    default:
      if constexpr (MB <= 2) return launch_stream<MB, 12>(G, num_wg, st);
      set_error("planes_gemm: J > 128 needs M <= 32");
      return APG_ERR_ARG;
  }
}

int launch_stream_shape(const GemmArgs &G, int num_wg, hipStream_t st) {
  const int NB = stream_nb(G.J);
  switch (stream_mb(G.M)) {
    case 1: return launch_stream_nb<1>(G, NB, num_wg, st);
    case 2: return launch_stream_nb<2>(G, NB, num_wg, st);
    default: return launch_stream_nb<4>(G, NB, num_wg, st);
  }
}

int launch_shape(const GemmArgs &G, int MB, int NB, int num_wg, hipStream_t st) {
  if (MB == 1) {
    switch (NB) {
      case 1: return launch<1, 1>(G, num_wg, st);
      case 2: return launch<1, 2>(G, num_wg, st);
      case 3: return launch<1, 3>(G, num_wg, st);
      case 4: return launch<1, 4>(G, num_wg, st);
      case 5: return launch<1, 5>(G, num_wg, st);
      default: return launch<1, 6>(G, num_wg, st);
    }
  }
  switch (NB) {
    case 1: return launch<2, 1>(G, num_wg, st);
    case 2: return launch<2, 2>(G, num_wg, st);
    case 3: return launch<2, 3>(G, num_wg, st);
    default: return launch<2, 4>(G, num_wg, st);
  }
}

// LDS bytes of a tile shape's ring (Shape<MB, NB>::lds_bytes at run time)
int shape_lds_bytes(int MB, int NB) {
  const int buf = (MB + NB) * 8 * kGS * 4;
  return ((APG_GEMM_ST_MAX >= 3 && 3 * buf <= kLdsBytes) ? 3 : 2) * buf;
}

int cu_count() { return device_cu_count(); }

// workgroups per CU as measured best (tools/bench_gemm.py, WGS sweep): what the
// LDS holds, but at most 2 - except the 32 x 32 shape, whose tiles are so
// small that 4 rings per CU are needed to keep enough bytes in flight
int default_wgs(int MB, int NB) {
  int per_cu = kLdsBytes / shape_lds_bytes(MB, NB);
  const int cap = MB + NB <= 2 ? 4 : 2;
  per_cu = per_cu < 1 ? 1 : per_cu > cap ? cap : per_cu;
  return cu_count() * per_cu;
}

// one workgroup (4 waves, one per SIMD) per CU: a wave holds all accumulator
// tiles and two sets of operand registers
int stream_default_wgs(int M, int J) {
  // ... except the smallest shape (the LSTM's 4 x 8 output layer: 12 live rows,
  // 1.5 KB per wave and chunk): too few bytes in flight with one workgroup per
  // CU, and its 76 registers leave room for four (18.2 -> 11.4 us; eight: 14.4)
  return cu_count() * (stream_mb(M) + stream_nb(J) <= 2 ? 4 : 1);
}

const char *check_problem(const float *A, const float *Bp, const int *bdesc,
                          const float *C, int M, int S, int J, int Jt, int sdiv,
                          int b_planes, long long N, int ldc, bool own_bias) {
  if (!A || !Bp || !bdesc || !C) return "NULL pointer";
  const int MB = (M + 31) / 32, NB = (Jt + 31) / 32;
  if (M < 1 || M > 64 || S < 1 || J < 1 || Jt > kMaxNB * 32 || N < 1 || sdiv < 1 ||
      ldc < (own_bias ? J : Jt) || (MB == 2 && NB > 4))
    return "need 1 <= M <= 64, J + ones <= 192 (<= 128 when M > 32), S, N, sdiv >= 1, "
           "ldc >= J + ones";
  if (b_planes < 1 || (long long)M * S * N * 4 >= kMaxOperandBytes ||
      (long long)b_planes * N * 4 >= kMaxOperandBytes)
    return "operands must be smaller than 4 GiB each (32-bit buffer offsets): pass "
           "B from the first plane the product uses, or split the batch";
  if ((long long)S * ((N + kKT - 1) / kKT) >= (1ll << 31)) return "too many tiles";
  return nullptr;
}

void fill_args(GemmArgs &G, const float *A, const float *Bp, const int *bdesc,
               float *part, int M, int S, int J, int sdiv, int with_ones,
               int b_planes, long long N) {
  G.A = A, G.Bp = Bp, G.bdesc = bdesc, G.part = part;
  G.a_bytes = (long long)M * S * N * 4;
  G.b_bytes = (long long)b_planes * N * 4;
  G.N = N, G.M = M, G.S = S, G.J = J;
  G.sdiv = sdiv;
  G.with_ones = with_ones ? 1 : 0;
  G.tiles_per_seg = (int)((N + kKT - 1) / kKT);
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

// which kernel runs a product: the register-streaming one for plain (S = 1)
// products and for segmented ones with fewer than 16 B rows (the position
// product of the conv gradient: 3 rows - the LDS-tile kernel moves 32-row tiles
// for them and takes 21-24 us for 60 MB), the LDS-tile one for the others (the
// sliding windows: their segments share B rows through the L2)
static bool use_stream(int S, int J) { return APG_GEMM_STREAM && (S == 1 || J < 16); }

static long long partial_floats(int M, int S, int J, int with_ones) {
  if (use_stream(S, J)) return stream_partial_floats(M, J);
  const int MB = (M + 31) / 32, NB = (J + (with_ones ? 1 : 0) + 31) / 32;
  return (long long)MB * 32 * NB * 32;
}

int apg_planes_gemm_workspace_floats(int M, int J, int with_ones, int num_wg) {
  // enough for either kernel (the segment count is not known here)
  const long long a = partial_floats(M, 1, J, with_ones);
  const long long b = partial_floats(M, 2, J, with_ones);
  return (int)(num_wg * (a > b ? a : b));
}

int apg_planes_gemm_default_wgs(int M, int S, int J, int with_ones) {
  if (use_stream(S, J)) return stream_default_wgs(M, J);
  return default_wgs((M + 31) / 32, (J + (with_ones ? 1 : 0) + 31) / 32);
}

int apg_planes_gemm(const float *A, int M, int S, const float *Bp,
                    const int *bdesc, int J, int sdiv, int with_ones,
                    int b_planes, long long N, float *workspace, int num_wg,
                    float *C, int ldc, float *bias_out, apg_stream_t stream) {
  const int Jt = J + (with_ones ? 1 : 0);
  if (!workspace || num_wg < 1) {
    set_error("apg_planes_gemm: workspace is NULL or num_wg < 1");
    return APG_ERR_ARG;
  }
  if (const char *why = check_problem(A, Bp, bdesc, C, M, S, J, Jt, sdiv, b_planes, N,
                                      ldc, bias_out != nullptr)) {
    set_error("apg_planes_gemm: %s", why);
    return APG_ERR_ARG;
  }
  GemmArgs G;
  fill_args(G, A, Bp, bdesc, workspace, M, S, J, sdiv, with_ones, b_planes, N);
  hipStream_t st = (hipStream_t)stream;
  ReduceArgs R;
  if (use_stream(S, J)) {
    if (int e = launch_stream_shape(G, num_wg, st)) return e;
    R.it[0] = ReduceItem{workspace, C, with_ones ? bias_out : nullptr, num_wg,
                         stream_nb(J) * 16 + 1, stream_mb(M) * 16, M, Jt, J, ldc};
  } else {
    const int MB = (M + 31) / 32, NB = (Jt + 31) / 32;
    if (int e = launch_shape(G, MB, NB, num_wg, st)) return e;
    R.it[0] = ReduceItem{workspace, C, with_ones ? bias_out : nullptr, num_wg, NB * 32,
                         MB * 32, M, Jt, J, ldc};
  }
  hipLaunchKernelGGL(planes_gemm_reduce_kernel, dim3((M * Jt + kRedX - 1) / kRedX, 1),
                     dim3(1024), 0, st, R);
  return check_launch("planes_gemm_reduce");
}

// Workgroups of every product of an apg_planes_gemm_multi call.  Short planes
// (<= kGroupMaxN columns: one column per trajectory, the concurrent mode): the
// plain (S = 1) products share ONE launch of cu_count() workgroups, divided in
// proportion to the planes they stream (at least one each) - seven separate
// launches would be mostly ramp and tail (fixed-wing step 0.53 -> 0.46 ms).
// Long planes: every product its own launch (sharing one made the LSTM step
// slower, 0.82 -> 0.93 ms).  A segmented product always gets its own launch.
constexpr long long kGroupMaxN = 262144;
static bool multi_grouped(const ApgGemmProblem *problems, int n) {
  int plain = 0;
  for (int p = 0; p < n; ++p)
    if (use_stream(problems[p].S, problems[p].J)) {
      if (problems[p].N > kGroupMaxN) return false;
      ++plain;
    }
  return plain > 1;
}

// a grouped product's share of the workgroups: the rows it streams.  (Adding
// a term per accumulator tile for the matrix instructions it issues moved the
// launch by < 1 us once those were on the 16-bit pipe: profiles/r03_gemm_stream_bf16x3.txt)
static double group_cost(const ApgGemmProblem &q) {
  return (double)(q.M + q.J) * (double)q.S * (double)q.N;
}

static void multi_wgs(const ApgGemmProblem *problems, int n, int *wgs) {
  const bool grouped = multi_grouped(problems, n);
  double total = 0;
  int plain = 0;
  for (int p = 0; p < n; ++p)
    if (use_stream(problems[p].S, problems[p].J)) {
      total += group_cost(problems[p]);
      ++plain;
    }
  const int pool = cu_count() > plain ? cu_count() - plain : 0;
  for (int p = 0; p < n; ++p) {
    const ApgGemmProblem &q = problems[p];
    if (grouped && use_stream(q.S, q.J))
      wgs[p] = 1 + (int)(pool * (group_cost(q) / total));
    else
      wgs[p] = apg_planes_gemm_default_wgs(q.M, q.S, q.J, q.with_ones);
  }
}

long long apg_planes_gemm_multi_workspace_floats(const ApgGemmProblem *problems,
                                                 int n) {
  if (!problems || n < 1 || n > kMaxGroup) return 0;
  int wgs[kMaxGroup];
  multi_wgs(problems, n, wgs);
  long long total = 0;
  for (int p = 0; p < n; ++p)
    total += (long long)wgs[p] * partial_floats(problems[p].M, problems[p].S,
                                                problems[p].J, problems[p].with_ones);
  return total;
}

int apg_planes_gemm_multi(const ApgGemmProblem *problems, int n, float *workspace,
                          apg_stream_t stream) {
  if (!problems || !workspace || n < 1 || n > kMaxGroup) {
    set_error("apg_planes_gemm_multi: need 1 <= n <= %d problems and a workspace",
              kMaxGroup);
    return APG_ERR_ARG;
  }
  for (int p = 0; p < n; ++p) {
    const ApgGemmProblem &q = problems[p];
    const int Jt = q.J + (q.with_ones ? 1 : 0);
    if (const char *why = check_problem(q.A, q.B, q.bdesc, q.C, q.M, q.S, q.J, Jt,
                                        q.sdiv, q.b_planes, q.N, q.ldc,
                                        q.bias_out && q.with_ones)) {
      set_error("apg_planes_gemm_multi: problem %d: %s", p, why);
      return APG_ERR_ARG;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  int wgs[kMaxGroup];
  multi_wgs(problems, n, wgs);
  const bool grouped = multi_grouped(problems, n);
  ReduceArgs R;
  StreamGroupArgs SG;
  SG.n = 0;
  SG.wg0[0] = 0;
  size_t group_lds = 0;
  float *part = workspace;
  int max_blocks = 1;
  for (int p = 0; p < n; ++p) {
    const ApgGemmProblem &q = problems[p];
    const int Jt = q.J + (q.with_ones ? 1 : 0);
    GemmArgs G;
    fill_args(G, q.A, q.B, q.bdesc, part, q.M, q.S, q.J, q.sdiv, q.with_ones,
              q.b_planes, q.N);
    int rows, W;
    if (use_stream(q.S, q.J) && !grouped) {
      if (int e = launch_stream_shape(G, wgs[p], st)) return e;
      rows = stream_mb(q.M) * 16, W = stream_nb(q.J) * 16 + 1;
    } else if (use_stream(q.S, q.J)) {
      const int mb = stream_mb(q.M), nb = stream_nb(q.J);
      rows = mb * 16, W = nb * 16 + 1;
      SG.g[SG.n] = G;
      SG.shape[SG.n] = mb * 16 + nb;
      SG.wg0[SG.n + 1] = SG.wg0[SG.n] + wgs[p];
      ++SG.n;
      const size_t lds = (size_t)4 * rows * W * sizeof(float);   // a region per wave
      group_lds = lds > group_lds ? lds : group_lds;
    } else {
      const int MB = (q.M + 31) / 32, NB = (Jt + 31) / 32;
      if (int e = launch_shape(G, MB, NB, wgs[p], st)) return e;
      rows = MB * 32, W = NB * 32;
    }
    R.it[p] = ReduceItem{part, q.C, q.with_ones ? q.bias_out : nullptr, wgs[p], W,
                         rows, q.M, Jt, q.J, q.ldc};
    part += (size_t)wgs[p] * rows * W;
    const int blocks = (q.M * Jt + kRedX - 1) / kRedX;
    max_blocks = blocks > max_blocks ? blocks : max_blocks;
  }
  if (SG.n > 0) {
    static PerDeviceOnce attr_set;
    if (!attr_set.test()) {  // room for the largest partial any shape can have
      if (hipFuncSetAttribute((const void *)planes_gemm_stream_grouped_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes) != hipSuccess)
        return check_launch("hipFuncSetAttribute(planes_gemm_stream_grouped)");
      attr_set.set();
    }
    hipLaunchKernelGGL(planes_gemm_stream_grouped_kernel, dim3(SG.wg0[SG.n]),
                       dim3(kThreads), group_lds, st, SG);
    if (int e = check_launch("planes_gemm_stream_grouped")) return e;
  }
  hipLaunchKernelGGL(planes_gemm_reduce_kernel, dim3(max_blocks, n), dim3(1024), 0, st,
                     R);
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
    if (q.b_planes < 1 || a_bytes >= kMaxOperandBytes || b_bytes >= kMaxOperandBytes) {
      set_error("apg_planes_gemm_grouped: problem %d: operands must be < 4 GiB", p);
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
  const size_t lds = Shape<2, 4>::lds_bytes;
  static PerDeviceOnce attr_set;
  if (!attr_set.test()) {
    if (hipFuncSetAttribute((const void *)planes_gemm_grouped_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return check_launch("hipFuncSetAttribute(planes_gemm_grouped)");
    attr_set.set();
  }
  hipLaunchKernelGGL(planes_gemm_grouped_kernel, dim3(given), dim3(kThreads), lds, st,
                     GA);
  if (int e = check_launch("planes_gemm_grouped")) return e;
  hipLaunchKernelGGL(planes_gemm_grouped_reduce_kernel, dim3((64 * 128 + 63) / 64, n),
                     dim3(256), 0, st, GA);
  return check_launch("planes_gemm_grouped_reduce");
}

}  // extern "C"
