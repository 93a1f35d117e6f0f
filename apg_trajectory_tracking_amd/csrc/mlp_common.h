// mlp_common.h - what the MLP-policy kernels of the two quadrotor training modes
// share (round 6: mlp.hip split per mode - mlp_rollout.hip: autoregressive sweeps
// + closed-loop evaluation; mlp_concurrent.hip: the concurrent step;
// mlp_planes.hip: the plane-writing reverse kernels of rounds 1-4, test library
// only): network dimensions, the packed operand tables of the forward sweeps
// (cfwd), the transposed tables the reverse tables are derived from (cbwd), the
// in-sweep table layouts and slot maps, the second stage of the training steps
// (fixed-order reduction of the workgroups' partial blocks, momentum SGD, resident
// table map), argument checks.  Everything lives in an anonymous namespace: every
// translation unit that includes this header gets its own copy.
#pragma once
#include "apg_device.h"
#include "policy_mfma.h"
#include "policy_mfma16.h"
#include "policy_tm.h"
#include "quad_math.h"
#include "learnt_residual.h"

namespace apg {
namespace {

constexpr int kH = 10, kRD = 9, kNF = 15, kNC = 20, kNP = kH - 2;
constexpr int kW = 64;               // width of s1, h1, h2, h3
constexpr int kN1 = kW + kNC * kNP;  // fc1 input width (224)
// What the autoregressive reverse sweep leaves for the conv weight gradient
// (same as lstm.hip): the window of (step k, position pos, tap t) is reference
// row k + pos + t, so dW[ch][c][t] = sum_{sigma,n} G[ch][sigma][n] R[sigma+t][c][n]
// with G[ch][sigma] = sum_{k+pos=sigma} d[ch][pos][k] - 17 diagonal sums per
// channel instead of 80 (pos, k) planes.  A lane holds the positions
// pos = 4 hi + ii of a channel: each half-wave keeps its own diagonals
// tau = k + ii (13 of them, sigma = tau + 4 hi) in four sliding registers per
// channel and stores a diagonal when its last term is in.
//   planes [0, kConvP):        G[ch][hi][tau]  (kNC x 2 x 13, B floats each)
//   planes [kConvP, +kNC*kH):  P[ch][k] = sum_pos d[ch][pos][k] (relative-
//                              position shift of window columns 0..2, bias)
constexpr int kTau = kH + 3;
constexpr int kConvP = kNC * 2 * kTau;          // 520
constexpr int kConvPlanes = kConvP + kNC * kH;  // 720
constexpr int kThreads = 512;
constexpr int kTrajPerBlock = kThreads / 2;
// ------------------------------------------------------------ forward sweep
// acc[rb] (row block rb of a 64-wide layer) = bias table at `tab`
__device__ __forceinline__ void init_bias(f32x16 (&acc)[2], const LdsView &L, int tab) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[rb][i] = L.T(tab + (rb * 16 + i) * 2);
}

// The operand tables are gathered ONCE per launch into a global workspace (one
// small kernel); every workgroup then fills its LDS with a linear, fully
// coalesced copy instead of scattered loads of its own.
struct PackArgs {
  ApgMlpPolicy pol;
  float *dst;
  int head_rows;   // rows of fc_out behind pol.w_out (4: autoregressive, 40: concurrent)
};

// Forward tables of the kernels (fp16 split operands,
// policy_mfma16.h): the small fp32 tables indexed by the half-wave first -
// biases [rb][16][2], the head's VALU weights of the autoregressive sweep
// [4][2][16][2] + its bias - then 60 A-operand blocks of 2 KB: states_in [rb],
// conv [kb], fc1 conv part [rb][position pair][kb], fc1 state part / fc2 /
// fc3 / the concurrent mode's 40-row head [rb][kb].
constexpr int hTbs = 0, hTb1 = 64, hTb2 = 128, hTb3 = 192, hTbo = 256, hTbc = 320;  // floats
constexpr int hTo = 384, hBo = 640;               // floats: [4][2][16][2], [4]
constexpr int hA = 4096;                          // bytes: first A block
constexpr int nS = 0, nC = 2, n1c = 4, n1s = 28, n2 = 36, n3 = 44, nO = 52, nBlocks16 = 60;
constexpr int kCfLds = (hA + nBlocks16 * kBlock16) / 4;  // 31 744 floats = 126 976 B
static_assert(hBo + 4 <= hA / 4, "LDS map");
constexpr int kNA = kH * 4;                       // head width of the concurrent mode (40)

// weight behind k-slot (kb, j, hi) of A block n, output row `row` (0..31 of
// the block's row block) - the single definition of the forward k-orders
__device__ __forceinline__ float cfwd_weight(const ApgMlpPolicy &p, int n, int row, int j,
                                             int hi, int head_rows) {
  if (n < nC) {                       // states_in: features 8 hi + j
    const int k = 8 * hi + j;
    return k < kNF ? p.w_s[((n - nS) * 32 + row) * kNF + k] : 0.f;
  }
  if (n < n1c) {                      // conv: slot s = (column j', tap), 15 of 16
    const int s = (n - nC) * 8 + j, jc = s / 3, tap = s % 3, q = hi ? 4 + jc : jc;
    return (s < 15 && row < kNC && (hi || jc < 4)) ? p.conv_w[row * 27 + q * 3 + tap] : 0.f;
  }
  if (n < n1s) {                      // fc1 on the conv outputs of a position pair
    const int m = n - n1c, rb = m / 12, pp = (m / 3) % 4, kb = m % 3;
    const int s = kb * 8 + j, pos = 2 * pp + s / 12, ch = rrow(s % 12) + 4 * hi;
    return ch < kNC ? p.w_1[(rb * 32 + row) * kN1 + kW + ch * kNP + pos] : 0.f;
  }
  const int m = (n - n1s) % 8, rb = m / 4, kb = m % 4, k = kin(kb, j, hi);
  const int out = rb * 32 + row;
  if (n < n2) return p.w_1[out * kN1 + k];
  if (n < n3) return p.w_2[out * kW + k];
  if (n < nO) return p.w_3[out * kW + k];
  return out < head_rows ? p.w_out[out * kW + k] : 0.f;
}

// (tid of T threads: the kernels below share the two bodies)
__device__ __forceinline__ void pack_cfwd(const PackArgs &A, int tid, int T) {
  const ApgMlpPolicy &p = A.pol;
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  // one thread per (block, lane, word): two weights -> fp16 high / low terms
  for (int idx = tid; idx < nBlocks16 * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    const float w0 = cfwd_weight(p, n, l & 31, 2 * q, l >> 5, A.head_rows);
    const float w1 = cfwd_weight(p, n, l & 31, 2 * q + 1, l >> 5, A.head_rows);
    unsigned h, lo;
    split_pair(w0, w1, h, lo);
    dst[(hA + n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(hA + n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
  for (int idx = tid; idx < 64; idx += T) {
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = idx >> 5;
    const int row = rb * 32 + rrow(i) + 4 * hi;
    A.dst[hTbs + idx] = p.b_s[row];
    A.dst[hTb1 + idx] = p.b_1[row];
    A.dst[hTb2 + idx] = p.b_2[row];
    A.dst[hTb3 + idx] = p.b_3[row];
    A.dst[hTbo + idx] = row < A.head_rows ? p.b_out[row] : 0.f;
  }
  for (int idx = tid; idx < 32; idx += T) {
    const int hi = idx & 1, i = idx >> 1, ch = rrow(i) + 4 * hi;
    A.dst[hTbc + idx] = ch < kNC ? p.conv_b[ch] : 0.f;
  }
  // the first four head rows as VALU weights (autoregressive sweep); a
  // concurrent-mode head (40 rows) has them too
  for (int idx = tid; idx < 256; idx += T) {
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = (idx >> 5) & 1, j = idx >> 6;
    A.dst[hTo + idx] = p.w_out[j * kW + rb * 32 + rrow(i) + 4 * hi];
  }
  for (int idx = tid; idx < 4; idx += T) A.dst[hBo + idx] = p.b_out[idx];
}

__global__ __launch_bounds__(256) void mlp_pack_cfwd_kernel(PackArgs A) {
  pack_cfwd(A, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
struct FwdArgs {
  const float *state0, *in_ref;
  float *states, *actions;
  float *feat, *x1, *h;  // [15][N], [224][N], [192][N] (h1, h2, h3)
  unsigned *mask;        // [5][N]
  const float *tables;  // packed operand tables (mlp_pack_cfwd_kernel)
  // [waves][H][4] or NULL: per step max relu(conv), |feature|, a bound of the
  // |window value|s of each wave's 32 trajectories (the in-sweep reverse kernel's
  // fixed-point scales, see mlp_rollout_bwd_tm_kernel)
  float *xmax;
  QuadConst c;
  int B;
};


// XMAX: also leave the step maxima at A.xmax (compile time: the step loop has no branch)

// ------------------------------------------------------------ reverse sweep
// Reverse tables of the training kernels (fp16 split operands,
// policy_mfma16.h): the autoregressive sweep's small fp32 tables first (head
// weights for the VALU [4][2][16][2], tap sums of the conv weights [20][3]),
// then 54 transposed A-operand blocks of 2 KB: the concurrent mode's head^T
// [rb][kb of 3], fc3^T, fc2^T, fc1^T state part [rb][kb], fc1^T conv part
// [32-row block eb of 5][kb], states_in^T [kb].
constexpr int gTo = 0, gAq = 256;                 // floats
constexpr int gA = 2048;                          // bytes: first A block
constexpr int mOT = 0, m3T = 6, m2T = 14, m1sT = 22, m1cT = 30, mST = 50, mBlocks16 = 54;
constexpr int kCbLds = (gA + mBlocks16 * kBlock16) / 4;  // 28 160 floats = 112 640 B
static_assert(gAq + kNC * 3 <= gA / 4, "LDS map");
// k index of head-output k-pair c of the concurrent mode (accumulator layout of
// the 40 outputs: row block 0 registers 0..15, row block 1 registers 0..3)
__host__ __device__ constexpr int khead(int c, int hi) {
  return (c < 16 ? rrow(c) : 32 + rrow(c - 16)) + 4 * hi;
}

// weight behind k-slot (kb, j, hi) of transposed A block n, output row `row`
__device__ __forceinline__ float cbwd_weight(const ApgMlpPolicy &p, int n, int row, int j,
                                             int hi, int head_rows) {
  if (n < m3T) {                      // head^T: slots = this lane's 20 dL/dz rows
    const int rb = n / 3, cc = (n % 3) * 8 + j;
    return (cc < 20 && khead(cc, hi) < head_rows) ? p.w_out[khead(cc, hi) * kW + rb * 32 + row]
                                                  : 0.f;
  }
  if (n < m1cT) {
    const int m = (n - m3T) % 8, rb = m / 4, k = kin(m % 4, j, hi), out = rb * 32 + row;
    if (n < m2T) return p.w_3[k * kW + out];
    if (n < m1sT) return p.w_2[k * kW + out];
    return p.w_1[k * kN1 + out];
  }
  if (n < mST) {
    const int m = n - m1cT, eb = m / 4, k = kin(m % 4, j, hi);
    return p.w_1[k * kN1 + kW + eb * 32 + row];
  }
  return row < kNF ? p.w_s[kin(n - mST, j, hi) * kNF + row] : 0.f;  // states_in^T
}

__device__ __forceinline__ void pack_cbwd(const PackArgs &A, int tid, int T) {
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  for (int idx = tid; idx < mBlocks16 * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    const float w0 = cbwd_weight(A.pol, n, l & 31, 2 * q, l >> 5, A.head_rows);
    const float w1 = cbwd_weight(A.pol, n, l & 31, 2 * q + 1, l >> 5, A.head_rows);
    unsigned h, lo;
    split_pair(w0, w1, h, lo);
    dst[(gA + n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(gA + n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
  const ApgMlpPolicy &p = A.pol;
  for (int idx = tid; idx < 256; idx += T) {   // first four head rows, VALU order
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = (idx >> 5) & 1, j = idx >> 6;
    A.dst[gTo + idx] = p.w_out[j * kW + rb * 32 + rrow(i) + 4 * hi];
  }
  for (int idx = tid; idx < kNC * 3; idx += T) {
    const int ch = idx / 3, q = idx % 3;
    A.dst[gAq + idx] = p.conv_w[ch * 27 + q * 3] + p.conv_w[ch * 27 + q * 3 + 1] +
                       p.conv_w[ch * 27 + q * 3 + 2];
  }
}

__device__ __forceinline__ void zero(f32x16 (&v)[2]) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) v[rb][i] = 0.f;
}

// The saved activations of a layer (planes [base, base + 64), accumulator
// layout) are requested one matrix product ahead of their use ...
__device__ __forceinline__ void load_acts(float (&hv)[2][16], const Planes &act,
                                          int base, unsigned vr, unsigned pN) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[rb][i] = act.ld(vr, (base + rb * 32 + rrow(i)) * pN);
}

// ... and applied here: v *= 1 - act^2 (tanh'), result written to the
// cotangent planes [out_base, out_base + 64)
__device__ __forceinline__ void tanh_adjoint(f32x16 (&v)[2], const float (&hv)[2][16],
                                             const Planes &out, int out_base,
                                             unsigned vr, unsigned pN) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      v[rb][i] *= 1.f - hv[rb][i] * hv[rb][i];
      out.st(vr, (out_base + rb * 32 + rrow(i)) * pN, v[rb][i]);
    }
}

// the same for a product that arrives scaled by 2^-ex (policy_mfma16.h)
__device__ __forceinline__ void tanh_adjoint(f32x16 (&v)[2], const float (&hv)[2][16],
                                             const Planes &out, int out_base,
                                             unsigned vr, unsigned pN, int ex) {
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      v[rb][i] = __builtin_amdgcn_ldexpf(v[rb][i], ex);
  tanh_adjoint(v, hv, out, out_base, vr, pN);
}

// ------------------------------------------ concurrent mode, rounds 4-5: the
// reverse pass of the network WITH its weight gradients, the step's second
// stage (mlp_concurrent_bwd_tm_kernel below; the round-4 "staged" kernel -
// cotangents transposed through LDS, owner waves - lives in
// tools/patches/mlp_concurrent_bwd_wg.patch).
// Reverse tables of the in-sweep kernels: [eb][kb] fc1^T conv part first, fc1^T
// state part, fc2^T, fc3^T, the concurrent head^T (cbwd_weight's blocks in the
// order the layers are passed, so that the tables of finished layers can be
// re-used as accumulator space).
constexpr int wC = 0, wS = 20, w2 = 28, w3 = 36, wO = 44, wBlocks = 50;
constexpr int kWgTabBytes = wBlocks * kBlock16;   // 102 400
constexpr int kWgTabFloats = kWgTabBytes / 4;
constexpr int kLdsAll = 160 * 1024;
// partial slots of a workgroup (1024 floats each, accumulator order [reg][lane])
constexpr int sOut = 0, sFc3 = 4, sFc2 = 8, sFc1 = 12, sSin = 26, sConv = 28;
// second stage: workgroups per first-level chunk, element columns of 256
constexpr int kRedChunk = 32;

// first planes of the x blocks in the activation buffer (feat | x1 | h1 h2 h3 | in_ref)
constexpr int pFeat = 0, pX1 = 15, pH1 = 239, pH2 = 303, pH3 = 367, pInr = 431,
              kActPlanes = 431 + kH * kRD;

__device__ __forceinline__ float cwg_weight(const ApgMlpPolicy &p, int n, int row, int j,
                                            int hi, int head_rows) {
  const int old = n < wS ? m1cT + n : n < w2 ? m1sT + (n - wS) : n < w3 ? m2T + (n - w2)
                  : n < wO ? m3T + (n - w3) : mOT + (n - wO);
  return cbwd_weight(p, old, row, j, hi, head_rows);
}

__device__ __forceinline__ void pack_cwg(const PackArgs &A, int tid, int T) {
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  for (int idx = tid; idx < wBlocks * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    const float w0 = cwg_weight(A.pol, n, l & 31, 2 * q, l >> 5, A.head_rows);
    const float w1 = cwg_weight(A.pol, n, l & 31, 2 * q + 1, l >> 5, A.head_rows);
    unsigned h, lo;
    split_pair(w0, w1, h, lo);
    dst[(n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
}

// forward tables at dst, the in-sweep reverse tables at dst + kCfLds.  (Round 4
// had a last block here that left the exponents of W_1's largest column 1-norms
// behind the tables; the reverse kernel takes them from the tables itself now.)
__global__ __launch_bounds__(256) void mlp_pack_step_kernel(PackArgs A, int fwd_blocks) {
  if ((int)blockIdx.x < fwd_blocks) {
    pack_cfwd(A, blockIdx.x * blockDim.x + threadIdx.x, fwd_blocks * blockDim.x);
  } else {
    A.dst += kCfLds;
    pack_cwg(A, (blockIdx.x - fwd_blocks) * blockDim.x + threadIdx.x,
             (gridDim.x - fwd_blocks) * blockDim.x);
  }
}

struct WgArgs {
  const float *acts;     // [521][B]: feat 0..14 | x1 15..238 | h1 h2 h3 239..430 | in_ref 431..520
  const unsigned *mask;  // [5][B]
  const float *d_zout;   // [40][B] (the forward kernel's dL/d(head pre-activations))
  float *part;           // [workgroups][kSlotsTm][1024]
  const float *tables;
  const float *xmax;     // [waves][4] (the forward kernel's; trajectory-major kernel only)
  int B;
  // ROWS: the feature / window blocks of x^T are read from the DATA SET's rows
  // through the batch's index (the forward kernel then writes no planes of them)
  const long long *index;
  const float *r_feat, *r_in_ref;
  int ld_feat, ld_in_ref;
  unsigned bytes_feat, bytes_in_ref;
};

// ---------------------------------------------------------------------------
// The same reverse pass with TRAJECTORY-MAJOR weight products (round 4, last
// third): no cotangent staging, no owner waves, one barrier per layer.
//
// The matrix instruction computes D[m][n] = sum_k A[m][k] B[k][n] with lane l
// supplying row m = l & 31 of A and column n = l & 31 of B, eight k-slots each.
// The two operand register layouts are the same, so issuing the instruction
// with its operands SWAPPED yields the transposed product: where the chain
// computes e = W^T delta (feature-major: row = feature in the registers,
// column = trajectory in the lane) from the table block (A) and the split
// cotangent (B), the same two register sets the other way round give
// e^T[trajectory][feature] - 16 trajectories per lane, the feature in the
// lane.  That is exactly the A operand of the weight product
//   dW[m][k] = sum_n delta[m][n] x[k][n]
// (row = feature, k-slots = trajectories), and the matching B operand - x with
// the feature in the lane and the same 16 trajectories in the registers - is
// four 16-byte loads per lane from the forward kernel's planes.  So every wave
// multiplies ITS 32 trajectories' cotangents against its own x (K = 32 per
// block product: two instructions x three split terms) and adds the 32 x 32
// blocks into the workgroup's fp32 accumulators in LDS (ds_add_f32); a layer's
// accumulators are flushed to the partial buffer behind ONE barrier while the
// next layer adds into another region.  Costs: every layer product twice (the
// matrix pipe was ~15 % busy).  The accumulators are FIXED POINT (below):
// integer sums do not depend on the order in which the eight waves add, so the
// kernel is bit-reproducible like the staged one
// LDS: tables [0, 100 K) as above; accumulator regions in the 60 KB behind them
// and, for fc1's 14 blocks, also in the tables of the layers already passed:
//   R_A [100 K, 116 K)  head, then fc2          R_B [116 K, 132 K)  fc3
//   fc1: blocks 0..6 in [72 K, 100 K) (w3 / head tables, dead and zeroed after
//        fc3's barrier), blocks 7..13 in [116 K, 144 K)
//   states_in [144 K, 152 K), conv [152 K, 156 K), biases [156 K, 157 K)
constexpr int tRA = 100 * 1024, tRB = 116 * 1024, tF1a = 72 * 1024, tF1b = 116 * 1024,
              tSin = 144 * 1024, tConv = 152 * 1024, tHeadEx = 156 * 1024,
              tHeadRow = tHeadEx + 512, tMeta = 157 * 1024,
              tConvLo = tMeta + 256;   // [20][32] low limbs of the conv block (2.5 KB)
// tHeadEx: [8 waves][40] biased exponents (bytes) of the head rows' largest
// cotangents, tHeadRow: [40] the rows' exponents (ints) - round 5
static_assert(tConvLo + kNC * 32 * 4 <= kLdsAll, "LDS map");
// partial slots of this kernel: as above up to sConv, which holds ALL positions
// ... and two bias slots: [4 waves][4 layers][64] float sums per wave each
constexpr int uConv = sConv, uBias = sConv + 1, kSlotsTm = sConv + 3;
// The accumulators are 32-bit FIXED POINT: ds_add_f32 costs ~0.4 us per wave
// instruction on this part (the first build: 350 us per launch), ds_add_u32 runs
// at LDS speed - and integer sums do not depend on the order of the eight
// waves, so the kernel is bit-reproducible.  Both operands of a block product
// are scaled into [-1, 1] by powers of two (exact): the cotangent by the
// WORKGROUP's exponent of the layer (the waves' maxima are exchanged through
// LDS one layer ahead, behind the barrier that is there anyway), x by 1 (tanh
// planes) or by the workgroup's exponent of its plane group (conv outputs,
// features, in_ref: measured from the planes before the first barrier).  A
// wave's block element is then |sum of 32 products| <= 32, eight waves <= 2^8:
// unit 2^-22, sums below 2^30.  Quantisation 2^-23 of the layer's largest
// cotangent x largest x per addition - the absolute accuracy the staged
// kernel's per-workgroup fp16 split has (2^-25), three bits coarser.  The
// cotangents of states_in and conv are produced inside the fc1 phase, so their
// exponents are BOUNDS: fc1's exponent + that of the largest column 1-norm of
// W_1's state / conv part (two floats behind the tables, mlp_pack_step_kernel).
// The conv block collects 8 positions as well - 2^11 terms, unit 2^-19 - and its
// cotangent's exponent is a loose bound (above), so it keeps a second limb: the
// rounding remainder of every addition in units of 2^-38 (compact [channel][32]).
// (fixed-point accumulators, block loads and splits, exponent exchange: policy_tm.h)
static_assert(kThreads == kTmThreads, "policy_tm.h");
struct TmMeta {           // at tMeta; written by plain stores, one slot per wave
  unsigned dmax[4][8];    // max |cotangent| bits of head, fc3, fc2, fc1
  float wnorm[8];         // largest column 1-norm of W_1 in row block `wave` of W_1^T
};

// Second stage: the workgroups' partial blocks summed in a fixed order
// (deterministic), scattered into the parameter gradients; block 0 also sums
// the loss partials of the forward kernel.
// destination of element (slot, reg i, lane) - or NULL (padding)
__device__ __forceinline__ float *wg_dest(const ApgMlpPolicyGrads &g, int slot, int i, int lane,
                                          int bias_slot, int head_rows = kNA,
                                          bool conv_bias_here = false) {
  const int rowb = rrow(i) + 4 * (lane >> 5), col = lane & 31;
  if (slot < sFc1) {                       // head, fc3, fc2: [cb][mb]
    const int q = slot & 3, cb = q >> 1, m = 32 * (q & 1) + rowb, k = 32 * cb + col;
    if (slot < sFc3) return m < head_rows ? g.w_out + m * kW + k : nullptr;
    return (slot < sFc2 ? g.w_3 : g.w_2) + m * kW + k;
  }
  if (slot < sSin) {                       // fc1: 7 column blocks x 2 row blocks
    const int it = slot - sFc1, cb = it >> 1, m = 32 * (it & 1) + rowb;
    return g.w_1 + m * kN1 + 32 * cb + col;
  }
  if (slot < sConv) {                      // states_in: 15 columns + the bias column
    const int m = 32 * (slot - sSin) + rowb;
    return col < kNF ? g.w_s + m * kNF + col : col == kNF ? g.b_s + m : nullptr;
  }
  if (slot < bias_slot) {                  // conv, position slot - sConv: only slot
    if (slot != sConv || rowb >= kNC) return nullptr;   // sConv collects all of them
    if (col < 27) return g.conv_w + rowb * 27 + (col % kRD) * 3 + col / kRD;
    return col == 27 && !conv_bias_here ? g.conv_b + rowb : nullptr;
  }
  // bias slot(s): [layer][64]; the trajectory-major kernels' per-wave sums are
  // further sources of the same elements (bias_src in the reduce kernel)
  const int e = i * 64 + lane;
  if (slot != bias_slot || e >= 4 * 64) return nullptr;
  const int layer = e >> 6, m = e & 63;
  if (layer == 0 && conv_bias_here && m >= 32 && m < 32 + kNC) return g.conv_b + (m - 32);
  return layer == 0 ? (m < head_rows ? g.b_out + m : nullptr)
         : layer == 1 ? g.b_3 + m : layer == 2 ? g.b_2 + m : g.b_1 + m;
}

// Second stage, two launches.  Level 1: blockIdx.y = a chunk of kRedChunk
// workgroups, summed per element in workgroup order into chunk_sums[chunk][slots
// * 1024] - thousands of blocks, the 38 MB of partials stream at the HBM rate
// (one block per element column over all 256 workgroups, the first version,
// took 248 us).  Level 2 sums the chunks in order, scatters into the parameter
// gradients and sums the forward kernel's loss partials.  (Both levels in ONE
// launch - the last block of a column, found by a ticket between device-scope
// fences, doing level 2 - was built and measured: 135 us.  A device-scope
// release on this part writes the XCD's L2 back; a thousand blocks doing it
// cost more than the launch boundary they save.  One launch of 148 blocks of
// 1 024 threads, four sub-groups per column each summing a quarter of the
// workgroups: 40 us - too few blocks to stream 38 MB.)
struct WgReduceArgs {
  const float *part;   // level 2's source: chunk sums, or the partials themselves
  ApgMlpPolicyGrads g;
  // optimizer step inside the second stage (apg_quad_mlp_concurrent_train_step):
  // the thread that has summed a gradient element also owns the parameter and
  // its momentum entry
  ApgMlpPolicyGrads param, mom;
  double lr, momentum;
  bool update;
  // slot layout of the reverse kernel that wrote `part`: slots per workgroup,
  // where the bias slot is, how many conv position blocks follow sConv
  int n_slots, bias_slot, conv_src;
  int bias_src;        // per-wave bias sums behind the bias slot's first 256 floats
                       // (every 256 floats, across slot boundaries): 8, or 1
  int head_rows;       // rows of fc_out (40: concurrent mode, 4: autoregressive)
  bool conv_bias_here; // the conv bias sits in the bias slot (layer 0, entries 32..51)
  const float *loss_partials;
  float *loss;
  float *loss_sum;       // or NULL: += the loss (an epoch loop's running sum)
  int wgs, n_partials;   // wgs: how many [n_slots * 1024] rows `part` has
  // resident operand tables (see kMapFlag32): the thread that has updated a
  // parameter also rewrites its entries of the packed tables in `ws`
  char *ws;
  const int *map;        // [n_slots * 1024][4] byte offsets into ws, -1: none
};

// Resident operand tables (round 5).  The step's kernels read the policy from
// PACKED tables (fp16 pairs in matrix-operand order + a few float tables,
// mlp_pack_step_kernel: a launch of its own at the head of every step, 5-6 us).
// When the update happens inside the second stage the new value of a parameter
// is in the register of exactly one thread - which then writes the parameter's
// table entries for the NEXT step itself, and the pack launch goes away.  Where
// a parameter sits in the tables is not re-derived by hand: once per workspace
// the pack kernel runs on parameter arrays that hold their own indices, and the
// result is inverted into map[reduce thread][4] (tabmap_* below; the fp16 pair
// of an index < 32 768 adds up to it exactly).
constexpr int kMapFlag32 = 1 << 30;   // the entry is one float (bias / VALU-head tables)
constexpr int kParamFloats = kW * kNF + kW + kNC * 27 + kNC + kW * kN1 + kW +
                             2 * (kW * kW + kW) + kNA * kW + kNA;   // 26 904
constexpr int kMapInts = (sConv + 3) * 1024 * 4;

__host__ __device__ inline ApgMlpPolicyGrads params_in(float *base) {
  ApgMlpPolicyGrads g;
  float *q = base;
  g.w_s = q, q += kW * kNF;
  g.b_s = q, q += kW;
  g.conv_w = q, q += kNC * 27;
  g.conv_b = q, q += kNC;
  g.w_1 = q, q += kW * kN1;
  g.b_1 = q, q += kW;
  g.w_2 = q, q += kW * kW;
  g.b_2 = q, q += kW;
  g.w_3 = q, q += kW * kW;
  g.b_3 = q, q += kW;
  g.w_out = q, q += kNA * kW;
  g.b_out = q;
  return g;
}

__global__ __launch_bounds__(256) void tabmap_iota_kernel(float *par) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < kParamFloats) par[i] = (float)(i + 1);
}

// owner[id] = the second-stage thread that sums (and updates) parameter `id`
__global__ __launch_bounds__(256) void tabmap_owner_kernel(float *par, int *owner, int *map,
                                                           int n_slots, int bias_slot) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_slots * 1024) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) map[t * 4 + k] = -1;
  const float *dst = wg_dest(params_in(par), t >> 10, (t >> 6) & 15, t & 63, bias_slot, kNA,
                             false);
  if (dst) owner[dst - par] = t;
}

// tab: the tables packed from index-valued parameters; every entry is appended
// to the list of its parameter's owner thread
__global__ __launch_bounds__(256) void tabmap_invert_kernel(const float *tab, const int *owner,
                                                            int *map) {
  const int p = blockIdx.x * 256 + threadIdx.x;   // float index into the workspace
  if (p >= kCfLds + kWgTabFloats) return;
  const auto record = [&](float x, int off) {
    const int id = (int)x - 1;
    if (id < 0 || id >= kParamFloats) return;
    int *m = map + owner[id] * 4;
    for (int k = 0; k < 4; ++k)
      if (atomicCAS(m + k, -1, off) == -1) return;
  };
  if (p < hA / 4) {                   // the forward kernels' float tables
    record(tab[p], p * 4 | kMapFlag32);
    return;
  }
  const int region = p < kCfLds ? hA : kCfLds * 4;   // first block of this table (bytes)
  if ((p * 4 - region) % kBlock16 >= 1024) return;   // a word of low terms
  const h16x2 h = __builtin_bit_cast(h16x2, tab[p]), l = __builtin_bit_cast(h16x2, tab[p + 256]);
  record((float)h[0] + (float)l[0], p * 4);
  record((float)h[1] + (float)l[1], p * 4 + 2);
}



__global__ __launch_bounds__(256) void mlp_wgrad_reduce1_kernel(const float *part,
                                                                float *chunk_sums, int wgs,
                                                                int n_slots) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_slots * 1024) return;
  const size_t stride = (size_t)n_slots * 1024;
  const int w0 = blockIdx.y * kRedChunk;
  const float *p = part + (size_t)w0 * stride + t;
  float v[kRedChunk];
#pragma unroll
  for (int k = 0; k < kRedChunk; ++k) v[k] = w0 + k < wgs ? p[(size_t)k * stride] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kRedChunk; ++k) s += v[k];   // fixed order
  chunk_sums[(size_t)blockIdx.y * stride + t] = s;
}

__global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(WgReduceArgs A) {
  const int t = blockIdx.x * 256 + threadIdx.x;   // (slot, reg, lane)
  const size_t stride = (size_t)A.n_slots * 1024;
  if (t < A.n_slots * 1024) {
    const int slot = t >> 10, i = (t >> 6) & 15, lane = t & 63;
    float *dst = wg_dest(A.g, slot, i, lane, A.bias_slot, A.head_rows, A.conv_bias_here);
    if (dst) {
      // what the update will need, requested BEFORE the partial sums (the loads
      // return in order: the parameter, its momentum entry and the table map
      // arrive under the sums' latency instead of behind it)
      float *pp = nullptr, *pm = nullptr;
      float p_old = 0.f, m_old = 0.f;
      int ent[4] = {-1, -1, -1, -1};
      if (A.update) {
        pp = wg_dest(A.param, slot, i, lane, A.bias_slot, A.head_rows, A.conv_bias_here);
        pm = wg_dest(A.mom, slot, i, lane, A.bias_slot, A.head_rows, A.conv_bias_here);
        p_old = *pp, m_old = *pm;
        if (A.map) {
#pragma unroll
          for (int k = 0; k < 4; ++k) ent[k] = A.map[t * 4 + k];
        }
      }
      const bool is_bias = slot == A.bias_slot;
      // the conv position blocks (1024 floats apart) / the waves' bias sums (256)
      const int n_src = slot == sConv ? A.conv_src : is_bias ? A.bias_src : 1;
      const int src_step = is_bias ? 256 : 1024;
      float s = 0.f;
      for (int q = 0; q < n_src; ++q) {
        const float *p = A.part + (size_t)slot * 1024 + (size_t)q * src_step + (t & 1023);
        // kRedChunk rows at a time (all loads in flight, then a fixed-order
        // sum); more than kRedChunk chunk rows - batches beyond 262 144
        // trajectories - take further rounds
        if (A.wgs <= 8) {   // (the eight chunk rows of a 65 536 batch: no idle slots)
          float v[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) v[w] = w < A.wgs ? p[(size_t)w * stride] : 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) s += v[w];
        } else {
          for (int w0 = 0; w0 < A.wgs; w0 += kRedChunk) {
            float v[kRedChunk];
#pragma unroll
            for (int w = 0; w < kRedChunk; ++w)
              v[w] = w0 + w < A.wgs ? p[(size_t)(w0 + w) * stride] : 0.f;
#pragma unroll
            for (int w = 0; w < kRedChunk; ++w) s += v[w];
          }
        }
      }
      *dst = s;
      if (A.update) {   // torch.optim.SGD: buf = momentum buf + grad, p -= lr buf
        // (in double with one rounding each, as torch's fused SGD kernel does
        // it: a trainer that steps through optimizer.step() - the multi-rank
        // form - gets the same bits)
        const float buf = (float)(A.momentum * (double)m_old + (double)s);
        *pm = buf;
        const float np_ = (float)((double)p_old - A.lr * (double)buf);
        *pp = np_;
        if (A.map) {      // this parameter's entries of the packed tables
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = ent[k];
            if (e < 0) break;
            char *q = A.ws + (e & (kMapFlag32 - 1));
            if (e & kMapFlag32) {
              *reinterpret_cast<float *>(q) = np_;
            } else {      // split_pair's two terms (policy_mfma16.h)
              const _Float16 h_ = (_Float16)np_;
              *reinterpret_cast<_Float16 *>(q) = h_;
              *reinterpret_cast<_Float16 *>(q + 1024) = (_Float16)(np_ - (float)h_);
            }
          }
        }
      }
    }
  }
  if (blockIdx.x == 0 && A.loss) {   // fixed-shape sum of the loss partials
    __shared__ double sm[4];
    double acc = 0.0;
    for (int k = threadIdx.x; k < A.n_partials; k += 256) acc += (double)A.loss_partials[k];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float l = (float)((sm[0] + sm[1]) + (sm[2] + sm[3]));
      *A.loss = l;
      if (A.loss_sum) *A.loss_sum += l;   // (one thread, stream order: deterministic)
    }
  }
}

int check_mlp(const ApgQuadParams *params, const ApgMlpPolicy *pol, int B, int H) {
  if (!params || !pol) { set_error("params / policy is NULL"); return APG_ERR_ARG; }
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if ((long long)B * kH * 4 * 256 >= (1ll << 32) - 64) {
    set_error("B too large for 32-bit plane offsets (max %d); split the batch",
              (int)(((1ll << 32) - 64) / (kH * 4 * 256)));
    return APG_ERR_ARG;
  }
  if (H != kH) {
    set_error("the fused autoregressive rollout is built for horizon %d (got %d)",
              kH, H);
    return APG_ERR_ARG;
  }
  if (!pol->w_s || !pol->b_s || !pol->conv_w || !pol->conv_b || !pol->w_1 ||
      !pol->b_1 || !pol->w_2 || !pol->b_2 || !pol->w_3 || !pol->b_3 ||
      !pol->w_out || !pol->b_out) {
    set_error("policy weight pointer is NULL");
    return APG_ERR_ARG;
  }
  return APG_OK;
}

template <typename K>
int raise_lds(K kernel, int floats) {
  if (hipFuncSetAttribute((const void *)kernel,
                          hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(floats * sizeof(float))) != hipSuccess)
    return check_launch("hipFuncSetAttribute(mlp_rollout)");
  return APG_OK;
}

inline bool all_set(const ApgMlpPolicyGrads &g) {
  return g.w_s && g.b_s && g.conv_w && g.conv_b && g.w_1 && g.b_1 && g.w_2 && g.b_2 &&
         g.w_3 && g.b_3 && g.w_out && g.b_out;
}

}  // namespace
}  // namespace apg
