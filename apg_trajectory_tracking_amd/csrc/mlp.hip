// mlp.hip - the AUTOREGRESSIVE quadrotor unroll with the MLP policy inside the
// kernel, on the matrix cores (BASELINE config 3 per GPU: quadrotor,
// autoregressive, H = 10).
//
// Replaces, for train_mode == "autoregressive", the loop of
//   TrainDrone.train_recurrent_model       scripts/train_drone.py:113-173
//   hutter_model.Net.forward (conv branch) neural_control/models/hutter_model.py:35-49
//   state_preprocessing                    neural_control/dataset.py:207-220
//   FlightmareDynamics / quad_mpc_loss     (see quad.hip)
// by a forward sweep and a reverse sweep (pinned window semantics of
// SURVEY.md §8a A4).  Network Net(15, 10, 9, 4, conv=1):
//   s1 = tanh(W_s feat + b_s)                      15 -> 64
//   cv = relu(conv1d(window^T; 9 -> 20, k = 3))    90 -> 160
//   h1 = tanh(W_1 [s1, cv] + b_1)                  224 -> 64
//   h2 = tanh(W_2 h1 + b_2),  h3 = tanh(W_3 h2 + b_3)
//   a  = sigmoid(W_o h3 + b_o)                     64 -> 4
// ~28 k FMA per env-step: this IS GEMM-shaped, with the batch as the N
// dimension, so the layers run on the matrix cores.  Since round 3 every
// kernel here (both sweeps, both concurrent-mode kernels, the closed-loop
// evaluation) uses v_mfma_f32_32x32x16_f16 on operands split into two fp16
// terms - three products per k-block, as exact as v_mfma_f32_32x32x2_f32
// (rounds 1-2) and 2.6 x faster per layer (policy_mfma16.h).
//
// Mapping.  A wave owns 32 trajectories: lane l works for trajectory l & 31,
// both half-waves carry the same state / window registers (the ~600-op
// dynamics are computed twice, which is cheaper than any exchange).  For
// D = A B + C with A = weights [32 outputs x 2 k], B = activations
// [2 k x 32 trajectories]:
//   A operand: lane l supplies A[l & 31][l >> 5]
//   B operand: lane l supplies B[l >> 5][l & 31]
//   C / D    : register i of lane l is row r(i) + 4 (l >> 5), column l & 31,
//              with r(i) = (i & 3) + 8 (i >> 2).
// Hence accumulator register i of a layer's output IS the B operand of the
// next layer for the k-pair (r(i), r(i) + 4) - layers chain with no shuffles;
// tanh / relu are applied to the accumulator registers in place.  The weights
// are gathered once per workgroup into LDS in A-operand order
// ([row block][k pair][lane], conflict-free ds_read_b32 per MFMA), so the
// C ABI takes the plain row-major torch parameters.
// One workgroup = 8 waves = 256 trajectories per CU; while one wave of a SIMD
// multiplies, the other runs dynamics / tanh on the VALU.
//
// Parameter gradients: the reverse sweep writes the pre-activation cotangent
// planes; the host reduces them against the saved activation planes with
// apg_planes_gemm.
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
template <bool XMAX>
__global__ __launch_bounds__(kThreads) void mlp_rollout_fwd_kernel(FwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCfLds);
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  // lanes beyond the batch (and the upper half for per-trajectory stores)
  // get an out-of-range buffer offset: their loads return 0, their stores are
  // dropped by the range check - no branch anywhere in the step loop
  const bool live = b < B;
  const bool st_lo = live && hi == 0;  // per-trajectory stores: lower half only
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pin(A.in_ref, 2 * kH * kRD, pitchB);
  const Planes Pst(A.states, kH * 12, pitchB), Pac(A.actions, kH * 4, pitchB);
  const Planes Pfe(A.feat, kNF, pitchN), Px1(A.x1, kN1, pitchN);
  const Planes Ph(A.h, 3 * kW, pitchN), Pmk(A.mask, 5, pitchN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vb_lo = st_lo ? vb : kDead;

  float s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = Ps0.ld(vb, i * pitchB);
  // sliding reference window, raw values: columns 0..4 in the lower half,
  // 4..8 in the upper half (see cfwd_weight)
  const unsigned vwin = live ? vb + (hi ? 4u * pitchB : 0u) : kDead;
  float w[kH][5];
#pragma unroll
  for (int r = 0; r < kH; ++r)
#pragma unroll
    for (int j = 0; j < 5; ++j) w[r][j] = Pin.ld(vwin, (r * kRD + j) * pitchB);
  float wmax_raw = 0.f;   // largest |window value| of the rows seen so far (A.xmax)
  if (XMAX) {
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) wmax_raw = fmaxf(wmax_raw, fabsf(w[r][j]));
  }

#pragma unroll 1
  for (int k = 0; k < kH; ++k) {
    const unsigned pB = opaque(pitchB), pN = opaque(pitchN);
    const unsigned col = (unsigned)b * 4u + (unsigned)k * pitchB;  // column k*B + b
    const unsigned vn = live ? col : kDead;
    const unsigned vn_lo = st_lo ? col : kDead;
    const unsigned vr = live ? col + (hi ? 4u * pitchN : 0u) : kDead;   // + row 4 hi
    const unsigned vc = live ? col + (hi ? 32u * pitchN : 0u) : kDead;  // + channel 4 hi
    const unsigned vm = live ? col + (hi ? pitchN : 0u) : kDead;        // + mask word hi
    const Trig t = make_trig(&s[3]);
    float feat[kNF];
    quad_features(s, t, feat);
#pragma unroll
    for (int j = 0; j < kNF; ++j) Pfe.st(vn_lo, j * pN, feat[j]);
    float xm_c = 0.f;   // largest relu(conv) of this step (see A.xmax)
    if (XMAX) {         // feature and window maxima: transient
      float xm_f = 0.f;
#pragma unroll
      for (int j = 0; j < kNF; ++j) xm_f = fmaxf(xm_f, fabsf(feat[j]));
      // window values are raw - position (columns 0..2, lower half): a bound
      const float xm_i = wmax_raw + (hi ? 0.f : fmaxf(fmaxf(fabsf(s[0]), fabsf(s[1])), fabsf(s[2])));
      const float rf = wave_fmax(xm_f), ri = wave_fmax(xm_i);
      if (lane == 0) {
        float *q = A.xmax + ((size_t)(blockIdx.x * (kThreads / 64) + wave) * kH + k) * 4;
        q[1] = rf, q[2] = ri;
      }
    }

    // the policy on the 16-bit matrix pipe (policy_mfma16.h): every operand as
    // two fp16 terms, three products per k-block
    f32x16 u[2], a[2];
    init_bias(u, L, hTbs);
    {  // state branch: one k-block, features 8 hi .. 8 hi + 7
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
      const Op16 x = split8(v);
      u[0] = mma3(L16.A(hA, nS + 0), x, u[0]);
      u[1] = mma3(L16.A(hA, nS + 1), x, u[1]);
    }
    // the window relative to the current position, split once per step: high
    // term in the low half-word, low term in the high half-word
    const float sub[3] = {hi ? 0.f : s[0], hi ? 0.f : s[1], hi ? 0.f : s[2]};
    init_bias(a, L, hTb1);
    unsigned mbits[3] = {0u, 0u, 0u};
#pragma unroll
    for (int pp = 0; pp < kNP / 2; ++pp) {
      float rv[24];  // relu(conv) of positions 2 pp, 2 pp + 1: registers 0..11 each
      // window rows 2 pp .. 2 pp + 3 relative to the current position, split:
      // high term in the low half-word, low term in the high half-word
      unsigned ws[4][5];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float xv = j < 3 ? w[2 * pp + r][j] - sub[j] : w[2 * pp + r][j];
          const _Float16 vh = (_Float16)xv, vl = (_Float16)(xv - (float)vh);
          const h16x2 pr = {vh, vl};
          ws[r][j] = __builtin_bit_cast(unsigned, pr);
        }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pos = 2 * pp + e;
        f32x16 cv;
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[i] = L.T(hTbc + i * 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          Op16 x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // slots 2 q, 2 q + 1 of this k-block
            const int s0 = kb * 8 + 2 * q, s1 = s0 + 1;
            const unsigned r0 = ws[e + s0 % 3][s0 / 3];
            const unsigned r1 = s1 < 15 ? ws[e + s1 % 3][s1 < 15 ? s1 / 3 : 0] : 0u;
            x.h[q] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);  // low half-words
            x.l[q] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);  // high half-words
          }
          cv = mma3(L16.A(hA, nC + kb), x, cv);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {  // rows r(i) + 4 hi < 20 are real channels
          float v = cv[i];
          mbits[i >> 2] |= (v > 0.f ? 1u : 0u) << ((i & 3) * 8 + pos);
          v = fmaxf(v, 0.f);
          if (XMAX) xm_c = fmaxf(xm_c, v);
          // plane 64 + (r(i) + 4 hi) * 8 + pos: the 4 hi * 8 rows are in vc
          Px1.st(i < 8 ? vc : vn_lo, (kW + rrow(i) * kNP + pos) * pN, v);
          rv[e * 12 + i] = v;
        }
      }
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
        const Op16 x = split8(v);
        a[0] = mma3(L16.A(hA, n1c + (0 * 4 + pp) * 3 + kb), x, a[0]);
        a[1] = mma3(L16.A(hA, n1c + (1 * 4 + pp) * 3 + kb), x, a[1]);
      }
    }
    // relu mask, trajectory-indexed: bit e = ch*8 + pos of word e >> 5
#pragma unroll
    for (int g = 0; g < 3; ++g)
      Pmk.stu(g < 2 ? vm : vn_lo, 2 * g * pN, mbits[g]);
    if (XMAX) {
      const float rc = wave_fmax(xm_c);
      if (lane == 0)
        A.xmax[((size_t)(blockIdx.x * (kThreads / 64) + wave) * kH + k) * 4] = rc;
    }
    // fc1 state part on s1 = tanh(states_in); h1 -> h2 -> h3 (the tanh of a
    // layer is applied, and stored, where the next layer consumes it)
    dense64_16(a, u, L16, hA, n1s, [&](int rb, int i, float v) {
      const float tv = tanh_fast(v);
      Px1.st(vr, (rb * 32 + rrow(i)) * pN, tv);
      return tv;
    });
    init_bias(u, L, hTb2);
    dense64_16(u, a, L16, hA, n2, [&](int rb, int i, float v) {
      const float tv = tanh_fast(v);
      Ph.st(vr, (rb * 32 + rrow(i)) * pN, tv);
      return tv;
    });
    init_bias(a, L, hTb3);
    dense64_16(a, u, L16, hA, n3, [&](int rb, int i, float v) {
      const float tv = tanh_fast(v);
      Ph.st(vr, (kW + rb * 32 + rrow(i)) * pN, tv);
      return tv;
    });
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a[rb][i] = tanh_fast(a[rb][i]);
        Ph.st(vr, (2 * kW + rb * 32 + rrow(i)) * pN, a[rb][i]);
      }
    // head on the VALU: each half sums its 32 of the 64 inputs
    float act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z0 = 0.f, z1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        z0 = fmaf(L.T(hTo + ((j * 2 + 0) * 16 + i) * 2), a[0][i], z0);
        z1 = fmaf(L.T(hTo + ((j * 2 + 1) * 16 + i) * 2), a[1][i], z1);
      }
      float z = z0 + z1;
      z += other_half(z);
      act[j] = sigmoidf_(z + L.U(hBo + j));
      Pac.st(vb_lo, (k * 4 + j) * pB, act[j]);
    }
    quad_step(s, act, c, t);
#pragma unroll
    for (int i = 0; i < 12; ++i) Pst.st(vb_lo, (k * 12 + i) * pB, s[i]);
    if (k + 1 < kH) {
#pragma unroll
      for (int r = 0; r + 1 < kH; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) w[r][j] = w[r + 1][j];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[kH - 1][j] = Pin.ld(vwin, ((k + kH) * kRD + j) * pB);
      if (XMAX) {
#pragma unroll
        for (int j = 0; j < 5; ++j) wmax_raw = fmaxf(wmax_raw, fabsf(w[kH - 1][j]));
      }
    }
  }
}

// ------------------------------------------------------ closed-loop evaluation
// N2: QuadEvaluator.follow_trajectory("rand")
// (scripts/evaluate_drone.py:81-194) for a batch of reference trajectories in
// ONE launch: per step Random.get_ref_traj (window = rows cur+1 .. cur+H,
// neural_control/trajectory/random_traj.py:60-79), QuadDataset.prepare_data
// (window -> [ref_pos - pos, ref_vel, ref_vel - vel], dataset.py:155-204),
// the policy, QuadRotorEnvBase.step (clip + dynamics + attitude check,
// drone_env.py:59-117), project_on_ref / divergence, and either the break
// (test_time) or the reset to the reference state (self-play data).
// Same matrix-core policy evaluation as the forward sweep, no saved planes.
struct LoopArgs {
  const float *traj;  // [L][9][B] (position, euler, velocity) rows
  float *div;         // [T][B]
  int *steps;         // [B] iterations executed
  float *drone;       // [T+1][12][B] or NULL: states after each step
  float *actions;     // [T][4][B] or NULL
  float *start;       // [T][12][B] or NULL: states the policy saw
  const float *tables;
  QuadConst c;
  int B, L, T, test_time;
  float thresh_div, thresh_stable;
  int learnt;         // the environment is a LearntDynamics: its packed weights
                      // follow the policy tables (learnt_residual.h)
};

// LEARNT: the environment is a LearntDynamics (a second instantiation, so that the
// analytic loop keeps its registers)
template <bool LEARNT>
__global__ __launch_bounds__(kThreads) void mlp_closed_loop_kernel(LoopArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCfLds + (LEARNT ? kLearntFloats : 0));
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B, T = A.T;
  const bool live = b < B;
  const bool st_lo = live && hi == 0;
  const unsigned pitchB = (unsigned)B * 4u;
  const QuadConst c = A.c;
  // a NULL output becomes an empty buffer: every store to it is dropped
  const Planes Ptr(A.traj, A.L * 9, pitchB), Pdv(A.div, T, pitchB);
  const Planes Pdr(A.drone, A.drone ? (T + 1) * 12 : 0, pitchB);
  const Planes Pac(A.actions, A.actions ? T * 4 : 0, pitchB);
  const Planes Pss(A.start, A.start ? T * 12 : 0, pitchB);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  // window columns of this half-wave: lower (x, y, z, vx, -), upper (vy, vz,
  // vx, vy, vz) - policy channels 0-3 / 4-8 (see cfwd_weight); trajectory
  // columns 6..8 are the velocity
  unsigned vcol[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int lo = j < 3 ? j : 6, up = j < 2 ? 7 + j : 4 + j;
    vcol[j] = live ? vb + (unsigned)(hi ? up : lo) * pitchB : kDead;
  }
  float s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = i < 3 ? Ptr.ld(vb, i * pitchB) : 0.f;  // zero_reset
  float w[kH][5];  // rows cur + 1 .. cur + H of the trajectory
#pragma unroll
  for (int r = 0; r < kH; ++r)
#pragma unroll
    for (int j = 0; j < 5; ++j) w[r][j] = Ptr.ld(vcol[j], ((1 + r) * 9) * pitchB);
#pragma unroll
  for (int i = 0; i < 12; ++i) Pdr.st(st_lo ? vb : kDead, i * pitchB, s[i]);
  bool alive = live;
  int steps = 0;

#pragma unroll 1
  for (int k = 0; k < T; ++k) {
    const unsigned pB = opaque(pitchB);
    const unsigned vrec = (alive && hi == 0) ? vb : kDead;
#pragma unroll
    for (int i = 0; i < 12; ++i) Pss.st(vrec, (k * 12 + i) * pB, s[i]);
    const Trig t = make_trig(&s[3]);
    float feat[kNF];
    quad_features(s, t, feat);
    // the policy on the 16-bit matrix pipe (policy_mfma16.h), as the forward
    // training sweep, nothing saved
    f32x16 u[2], a[2];
    init_bias(u, L, hTbs);
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
      const Op16 x = split8(v);
      u[0] = mma3(L16.A(hA, nS + 0), x, u[0]);
      u[1] = mma3(L16.A(hA, nS + 1), x, u[1]);
    }
    init_bias(a, L, hTb1);
    // lower: position columns relative to the drone; upper: the last three
    // columns are reference velocity minus drone velocity
    const float sub[5] = {hi ? 0.f : s[0], hi ? 0.f : s[1], hi ? s[6] : s[2],
                          hi ? s[7] : 0.f, hi ? s[8] : 0.f};
#pragma unroll
    for (int pp = 0; pp < kNP / 2; ++pp) {
      float rv[24];
      unsigned ws[4][5];  // window rows 2 pp .. 2 pp + 3, split (high | low << 16)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float xv = w[2 * pp + r][j] - sub[j];
          const _Float16 vh = (_Float16)xv, vl = (_Float16)(xv - (float)vh);
          const h16x2 pr = {vh, vl};
          ws[r][j] = __builtin_bit_cast(unsigned, pr);
        }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x16 cv;
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[i] = L.T(hTbc + i * 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          Op16 x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int s0 = kb * 8 + 2 * q, s1 = s0 + 1;
            const unsigned r0 = ws[e + s0 % 3][s0 / 3];
            const unsigned r1 = s1 < 15 ? ws[e + s1 % 3][s1 < 15 ? s1 / 3 : 0] : 0u;
            x.h[q] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);
            x.l[q] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);
          }
          cv = mma3(L16.A(hA, nC + kb), x, cv);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) rv[e * 12 + i] = fmaxf(cv[i], 0.f);
      }
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
        const Op16 x = split8(v);
        a[0] = mma3(L16.A(hA, n1c + (0 * 4 + pp) * 3 + kb), x, a[0]);
        a[1] = mma3(L16.A(hA, n1c + (1 * 4 + pp) * 3 + kb), x, a[1]);
      }
    }
    const auto th = [](int, int, float v) { return tanh_fast(v); };
    dense64_16(a, u, L16, hA, n1s, th);
    init_bias(u, L, hTb2);
    dense64_16(u, a, L16, hA, n2, th);
    init_bias(a, L, hTb3);
    dense64_16(a, u, L16, hA, n3, th);
    float act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z0 = 0.f, z1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        z0 = fmaf(L.T(hTo + ((j * 2 + 0) * 16 + i) * 2), tanh_fast(a[0][i]), z0);
        z1 = fmaf(L.T(hTo + ((j * 2 + 1) * 16 + i) * 2), tanh_fast(a[1][i]), z1);
      }
      float z = z0 + z1;
      z += other_half(z);
      act[j] = fminf(fmaxf(sigmoidf_(z + L.U(hBo + j)), 0.f), 1.f);  // np.clip
      Pac.st(vrec, (k * 4 + j) * pB, act[j]);
    }
    if (LEARNT) learnt_quad_step(s, act, c, t, lds + kCfLds, hi);
    else quad_step(s, act, c, t);
    // window row 0 is reference[cur] after get_ref_traj: project_on_ref
    float d2 = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float oth = other_half(w[0][q]);
      const float e = (hi ? oth : w[0][q]) - s[q];
      d2 = fmaf(e, e, d2);
    }
    const float dv = sqrtf(d2);
    const bool stable = fabsf(s[3]) < A.thresh_stable && fabsf(s[4]) < A.thresh_stable;
    const bool failed = dv > A.thresh_div || !stable;
#pragma unroll
    for (int i = 0; i < 12; ++i) Pdr.st(vrec, ((k + 1) * 12 + i) * pB, s[i]);
    Pdv.st(vrec, k * pB, dv);
    if (alive) steps = k + 1;
    if (A.test_time) {
      alive = alive && !failed;
      if (!__any(alive)) break;
    } else if (__any(failed)) {  // get_current_full_state: row cur, zero rates
      const int cur = k + 1 < A.L - kH ? k + 1 : A.L - kH;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float rv = Ptr.ld(vb, (cur * 9 + i) * pB);
        s[i] = failed ? rv : s[i];
      }
#pragma unroll
      for (int i = 9; i < 12; ++i) s[i] = failed ? 0.f : s[i];
    }
    if (k + 2 <= A.L - kH) {  // get_ref_traj advanced: slide, fetch row k+1+H
#pragma unroll
      for (int r = 0; r + 1 < kH; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) w[r][j] = w[r + 1][j];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[kH - 1][j] = Ptr.ld(vcol[j], ((k + 1 + kH) * 9) * pB);
    }
  }
  if (st_lo) A.steps[b] = steps;
}

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

__global__ __launch_bounds__(256) void mlp_pack_cbwd_kernel(PackArgs A) {
  pack_cbwd(A, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// both tables in one launch (the concurrent step runs its sweeps back to
// back): blocks [0, fwd_blocks) write the forward tables at dst, the others the
// reverse tables at dst + kCfLds
__global__ __launch_bounds__(256) void mlp_pack_pair_kernel(PackArgs A, int fwd_blocks) {
  if ((int)blockIdx.x < fwd_blocks) {
    pack_cfwd(A, blockIdx.x * blockDim.x + threadIdx.x, fwd_blocks * blockDim.x);
  } else {
    A.dst += kCfLds;
    pack_cbwd(A, (blockIdx.x - fwd_blocks) * blockDim.x + threadIdx.x,
              (gridDim.x - fwd_blocks) * blockDim.x);
  }
}

struct BwdArgs {
  const float *state0, *states, *actions, *ref, *x1, *h;
  const unsigned *mask;
  float *loss_partials;
  float *d_pre;   // [256][N]: d_pre1, d_pre2, d_pre3, d_pre_s (64 each)
  float *d_zout;  // [4][N]
  float *d_conv;  // [720][B]: window-diagonal sums of the conv cotangents (kConvP)
  float *grad_state0;
  const float *tables;  // packed operand tables (mlp_pack_cbwd_kernel)
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
};

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

__global__ __launch_bounds__(kThreads) void mlp_rollout_bwd_kernel(BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCbLds);
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;  // dead lanes: out-of-range offsets, see forward
  const bool st_lo = live && hi == 0;
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pst(A.states, kH * 12, pitchB);
  const Planes Pac(A.actions, kH * 4, pitchB), Prf(A.ref, kH * A.ref_cols, pitchB);
  const Planes Px1(A.x1, kN1, pitchN), Ph(A.h, 3 * kW, pitchN);
  const Planes Pmk(A.mask, 5, pitchN), Pdp(A.d_pre, 4 * kW, pitchN);
  const Planes Pdz(A.d_zout, 4, pitchN), Pdc(A.d_conv, kConvPlanes, pitchB);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;

  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float loss = 0.f;
  // sliding diagonal sums of the conv cotangents: dgn[ch][ii] = diagonal
  // tau = k + ii of this half-wave's positions (see kConvP)
  float dgn[kNC][4];
#pragma unroll
  for (int ch = 0; ch < kNC; ++ch)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) dgn[ch][ii] = 0.f;
  const unsigned vg = live ? (unsigned)b * 4u + (hi ? kTau * pitchB : 0u) : kDead;
  const unsigned vb_lo = st_lo ? (unsigned)b * 4u : kDead;

#pragma unroll 1
  for (int k = kH - 1; k >= 0; --k) {
    const unsigned pB = opaque(pitchB), pN = opaque(pitchN);
    const unsigned col = (unsigned)b * 4u + (unsigned)k * pitchB;
    const unsigned vn = live ? col : kDead;
    const unsigned vn_lo = st_lo ? col : kDead;
    const unsigned vr = live ? col + (hi ? 4u * pitchN : 0u) : kDead;
    float sn[12], sc[12], a[4], rp[3], rv[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      sn[i] = Pst.ld(vb, (k * 12 + i) * pB);
      sc[i] = k > 0 ? Pst.ld(vb, ((k - 1) * 12 + i) * pB) : Ps0.ld(vb, i * pB);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = Pac.ld(vb, (k * 4 + j) * pB);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rp[i] = Prf.ld(vb, (k * A.ref_cols + i) * pB);
      rv[i] = Prf.ld(vb, (k * A.ref_cols + A.vel_col + i) * pB);
    }
    unsigned mw[5];
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vn, eb * pN);
    float hv[2][16];
    load_acts(hv, Ph, 2 * kW, vr, pN);  // h3
    __builtin_amdgcn_sched_barrier(0);
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = sn[i] - rp[i], dv = sn[6 + i] - rv[i], wn = sn[9 + i];
      lp += dp * dp, lv += dv * dv, lw += wn * wn;
      lam[i] += 2.f * A.w.pos * dp;
      lam[6 + i] += 2.f * A.w.vel * dv;
      lam[9 + i] += 2.f * A.w.av * wn;
    }
    const float da0 = a[0] - 0.5f;
    float ga[4];
    ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      const float d = a[j] - 0.5f;
      lr += d * d;
      ga[j] = 2.f * A.w.rates * d;
    }
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
    const Trig t = make_trig(&sc[3]);
    quad_step_adjoint(lam, ga, a[0], &sc[9], c, t);

    // head (VALU): d/dh3 in accumulator layout
    float dz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dz[j] = ga[j] * a[j] * (1.f - a[j]);
      Pdz.st(vn_lo, j * pN, dz[j]);
    }
    f32x16 d[2], e[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v = fmaf(L.T(gTo + ((j * 2 + rb) * 16 + i) * 2), dz[j], v);
        d[rb][i] = v;
      }
    __builtin_amdgcn_sched_barrier(0);
    // the reverse layers on the 16-bit matrix pipe (policy_mfma16.h):
    // cotangents scaled per trajectory, two fp16 terms, three products
    tanh_adjoint(d, hv, Pdp, 2 * kW, vr, pN);  // d_pre3
    load_acts(hv, Ph, kW, vr, pN);             // h2, lands under the product
    __builtin_amdgcn_sched_barrier(0);
    Op16 x[4];
    zero(e);
    int ex = scaled_split64(d, x);
    dense64T_16(e, x, L16, gA, m3T);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(e, hv, Pdp, kW, vr, pN, ex);  // d_pre2
    load_acts(hv, Ph, 0, vr, pN);              // h1
    __builtin_amdgcn_sched_barrier(0);
    zero(d);
    ex = scaled_split64(e, x);
    dense64T_16(d, x, L16, gA, m2T);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(d, hv, Pdp, 0, vr, pN, ex);   // d_pre1
    load_acts(hv, Px1, 0, vr, pN);             // s1
    __builtin_amdgcn_sched_barrier(0);
    // fc1 inputs, state branch; d_pre1's split also feeds the conv part below
    zero(e);
    const int ex1 = scaled_split64(d, x);
    dense64T_16(e, x, L16, gA, m1sT);
    __builtin_amdgcn_sched_barrier(0);
    tanh_adjoint(e, hv, Pdp, 3 * kW, vr, pN, ex1);  // d_pre_s
    // features: one 32-row block (15 real rows), then both halves need all
    f32x16 f;
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = 0.f;
    {
      Op16 xs[4];
      const int exs = scaled_split64(e, xs);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) f = mma3(L16.A(gA, mST + kb), xs[kb], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_ldexpf(f[i], exs);
    }
    float dfeat[kNF], gs[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float own = f[i], oth = other_half(own);
      dfeat[rrow(i)] = hi ? oth : own;
      if (rrow(i) + 4 < kNF) dfeat[rrow(i) + 4 < kNF ? rrow(i) + 4 : 0] = hi ? own : oth;
    }
    quad_features_adjoint(sc, t, dfeat, gs);
#pragma unroll
    for (int i = 0; i < 12; ++i) lam[i] += gs[i];
    // fc1 inputs, conv outputs: five 32-row blocks over e = ch*8 + pos
    float dpos[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) {
      f32x16 y;
#pragma unroll
      for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) y = mma3(L16.A(gA, m1cT + eb * 4 + kb), x[kb], y);
#pragma unroll
      for (int i = 0; i < 16; ++i) y[i] = __builtin_amdgcn_ldexpf(y[i], ex1);
      const unsigned mws = hi ? mw[eb] >> 4 : mw[eb];  // bit r(i) + 4 hi -> bit r(i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3: channel eb*4 + g,
        const int ch = eb * 4 + g;     // positions ii + 4 hi
        float sum = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * g + ii;
          const float dcp = ((mws >> rrow(i)) & 1u) ? y[i] : 0.f;
          dgn[ch][ii] += dcp;
          sum += dcp;
        }
        // diagonal tau = k + 3 is complete; the others move up one position
        Pdc.st(vg, (unsigned)(ch * 2 * kTau + k + 3) * pB, dgn[ch][3]);
        dgn[ch][3] = dgn[ch][2], dgn[ch][2] = dgn[ch][1], dgn[ch][1] = dgn[ch][0];
        dgn[ch][0] = 0.f;
        Pdc.st(vb_lo, (unsigned)(kConvP + ch * kH + k) * pB, sum + other_half(sum));
#pragma unroll
        for (int q = 0; q < 3; ++q)
          dpos[q] = fmaf(L.U(gAq + (eb * 4 + g) * 3 + q), sum, dpos[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) lam[q] -= dpos[q] + other_half(dpos[q]);
  }
  // the diagonals tau = 0..2 (after the last shift they sit in slots 1..3)
#pragma unroll
  for (int ch = 0; ch < kNC; ++ch)
#pragma unroll
    for (int tau = 0; tau < 3; ++tau)
      Pdc.st(vg, (unsigned)(ch * 2 * kTau + tau) * pitchB, dgn[ch][tau + 1]);
  if (st_lo && A.grad_state0)
#pragma unroll
    for (int i = 0; i < 12; ++i) A.grad_state0[(size_t)i * B + b] = lam[i];
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
}


// ---------------------------------------------- concurrent mode, policy fused
// The concurrent training step (BASELINE config 2, the headline workload) with
// the policy inside: TrainBase.run_epoch's concurrent branch
// (scripts/train_base.py:198-204: actions = sigmoid(net(in_state, in_ref)),
// reshape [B, H, 4]) + TrainDrone.train_controller_model
// (scripts/train_drone.py:175-203: H x dynamics, quad_mpc_loss, backward).
// The network runs ONCE per trajectory (40 outputs = H x 4 actions), then the
// register-resident rollout and its adjoint (as quad.hip), then - second
// kernel - the reverse pass of the network from dL/d(head pre-activations).
// Planes are [feature][B]; the weight gradients come from apg_planes_gemm.


struct ConcArgs {
  const float *feat, *in_ref, *state0, *ref;  // [15][B], [H][9][B], [12][B], [H][C][B]
  float *x1, *h;        // [224][B], [192][B]
  unsigned *mask;       // [5][B]
  float *d_zout;        // [40][B]
  float *d_pre, *d_conv;  // [256][B], [160][B] (second kernel)
  float *states;        // [H][12][B] or NULL
  float *loss_partials;
  const float *tables;
  // [waves][4] or NULL: max |conv output|, |feature|, |in_ref| of each wave's 32
  // trajectories (the trajectory-major reverse kernel scales by them)
  float *xmax;
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
  // ROWS (below): feat / in_ref / state0 / ref are the DATA SET's tensors [N][ld_*]
  // and `index` [B] names this batch's rows; the feature and window planes the
  // reverse kernel reads are written to o_feat [15][B] / o_in_ref [90][B]
  const long long *index;
  float *o_feat, *o_in_ref;
  int ld_feat, ld_in_ref, ld_state0, ld_ref;
  unsigned bytes_feat, bytes_in_ref, bytes_state0, bytes_ref;
};

// ROWS: the minibatch gather folded into this kernel (VERDICT r4 next #4;
// TrainBase.run_epoch's batch selection, scripts/train_base.py:191-194).  The
// workgroup's rows are brought into LDS through the index (gather_rows_issue,
// policy_mfma.h) before the operand tables - features + windows [256][15] /
// [256][91] where the tables go afterwards, the start states [256][13] behind the
// tables - and the reference rows [256][91] over the tables once the policy is
// done with them, while the rollout runs.
constexpr int kRowPadW = kH * kRD + 1, kRowPadF = kNF, kRowPadS = 13;   // odd strides
constexpr int zWin = 0, zFeat = kTrajPerBlock * kRowPadW,               // floats
              zS0 = kCfLds, zRows = zS0 + kTrajPerBlock * kRowPadS,
              kCfRowsLds = zRows + kTrajPerBlock;
static_assert(zFeat + kTrajPerBlock * kRowPadF <= kCfLds, "staging under the tables");
static_assert(kCfRowsLds * 4 <= 160 * 1024, "LDS");

template <bool ROWS>
__global__ __launch_bounds__(kThreads) void mlp_concurrent_fwd_kernel(ConcArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (ROWS) {
    const int t = threadIdx.x, b_ = blockIdx.x * kTrajPerBlock + t;
    // (a dead trajectory reads the batch's last row: finite data, never stored)
    if (t < kTrajPerBlock)
      reinterpret_cast<int *>(lds + zRows)[t] = (int)A.index[b_ < A.B ? b_ : A.B - 1];
    __syncthreads();
    const int *rows = reinterpret_cast<const int *>(lds + zRows);
    gather_rows_issue<kRowPadW>(lds + zWin, rows, A.in_ref, A.bytes_in_ref, A.ld_in_ref,
                                kH * kRD);
    gather_rows_issue<kRowPadF>(lds + zFeat, rows, A.feat, A.bytes_feat, A.ld_feat, kNF);
    gather_rows_issue<kRowPadS>(lds + zS0, rows, A.state0, A.bytes_state0, A.ld_state0, 12);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    fill_lds(lds, A.tables, kCfLds);
  }
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const bool st_lo = live && hi == 0;
  const unsigned pN = (unsigned)B * 4u;  // every plane here is [..][B]
  const QuadConst c = A.c;
  const Planes Pfe(A.feat, kNF, pN), Pin(A.in_ref, kH * kRD, pN);
  const Planes Ps0(A.state0, 12, pN), Prf(A.ref, kH * A.ref_cols, pN);
  const Planes Px1(A.x1, kN1, pN), Ph(A.h, 3 * kW, pN), Pmk(A.mask, 5, pN);
  const Planes Pdz(A.d_zout, kNA, pN);
  const Planes Pst(A.states, A.states ? kH * 12 : 0, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vb_lo = st_lo ? vb : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;   // + row 4 hi
  const unsigned vc = live ? vb + (hi ? 32u * pN : 0u) : kDead;  // + channel 4 hi
  const unsigned vm = live ? vb + (hi ? pN : 0u) : kDead;        // + mask word hi

  float feat[kNF];
  float w[kH][5];  // policy reference input, columns 0..4 / 4..8 per half
  const int tl = wave * 32 + (lane & 31);   // this lane's trajectory of the workgroup
  if (ROWS) {
    const float *pf = lds + zFeat + tl * kRowPadF, *pw = lds + zWin + tl * kRowPadW + 4 * hi;
#pragma unroll
    for (int j = 0; j < kNF; ++j) feat[j] = pf[j];
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) w[r][j] = pw[r * kRD + j];
    __syncthreads();                      // every wave has its rows: the tables may land
    fill_lds_issue(lds, A.tables, kCfLds);
  } else {
#pragma unroll
    for (int j = 0; j < kNF; ++j) feat[j] = Pfe.ld(vb, j * pN);
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) w[r][j] = Pin.ld(vr, (r * kRD + j) * pN);
  }
  // (maxima of the |v| BIT PATTERNS, unsigned: inf / NaN lie above every finite
  // value - see TmMeta)
  unsigned xm_conv = 0u;
  const auto umax = [](unsigned m, float v) {
    const unsigned b_ = __builtin_bit_cast(unsigned, v) & 0x7fffffffu;
    return b_ > m ? b_ : m;
  };
  if (A.xmax) {   // (wave-uniform)
    unsigned mf = 0u, mi = 0u;
#pragma unroll
    for (int j = 0; j < kNF; ++j) mf = umax(mf, feat[j]);
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) mi = umax(mi, w[r][j]);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned of = (unsigned)__shfl_xor((int)mf, sft, 64),
                     oi = (unsigned)__shfl_xor((int)mi, sft, 64);
      mf = of > mf ? of : mf, mi = oi > mi ? oi : mi;
    }
    if (lane == 0) {
      unsigned *q = reinterpret_cast<unsigned *>(A.xmax) +
                    (size_t)(blockIdx.x * (kThreads / 64) + wave) * 4;
      q[1] = mf, q[2] = mi;
    }
  }

  if (ROWS) {   // the table DMA issued above
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the reverse kernel reads the feature / window blocks of x^T from the data
    // set's rows itself: no planes of them are written)
  }
  // ---- policy forward on the 16-bit matrix pipe (policy_mfma16.h): every
  // operand as two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  f32x16 u[2], a[2];
  init_bias(u, L, hTbs);
  {  // states_in: one k-block, features 8 hi .. 8 hi + 7
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
    const Op16 x = split8(v);
    u[0] = mma3(L16.A(hA, nS + 0), x, u[0]);
    u[1] = mma3(L16.A(hA, nS + 1), x, u[1]);
  }
  init_bias(a, L, hTb1);
  unsigned mbits[3] = {0u, 0u, 0u};
#pragma unroll
  for (int pp = 0; pp < kNP / 2; ++pp) {
    float rv[24];  // relu(conv) of positions 2 pp, 2 pp + 1: registers 0..11 each
    // window rows 2 pp .. 2 pp + 3, split: high term in the low half-word, low
    // term in the high half-word of one register per value
    unsigned ws[4][5];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float xv = w[2 * pp + r][j];
        const _Float16 vh = (_Float16)xv, vl = (_Float16)(xv - (float)vh);
        const h16x2 pr = {vh, vl};
        ws[r][j] = __builtin_bit_cast(unsigned, pr);
      }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pos = 2 * pp + e;
      f32x16 cv;
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = L.T(hTbc + i * 2);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        Op16 x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // slots 2 q, 2 q + 1 of this k-block
          const int s0 = kb * 8 + 2 * q, s1 = s0 + 1;
          const unsigned r0 = ws[e + s0 % 3][s0 / 3];
          const unsigned r1 = s1 < 15 ? ws[e + s1 % 3][s1 < 15 ? s1 / 3 : 0] : 0u;
          x.h[q] = __builtin_amdgcn_perm(r1, r0, 0x05040100u);  // low half-words
          x.l[q] = __builtin_amdgcn_perm(r1, r0, 0x07060302u);  // high half-words
        }
        cv = mma3(L16.A(hA, nC + kb), x, cv);
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        float v = cv[i];
        mbits[i >> 2] |= (v > 0.f ? 1u : 0u) << ((i & 3) * 8 + pos);
        xm_conv = umax(xm_conv, v);   // (before the relu: it would drop a NaN)
        v = fmaxf(v, 0.f);
        Px1.st(i < 8 ? vc : vb_lo, (kW + rrow(i) * kNP + pos) * pN, v);
        rv[e * 12 + i] = v;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
      const Op16 x = split8(v);
      a[0] = mma3(L16.A(hA, n1c + (0 * 4 + pp) * 3 + kb), x, a[0]);
      a[1] = mma3(L16.A(hA, n1c + (1 * 4 + pp) * 3 + kb), x, a[1]);
    }
  }
#pragma unroll
  for (int g = 0; g < 3; ++g) Pmk.stu(g < 2 ? vm : vb_lo, 2 * g * pN, mbits[g]);
  if (A.xmax) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned o = (unsigned)__shfl_xor((int)xm_conv, sft, 64);
      xm_conv = o > xm_conv ? o : xm_conv;
    }
    if (lane == 0)
      reinterpret_cast<unsigned *>(A.xmax)[(size_t)(blockIdx.x * (kThreads / 64) + wave) * 4] =
          xm_conv;
  }
  // fc1 state part on s1 = tanh(states_in), stored as the reverse pass needs it
  dense64_16(a, u, L16, hA, n1s, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Px1.st(vr, (rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  init_bias(u, L, hTb2);
  dense64_16(u, a, L16, hA, n2, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  init_bias(a, L, hTb3);
  dense64_16(a, u, L16, hA, n3, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (kW + rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  // head: 40 outputs = row block 0 + rows 0..7 of row block 1
  init_bias(u, L, hTbo);
  dense64_16(u, a, L16, hA, nO, [&](int rb, int i, float v) {
    const float tv = tanh_fast(v);
    Ph.st(vr, (2 * kW + rb * 32 + rrow(i)) * pN, tv);
    return tv;
  });
  // every lane needs all 40 actions (both halves run the same rollout)
  float act[kH][4];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) {
    const float own = cc < 16 ? u[0][cc] : u[1][cc - 16], oth = other_half(own);
    const int row = khead(cc, 0);
    act[row >> 2][row & 3] = sigmoidf_(hi ? oth : own);
    act[(row + 4) >> 2][(row + 4) & 3] = sigmoidf_(hi ? own : oth);
  }

  // ---- rollout + adjoint in registers (quad_rollout_reg_kernel's structure)
  float s[12];
  if (ROWS) {
    // the tables are dead: the reference rows land over them while the rollout runs
    __syncthreads();
    gather_rows_issue<kRowPadW>(lds + zWin, reinterpret_cast<const int *>(lds + zRows), A.ref,
                                A.bytes_ref, A.ld_ref, kH * A.ref_cols);
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = lds[zS0 + tl * kRowPadS + i];
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = Ps0.ld(vb, i * pN);
  }
  Trig st_trig[kH];
  float st_w[kH + 1][3], st_pv[kH][6];
#pragma unroll
  for (int k = 0; k < kH; ++k) {
#pragma unroll
    for (int i = 0; i < 3; ++i) st_w[k][i] = s[9 + i];
    st_trig[k] = make_trig(&s[3]);
    quad_step(s, act[k], c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) st_pv[k][i] = s[i], st_pv[k][3 + i] = s[6 + i];
#pragma unroll
    for (int i = 0; i < 12; ++i) Pst.st(vb_lo, (k * 12 + i) * pN, s[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[kH][i] = s[9 + i];
  float loss = 0.f, lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float rp[3], rv[3];
  const float *pr = lds + zWin + tl * kRowPadW;
  if (ROWS) {   // the reference rows (and every store so far) have landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rp[i] = ROWS ? pr[(kH - 1) * A.ref_cols + i]
                 : Prf.ld(vb, ((kH - 1) * A.ref_cols + i) * pN);
    rv[i] = ROWS ? pr[(kH - 1) * A.ref_cols + A.vel_col + i]
                 : Prf.ld(vb, ((kH - 1) * A.ref_cols + A.vel_col + i) * pN);
  }
#pragma unroll
  for (int k = kH - 1; k >= 0; --k) {
    float np_[3], nv_[3];  // next iteration's reference row, one step ahead
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      np_[i] = k == 0 ? 0.f
               : ROWS ? pr[(k - 1) * A.ref_cols + i]
                      : Prf.ld(vb, ((k - 1) * A.ref_cols + i) * pN);
      nv_[i] = k == 0 ? 0.f
               : ROWS ? pr[(k - 1) * A.ref_cols + A.vel_col + i]
                      : Prf.ld(vb, ((k - 1) * A.ref_cols + A.vel_col + i) * pN);
    }
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rp[i], dv = st_pv[k][3 + i] - rv[i];
      const float wn = st_w[k + 1][i];
      lp += dp * dp, lv += dv * dv, lw += wn * wn;
      lam[i] += 2.f * A.w.pos * dp;
      lam[6 + i] += 2.f * A.w.vel * dv;
      lam[9 + i] += 2.f * A.w.av * wn;
    }
    const float a0 = act[k][0], da0 = a0 - 0.5f;
    float ga[4];
    ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = act[k][i] - 0.5f;
      lr += d * d;
      ga[i] = 2.f * A.w.rates * d;
    }
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
    quad_step_adjoint(lam, ga, a0, st_w[k], c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 4; ++i)  // dL/d(head pre-activation), in place
      act[k][i] = ga[i] * act[k][i] * (1.f - act[k][i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) rp[i] = np_[i], rv[i] = nv_[i];
  }
  // own rows of dL/dz in accumulator layout: rows khead(c, hi)
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) {
    const int row = khead(cc, 0);
    const float v = hi ? act[(row + 4) >> 2][(row + 4) & 3] : act[row >> 2][row & 3];
    Pdz.st(vr, row * pN, v);
  }
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
}

__global__ __launch_bounds__(kThreads) void mlp_concurrent_bwd_kernel(ConcArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kCbLds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Px1(A.x1, kN1, pN), Ph(A.h, 3 * kW, pN), Pmk(A.mask, 5, pN);
  const Planes Pdz(A.d_zout, kNA, pN), Pdp(A.d_pre, 4 * kW, pN);
  const Planes Pdc(A.d_conv, kNC * kNP, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;

  float dzr[20];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) dzr[cc] = Pdz.ld(vr, khead(cc, 0) * pN);
  unsigned mw[5];
#pragma unroll
  for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vb, eb * pN);
  float hv[2][16];
  load_acts(hv, Ph, 2 * kW, vr, pN);  // h3
  __builtin_amdgcn_sched_barrier(0);
  // the reverse layers on the 16-bit matrix pipe: cotangents scaled per
  // trajectory, split into two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  f32x16 d[2], e[2];
  zero(d);
  int ex;
  {  // dL/dh3 = W_out^T dL/dz: this lane's 20 rows fill 2.5 k-blocks
    float amax = 0.f;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amax = fmaxf(amax, fabsf(dzr[cc]));
    ex = scale_exponent(amax);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = kb * 8 + j < 20 ? __builtin_amdgcn_ldexpf(dzr[kb * 8 + j < 20 ? kb * 8 + j : 0], -ex)
                               : 0.f;
      const Op16 x = split8(v);
      d[0] = mma3(L16.A(gA, mOT + kb), x, d[0]);
      d[1] = mma3(L16.A(gA, mOT + 3 + kb), x, d[1]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(d, hv, Pdp, 2 * kW, vr, pN, ex);  // d_pre3
  load_acts(hv, Ph, kW, vr, pN);                 // h2
  __builtin_amdgcn_sched_barrier(0);
  Op16 x[4];
  zero(e);
  ex = scaled_split64(d, x);
  dense64T_16(e, x, L16, gA, m3T);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(e, hv, Pdp, kW, vr, pN, ex);      // d_pre2
  load_acts(hv, Ph, 0, vr, pN);                  // h1
  __builtin_amdgcn_sched_barrier(0);
  zero(d);
  ex = scaled_split64(e, x);
  dense64T_16(d, x, L16, gA, m2T);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(d, hv, Pdp, 0, vr, pN, ex);       // d_pre1
  load_acts(hv, Px1, 0, vr, pN);                 // s1
  __builtin_amdgcn_sched_barrier(0);
  zero(e);
  ex = scaled_split64(d, x);                     // d_pre1 feeds both fc1^T parts
  dense64T_16(e, x, L16, gA, m1sT);
  __builtin_amdgcn_sched_barrier(0);
  tanh_adjoint(e, hv, Pdp, 3 * kW, vr, pN, ex);  // d_pre_s
  // conv outputs (the network inputs carry no gradient in this mode)
#pragma unroll
  for (int eb = 0; eb < 5; ++eb) {
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) y = mma3(L16.A(gA, m1cT + eb * 4 + kb), x[kb], y);
    const unsigned mws = hi ? mw[eb] >> 4 : mw[eb];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      Pdc.st(vr, (eb * 32 + rrow(i)) * pN,
             ((mws >> rrow(i)) & 1u) ? __builtin_amdgcn_ldexpf(y[i], ex) : 0.f);
  }
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

template <bool ROWS>
__global__ __launch_bounds__(kThreads) void mlp_concurrent_bwd_tm_kernel(WgArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b0 = blockIdx.x * kTrajPerBlock;
  const int b = b0 + wave * 32 + row;
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pact(A.acts, kActPlanes, pN), Pdz(A.d_zout, kNA, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;
  // trajectory-major addressing: lane = plane `row` of a 32-plane block, its
  // 16 trajectories start 4 hi into the wave's 32
  const unsigned wcol = (unsigned)(b0 + wave * 32) * 4u;
  const unsigned vt = (unsigned)row * pN + (unsigned)hi * 16u;
  const int nw = B - (b0 + wave * 32);      // live trajectories of this wave (may be <= 0)
  float *part = A.part + (size_t)blockIdx.x * kSlotsTm * 1024;
  char *lane_blk = lds + lane * 4;           // + region + block * 4096 + i * 256
  TmMeta &meta = *reinterpret_cast<TmMeta *>(lds + tMeta);
  bool bad = false;         // (workgroup-uniform) a non-finite operand was seen
  // The column 1-norms of W_1 (a bound on |W_1^T delta| per unit of max |delta|:
  // the scales of the states_in / conv cotangents, which are produced inside
  // the fc1 phase) - from the packed tables, first thing (nothing else is live
  // yet): wave w takes row block w of W_1^T (five blocks of the conv part, two
  // of the state part), a lane its row's 2 x 32 entries; the maxima are read
  // behind the layer barriers.  (Round 4: a block of the pack launch computed
  // them from the weights.  The tables' fp16 pairs carry the weights to 2^-22:
  // the same exponents.)
  if (wave < 7) {
    const int n0 = wave < 5 ? wC + 4 * wave : wS + 4 * (wave - 5);
    const u32x4 *tb = reinterpret_cast<const u32x4 *>(A.tables) + n0 * (kBlock16 / 16) + lane;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const u32x4 th = tb[kb * (kBlock16 / 16)], tl = tb[kb * (kBlock16 / 16) + 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const h16x2 h_ = __builtin_bit_cast(h16x2, th[q]), l_ = __builtin_bit_cast(h16x2, tl[q]);
        sum += fabsf((float)h_[0] + (float)l_[0]) + fabsf((float)h_[1] + (float)l_[1]);
      }
    }
    sum += other_half(sum);
    sum = wave_fmax(sum);
    if (lane == 0) meta.wnorm[wave] = sum;
  }

  // ---- this wave's inputs: dL/dz feature-major (20 rows per half-wave) and
  // trajectory-major (rows 0..31 and 32..39), h3's first block; the maxima of
  // the unbounded x plane groups (this wave's 32 trajectories)
  float dzr[20];
#pragma unroll
  for (int cc = 0; cc < 20; ++cc) dzr[cc] = Pdz.ld(vr, khead(cc, 0) * pN);
  TBlock tz[2], tx;
  tz[0].load(Pdz, vt, wcol);
  tz[1].load(Pdz, row < kNA - 32 ? vt : kDead, 32u * pN + wcol);
  tx.load(Pact, vt, (unsigned)pH3 * pN + wcol);
  {
    zero_region(lds, tRA, tHeadEx - tRA);
    zero_region(lds, tConvLo, kNC * 32 * 4);
    fill_lds_issue(lds_f, A.tables, kWgTabFloats);
    unsigned amax = 0u;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amax = umax_abs(amax, dzr[cc]);
    amax = wave_umax(amax);
    if (lane == 0) meta.dmax[0][wave] = amax;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the table DMA, tz
    // The head's 40 rows are the actions of ten different steps: their
    // cotangents differ by orders of magnitude, and ONE unit for the block
    // leaves the small rows with a few bits (round 5: 27 % of a row's own scale
    // with a x1e3 outlier in the workgroup, tests/test_gpu_round5.py).  Every
    // row gets its own exponent: the trajectory-major blocks have the row in
    // the lane, so a row's largest entry of this wave is lane-local; the
    // waves' biased exponents are exchanged as bytes behind this barrier.
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float v[16];
      tz[mb].get(v);
      unsigned m = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = rrow(i) + 4 * hi < nw ? umax_abs(m, v[i]) : m;
      const unsigned o = (unsigned)__shfl_xor((int)m, 32, 64);
      m = o > m ? o : m;
      if (hi == 0 && 32 * mb + row < kNA)
        reinterpret_cast<unsigned char *>(lds + tHeadEx)[wave * kNA + 32 * mb + row] =
            (unsigned char)(m >> 23);
    }
    __syncthreads();
  }
  // 2^ns, 2^nc: above the largest column 1-norm of W_1's state / conv part
  // (a non-finite norm: 0 - the gradients are non-finite anyway); scalars
  const auto norm_exp = [](float m) {
    return __builtin_amdgcn_readfirstlane(
        m > 0.f && m < 3.0e38f ? __builtin_amdgcn_frexp_expf(m) : 0);
  };
  int nc = norm_exp(fmaxf(fmaxf(fmaxf(meta.wnorm[0], meta.wnorm[1]),
                                fmaxf(meta.wnorm[2], meta.wnorm[3])), meta.wnorm[4]));
  int ns = norm_exp(fmaxf(meta.wnorm[5], meta.wnorm[6]));
  asm volatile("" : "+s"(nc), "+s"(ns));    // (computed HERE, kept in scalar registers)
  // exponent of this lane's head rows (32 mb + row): 2^e above the row's largest
  // cotangent of the workgroup (biased exponent field E: value < 2^(E - 126))
  int erow[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int r_ = 32 * mb + row < kNA ? 32 * mb + row : 0;
    unsigned e_ = 0u;
#pragma unroll
    for (int w8 = 0; w8 < kThreads / 64; ++w8) {
      const unsigned o = reinterpret_cast<const unsigned char *>(lds + tHeadEx)[w8 * kNA + r_];
      e_ = o > e_ ? o : e_;
    }
    erow[mb] = e_ ? (int)e_ - 126 : 0;
    if (wave == 0 && hi == 0 && 32 * mb + row < kNA)
      reinterpret_cast<int *>(lds + tHeadRow)[32 * mb + row] = erow[mb];
  }
  // The scales of the unbounded x plane groups (conv outputs, features + the
  // ones row, in_ref): the workgroup's maxima, which the forward kernel left per
  // wave (reading the planes for them here cost 9-16 us, wherever it was put)
  unsigned mc = 0u, mf = 0x3f800000u /* the ones row */, mi = 0u;
  {
    const unsigned *q = reinterpret_cast<const unsigned *>(A.xmax) +
                        (size_t)blockIdx.x * (kThreads / 64) * 4;
#pragma unroll
    for (int w8 = 0; w8 < kThreads / 64; ++w8) {
      mc = q[4 * w8] > mc ? q[4 * w8] : mc;
      mf = q[4 * w8 + 1] > mf ? q[4 * w8 + 1] : mf;
      mi = q[4 * w8 + 2] > mi ? q[4 * w8 + 2] : mi;
    }
  }
  const int fc = bits_exp(mc, bad, true), ff = bits_exp(mf, bad, true),
            fi = bits_exp(mi, bad, true);
  const LdsView16 L16(lds, lane);
  float hv[2][16];
  auto load_hv = [&](int plane) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) hv[rb][i] = Pact.ld(vr, (plane + rb * 32 + rrow(i)) * pN);
  };
  // this wave's largest next-layer cotangent -> its slot (read behind the barrier)
  auto post = [&](const f32x16 (&v)[2], int phase) {
    unsigned am = 0u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) am = umax_abs(am, v[rb][i]);
    am = wave_umax(am);
    if (lane == 0) meta.dmax[phase][wave] = am;
  };
  // bias gradient of 32 rows: sums over the lane's 16 trajectories, both halves -
  // a float per wave and row, straight into the partial buffer (two slots of
  // [4 waves][4 layers][64]); the second stage sums the eight waves in order.
  // (Until round 4 a fixed-point LDS accumulator with the layer's unit: under a
  // x1e3 outlier in the workgroup the biases were 12-40 x noisier than the
  // plane path, tests/test_gpu_round5.py.)
  float *bias_part = part + (size_t)(uBias + (wave >> 2)) * 1024 + (wave & 3) * 256;
  auto add_bias = [&](const float (&v)[16], int layer, int mb, int rows) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    s += other_half(s);
    if (hi == 0 && row < rows)
      __builtin_nontemporal_store(bad ? __builtin_nanf("") : s,
                                  bias_part + layer * 64 + 32 * mb + row);
  };

  // ------------------------------------------------------------- head
  f32x16 d[2], e[2];
  const int e0 = wg_exp(meta.dmax[0], bad);
  (void)e0;   // (only its `bad` flag: the head's rows have their own exponents)
  {
    // The chain's operands are scaled PER TRAJECTORY (round 5; until round 4 by
    // the workgroup's exponent: a trajectory whose cotangents are 1e-4 of the
    // workgroup's largest then kept 2^-22 x 1e4 of relative accuracy - its
    // weight terms and bias sums were as noisy as that, tests/test_gpu_round5.py);
    // the transposed operands' rows - trajectories - come back with their own
    // scales, the exponents are brought into accumulator layout by texp.
    float amx = 0.f;
#pragma unroll
    for (int cc = 0; cc < 20; ++cc) amx = fmaxf(amx, fabsf(dzr[cc]));
    const int ex0 = scale_exponent(amx);
    Op16 x0[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = kb * 8 + j < 20 ? __builtin_amdgcn_ldexpf(dzr[kb * 8 + j < 20 ? kb * 8 + j : 0], -ex0)
                               : 0.f;
      x0[kb] = split8(v);
    }
    Op16 az[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      float v[16];
      tz[mb].get(v);
#pragma unroll
      for (int i = 0; i < 16; ++i)   // columns beyond B are somebody else's plane
        v[i] = rrow(i) + 4 * hi < nw ? v[i] : 0.f;
      add_bias(v, 0, mb, mb ? kNA - 32 : 32);
      split16(v, erow[mb] - kPreD, az[mb]);   // (per lane: the row's own exponent)
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float xv[16];
      tx.get(xv);
      if (nb == 0) tx.load(Pact, vt, (unsigned)(pH3 + 32) * pN + wcol);
      else tx.load(Pact, vt, (unsigned)pH2 * pN + wcol);     // fc3's first x block

      Op16 bx[2];
      split16(xv, -kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(az[mb][kk], bx[kk], acc);
        add_block(lane_blk + tRA + (2 * nb + mb) * 4096, acc);
      }
      if (nb == 0) load_hv(pH3);
    }
    zero(d);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      d[0] = mma3(L16.A(0, wO + kb), x0[kb], d[0]);
      d[1] = mma3(L16.A(0, wO + 3 + kb), x0[kb], d[1]);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        d[rb][i] = __builtin_amdgcn_ldexpf(d[rb][i], ex0) * (1.f - hv[rb][i] * hv[rb][i]);
    post(d, 1);
  }
  __syncthreads();
  {  // the head's four blocks [2 nb + mb]: row r(i) + 4 hi + 32 mb has its own unit
    const i32x4_ z = {0, 0, 0, 0};
    for (int idx = threadIdx.x; idx < 4 * 256; idx += kThreads) {
      i32x4_ *p = reinterpret_cast<i32x4_ *>(lds + tRA) + idx;
      const i32x4_ q = *p;
      const int el = 4 * idx, blk = el >> 10, i = (el >> 6) & 15, ln = el & 63;
      const int r_ = 32 * (blk & 1) + rrow(i) + 4 * (ln >> 5);
      const int e_ = reinterpret_cast<const int *>(lds + tHeadRow)[r_ < kNA ? r_ : 0];
      f32x4_ v;
#pragma unroll
      for (int c_ = 0; c_ < 4; ++c_)
        v[c_] = bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)q[c_], e_ - kFix);
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4_ *>(part + sOut * 1024) + idx);
      *p = z;
    }
  }

  // The A operands of a layer's weight blocks from its cotangent in the chain's
  // orientation (round 6, as in mlp_rollout_bwd_tm_kernel; until round 5 the chain
  // ran a second time with swapped operands for them:
  // profiles/r06_transposition_probe.jsonl): the chain's own split x[kb] times an
  // identity B operand = trajectory r(i) + 4 hi of feature `lane & 31` in register
  // i (4 matrix instructions per 32 features, exact), the trajectories' exponents
  // in the same layout (texp), one ldexp per value to the workgroup's unit.  The
  // bias gradient: the wave's float sum per row, stored as before.
  u32x4 ident[2];
  ident_operands(lane, ident);
  auto transposed_operands = [&](const Op16 (&x)[4], int ex, int e_, int bias_id,
                                 Op16 (&ad)[2][2]) {
    int E[16];
    texp(ex, hi, E);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const Op16 pr[2] = {x[2 * mb], x[2 * mb + 1]};
      const f32x16 tz = to_feature_major(pr, ident);
      float v[16], sb = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = __builtin_amdgcn_ldexpf(tz[i], E[i] - e_ + kPreD);
        sb += v[i];
      }
      sb += other_half(sb);
      if (hi == 0)
        __builtin_nontemporal_store(
            bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf(sb, e_ - kPreD),
            bias_part + bias_id * 64 + 32 * mb + row);
      split16(v, 0, ad[mb]);
    }
  };
  // One 64 x 64 layer: dl = its cotangent (accumulator layout), e_ = the
  // workgroup's exponent for it.  Weight blocks against the two x blocks (the
  // second one and `next_plane`'s first are requested on the way), the
  // cotangent of the layer below (tables `tab`), tanh' with the planes
  // `x_plane`; its maxima go to slot `phase + 1`.
  auto layer64 = [&](f32x16 (&dl)[2], f32x16 (&nx)[2], int e_, int tab, int x_plane,
                     int region, int bias_id, int next_plane, int phase) {
    Op16 x[4];
    const int ex = scaled_split64(dl, x);     // per trajectory
    Op16 ad[2][2];
    transposed_operands(x, ex, e_, bias_id, ad);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float xv[16];
      tx.get(xv);
      tx.load(Pact, vt, (unsigned)(nb == 0 ? x_plane + 32 : next_plane) * pN + wcol);
      Op16 bx[2];
      split16(xv, -kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        add_block(lane_blk + region + (2 * nb + mb) * 4096, acc);
      }
      // (the tanh' operands of the feature-major chain below: requested here so
      // that they land under the second block's products)
      if (nb == 0) load_hv(x_plane);
    }
    zero(nx);
    dense64T_16(nx, x, L16, 0, tab);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        nx[rb][i] = __builtin_amdgcn_ldexpf(nx[rb][i], ex) * (1.f - hv[rb][i] * hv[rb][i]);
    post(nx, phase + 1);
  };
  // ---- fc3: x = h2 -> cotangent of h2's pre-activations
  const int e3 = wg_exp(meta.dmax[1], bad);
  layer64(d, e, e3, w3, pH2, tRB, 1, pH1, 1);
  __syncthreads();
  flush_region(lds, tRB, 4 * 1024, part + sFc3 * 1024, e3, true, bad);
  zero_region(lds, tF1a, tRA - tF1a);        // w3 and head tables: fc1's first blocks
  // ---- fc2: x = h1
  const int e2 = wg_exp(meta.dmax[2], bad);
  layer64(e, d, e2, w2, pH1, tRA, 2, pX1, 2);
  __syncthreads();
  flush_region(lds, tRA, 4 * 1024, part + sFc2 * 1024, e2, false, bad);

  // ---- fc1 (x = the 224 x1 planes: s1 | relu(conv)), states_in and conv
  const int e1 = wg_exp(meta.dmax[3], bad);
  const int es = e1 + ns, ec = e1 + nc;      // bounds of |d_pre_s|, |d conv|
  {
    Op16 x1s[4];
    const int ex1 = scaled_split64(d, x1s);   // per trajectory
    Op16 ad[2][2];
    transposed_operands(x1s, ex1, e1, 3, ad);
    // B operands that stay: the 15 feature planes + a row of ones (states_in's
    // bias column), the 90 in_ref planes in three blocks (conv windows)
    Op16 bfeat[2], binr[3][2];
    if (ROWS) {
      // x^T straight from the data set: lane = column `row` of the block, its 16
      // trajectories c + 8 g + 4 hi are 16 rows named by the index - one dword
      // load each, a half-wave on 32 consecutive floats of ONE row.  The wave's
      // 32 row numbers: one per lane, handed around by v_readlane.
      const int bw = b0 + wave * 32 + row;
      const unsigned r_ = (unsigned)A.index[bw < B ? bw : B - 1];
      const unsigned rf = r_ * (unsigned)A.ld_feat, ri = r_ * (unsigned)A.ld_in_ref;
      const auto sf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.r_feat), 0,
                                                        (int)A.bytes_feat, 0x00020000);
      const auto si = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.r_in_ref), 0,
                                                        (int)A.bytes_in_ref, 0x00020000);
      float v[16];
      const auto rows_block = [&](__amdgpu_buffer_rsrc_t rs, unsigned rbase, int col, bool on) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned a0 = (unsigned)__builtin_amdgcn_readlane((int)rbase, c + 8 * g),
                           a1 = (unsigned)__builtin_amdgcn_readlane((int)rbase, c + 8 * g + 4);
            const unsigned off = on ? ((hi ? a1 : a0) + (unsigned)col) * 4u : kDead;
            v[4 * g + c] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, APG_PLANES_LD_AUX));
          }
      };
      rows_block(sf, rf, row, row < kNF);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = row == kNF ? 1.f : v[i];
      split16(v, ff - kPreX, bfeat);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        rows_block(si, ri, 32 * jb + row, 32 * jb + row < kH * kRD);
        split16(v, fi - kPreXc, binr[jb]);
      }
    } else {
      TBlock tf;
      tf.load(Pact, row < kNF ? vt : kDead, (unsigned)pFeat * pN + wcol);
      float v[16];
      tf.get(v);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = row == kNF ? 1.f : v[i];
      split16(v, ff - kPreX, bfeat);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        tf.load(Pact, 32 * jb + row < kH * kRD ? vt : kDead,
                (unsigned)(pInr + 32 * jb) * pN + wcol);
        tf.get(v);
        split16(v, fi - kPreXc, binr[jb]);
      }
    }
    // fc1's weight blocks 2 nb, 2 nb + 1 against x block nb (scaled by 2^-fx)
    auto fc1_blocks = [&](const float (&xv)[16], int fx, int nb) {
      Op16 bx[2];
      split16(xv, fx - kPreX, bx);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        const int blk = 2 * nb + mb;     // blocks 0..6 in the first piece
        add_block(lane_blk + (blk < 7 ? tF1a + blk * 4096 : tF1b + (blk - 7) * 4096), acc);
      }
    };
    // the transposed product of d_pre1 against four table blocks from `blk0`
    auto transposed = [&](int blk0) {
      const char *tb = L16.b0 + blk0 * kBlock16;   // (all below 60 KB)
      f32x16 t;
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        Op16 w;
        w.h = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16);
        w.l = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16 + 1024);
        t = mma3(x1s[kb], w, t);
      }
      return t;
    };
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      // s1's blocks: also states_in - cotangent of its pre-activations, block
      // nb, and the weight block against the features
      float xv[16];
      tx.get(xv);
      tx.load(Pact, vt, (unsigned)(pX1 + 32 * (nb + 1)) * pN + wcol);
      fc1_blocks(xv, 0, nb);
      const f32x16 t = transposed(wS + 4 * nb);
      int E1[16];
      texp(ex1, hi, E1);
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        v[i] = __builtin_amdgcn_ldexpf(t[i], E1[i]) * (1.f - xv[i] * xv[i]);
      Op16 as[2];
      split16(v, es - kPreD, as);
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc = mma3(as[kk], bfeat[kk], acc);
      // 16 columns are real (15 features + the ones row): compact [reg][half][16],
      // 2 KB of high limbs per block, the low limbs 4 KB further
      if (row < 16) {
        char *q = lds + tSin + nb * 2048 + (hi * 16 + row) * 4;
#pragma unroll
        for (int i = 0; i < 16; ++i) lds_add2(q + i * 128, q + 4096 + i * 128, acc[i], kFix);
      }
    }
#pragma unroll 1
    for (int eb = 0; eb < 5; ++eb) {
      // conv blocks: x block 2 + eb = the saved conv outputs e = 32 eb + row
      // (channel 4 eb + row / 8, position row % 8); their cotangent with relu'
      // from the saved outputs, then its products against the in_ref planes
      float xv[16];
      tx.get(xv);
      if (eb < 4) tx.load(Pact, vt, (unsigned)(pX1 + 32 * (eb + 3)) * pN + wcol);
      fc1_blocks(xv, fc, eb + 2);
      const f32x16 t = transposed(wC + 4 * eb);
      int E1[16];
      texp(ex1, hi, E1);
      float v[16], sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = xv[i] > 0.f ? __builtin_amdgcn_ldexpf(t[i], E1[i] - ec) : 0.f;   // / 2^ec
        sum += v[i];
      }
      // the conv block's rows of channels 4 eb .. 4 eb + 3 (accumulator layout:
      // channel ch = register (ch & 3) + 4 (ch >> 3) of half-wave (ch >> 2) & 1)
      char *cblk = lds + tConv + ((4 * (eb >> 1)) * 64 + 32 * (eb & 1)) * 4;
      char *clo = lds + tConvLo + 4 * eb * 32 * 4;        // low limbs: [channel][32]
      sum += other_half(sum);
      // bias: column 27; the block's unit carries in_ref's scale 2^fi as well
      if (hi == 0)
        lds_add2(cblk + ((row >> 3) * 64 + 27) * 4, clo + ((row >> 3) * 32 + 27) * 4,
                 __builtin_amdgcn_ldexpf(sum, kFixConv - fi));
      Op16 ac[2];
      split16(v, -kPreDc, ac);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ac[kk], binr[jb][kk], acc);
        // register 4 g + c of lane (hi, col): conv output row c + 8 g + 4 hi of the
        // block = channel 4 eb + g at position c + 4 hi, against in_ref plane
        // j = 32 jb + col: tap q = j - 9 position of that channel's 27
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int q = 32 * jb + row - kRD * (c + 4 * hi);
          if (q >= 0 && q < 27) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              lds_add2(cblk + (g * 64 + q) * 4, clo + (g * 32 + q) * 4, acc[4 * g + c]);
          }
        }
      }
    }
  }
  __syncthreads();
  // fc1: blocks 0..3 against s1 (unit 2^e1), 4..13 against the conv outputs (2^(e1 + fc))
  flush_region(lds, tF1a, 4 * 1024, part + sFc1 * 1024, e1, false, bad);
  flush_region(lds, tF1a + 4 * 4096, 3 * 1024, part + (sFc1 + 4) * 1024, e1 + fc, false, bad);
  flush_region(lds, tF1b, 7 * 1024, part + (sFc1 + 7) * 1024, e1 + fc, false, bad);
  for (int idx = threadIdx.x; idx < 2 * 512; idx += kThreads) {   // states_in: both limbs
    const int nb = idx >> 9, r_ = idx & 511, at = (r_ >> 5) * 64 + 32 * ((r_ >> 4) & 1) + (r_ & 15);
    const int *q = reinterpret_cast<const int *>(lds + tSin) + nb * 512 + r_;
    const double v = (double)q[0] + (double)q[1024] * (1.0 / (double)(1 << kFix));
    part[(sSin + nb) * 1024 + at] =
        bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)v, es + ff - kFix);
  }
  // conv: both limbs of the 20 x 28 used elements (accumulator layout in the slot)
  for (int idx = threadIdx.x; idx < kNC * 32; idx += kThreads) {
    const int ch = idx >> 5, q = idx & 31;
    const int at = ((ch & 3) + 4 * (ch >> 3)) * 64 + 32 * ((ch >> 2) & 1) + q;
    const double v = (double)reinterpret_cast<const int *>(lds + tConv)[at] +
                     (double)reinterpret_cast<const int *>(lds + tConvLo)[idx] *
                         (1.0 / (double)(1 << kFixConv));
    part[uConv * 1024 + at] =
        bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)v, ec + fi - kFixConv);
  }
}

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

// ---------------------------------------------------------------------------
// Round 5: the AUTOREGRESSIVE reverse sweep with every weight gradient inside
// (TrainDrone.train_recurrent_model's loss.backward(), scripts/train_drone.py:
// 113-173, in ONE launch: no cotangent planes, no product launches).
//
// Until round 4 the reverse sweep wrote 256 cotangent planes of H B floats and
// 720 conv diagonals for nine planes_gemm launches that read them and the 431
// activation planes again (474 us of the 1 111 us step at B = 65 536).  Here
// the products happen where the cotangents are, in the trajectory-major form
// of mlp_concurrent_bwd_tm_kernel: per step and layer every wave multiplies
// ITS 32 trajectories' cotangent (swapped-operand product: trajectory in the
// registers, feature in the lane) against its own x (four 16-byte loads per
// lane from the forward sweep's planes) and adds the 32 x 32 blocks into the
// workgroup's 32-bit fixed-point accumulators in LDS (ds_add_u32: order-free,
// bit-reproducible).  What the recurrence adds to the concurrent form:
//  * the operand tables (50 blocks, 102 KB) stay live for all H steps, so only
//    56 KB of LDS are left for accumulators: a step is FIVE phases - head +
//    fc3 | fc2 | fc1 (s1 columns) + states_in | fc1 (conv columns 0..95) |
//    fc1 (conv columns 96..159), the conv block collecting over the last two -
//    whose blocks alternate between two 24 KB regions; behind each phase's
//    barrier the region is FLUSHED into the workgroup's own partial buffer in
//    global memory (fixed point -> float x the step's scale, one
//    global_atomic_add_f32 per element, no return value; the first step
//    stores) while the next phase adds into the other region.  The partial
//    buffer is 120 KB per workgroup, L2 / Infinity-Cache resident: 10 x 31 MB
//    of read-modify-write at the caches against 2.9 GB of HBM planes gone.
//    One thread owns an element for the whole sweep and the steps add in
//    order: the sums are deterministic.
//  * the cotangent scale changes from step to step, so every phase has its own
//    workgroup exponent (the waves' maxima are exchanged through LDS behind
//    the barrier that is there anyway; a step's first barrier sits behind the
//    NEXT step's dynamics adjoint and head, which produce the first maxima)
//    and the flush applies it: the global accumulators are plain floats.
//  * the feature-major chain feeds the dynamics adjoint, so - unlike in the
//    concurrent kernel - it keeps the PER-TRAJECTORY power-of-two scaling of
//    scaled_split64; the swapped products take the same operands, their rows
//    (trajectories) therefore arrive with different scales, and the exponents
//    are brought into the accumulator layout by one more matrix instruction
//    (texp: D[trajectory][feature] = ex[trajectory]).
//  * the windows of the conv product are relative to the drone's position of
//    the step: the in_ref blocks are loaded trajectory-major per step and the
//    position planes subtracted from their columns 0..2 before the split.
// Tables (bytes from gA; blocks of 2 KB in cbwd_weight's order): fc1^T conv
// part [eb][kb] first (addressed with a run-time block index: below 60 KB),
// states_in^T, the two head^T blocks (4 real k-slots), fc3^T, fc2^T, fc1^T
// state part.
constexpr int a1c = 0, aS = 20, aH = 24, a3 = 26, a2 = 34, a1s = 42, aBlocks = 50;
constexpr int kArTabBytes = gA + aBlocks * kBlock16;   // 104 448
constexpr int kArTabFloats = kArTabBytes / 4;
// LDS behind the tables: two alternating accumulator regions of six blocks, the
// conv block, the head block [4][64], meta
constexpr int kArRegion = 6 * 4096;
constexpr int rX = kArTabBytes, rY = rX + kArRegion, rConv = rY + kArRegion,
              rHead = rConv + 4096, rMeta = rHead + 1024;
static_assert((rMeta - rX) % 16 == 0, "zeroed in 16-byte pieces");

__device__ __forceinline__ float car_weight(const ApgMlpPolicy &p, int n, int row, int j,
                                            int hi) {
  const int old = n < aS ? m1cT + (n - a1c) : n < aH ? mST + (n - aS)
                  : n < a3 ? mOT + 3 * (n - aH) : n < a2 ? m3T + (n - a3)
                  : n < a1s ? m2T + (n - a2) : m1sT + (n - a1s);
  return cbwd_weight(p, old, row, j, hi, 4);
}

// forward tables at dst, the reverse tables of mlp_rollout_bwd_tm_kernel at
// dst + kCfLds, behind them ns, nc (see mlp_pack_step_kernel; the LAST block)
__global__ __launch_bounds__(256) void mlp_pack_ar_kernel(PackArgs A, int fwd_blocks) {
  if (blockIdx.x + 1 == gridDim.x) {
    __shared__ float wmax[4];
    const int t = threadIdx.x;
    float sum = 0.f;
    if (t < kN1) {
#pragma unroll
      for (int k = 0; k < kW; ++k) sum += fabsf(A.pol.w_1[k * kN1 + t]);
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) sum = fmaxf(sum, __shfl_xor(sum, sft, 64));
    if ((t & 63) == 0) wmax[t >> 6] = sum;
    __syncthreads();
    if (t < 2) {
      const float m = t ? fmaxf(fmaxf(wmax[1], wmax[2]), wmax[3]) : wmax[0];
      A.dst[kCfLds + kArTabFloats + t] =
          m > 0.f && m < 3.0e38f ? (float)__builtin_amdgcn_frexp_expf(m) : 0.f;
    }
    return;
  }
  if ((int)blockIdx.x < fwd_blocks) {
    pack_cfwd(A, blockIdx.x * blockDim.x + threadIdx.x, fwd_blocks * blockDim.x);
    return;
  }
  const int tid = (blockIdx.x - fwd_blocks) * blockDim.x + threadIdx.x;
  const int T = (gridDim.x - 1 - fwd_blocks) * blockDim.x;
  float *dstf = A.dst + kCfLds;
  unsigned *dst = reinterpret_cast<unsigned *>(dstf);
  for (int idx = tid; idx < aBlocks * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    const float w0 = car_weight(A.pol, n, l & 31, 2 * q, l >> 5);
    const float w1 = car_weight(A.pol, n, l & 31, 2 * q + 1, l >> 5);
    unsigned h, lo;
    split_pair(w0, w1, h, lo);
    dst[(gA + n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(gA + n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
  const ApgMlpPolicy &p = A.pol;
  for (int idx = tid; idx < 256; idx += T) {   // the four head rows, VALU order
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = (idx >> 5) & 1, j = idx >> 6;
    dstf[gTo + idx] = p.w_out[j * kW + rb * 32 + rrow(i) + 4 * hi];
  }
  for (int idx = tid; idx < kNC * 3; idx += T) {
    const int ch = idx / 3, q = idx % 3;
    dstf[gAq + idx] = p.conv_w[ch * 27 + q * 3] + p.conv_w[ch * 27 + q * 3 + 1] +
                      p.conv_w[ch * 27 + q * 3 + 2];
  }
}

struct ArTmArgs {
  const float *state0, *states, *actions, *ref, *in_ref;
  const float *feat, *x1, *h;    // [15][N], [224][N], [192][N] (the forward sweep's)
  const unsigned *mask;          // [5][N]
  float *loss_partials;
  float *part;                   // [workgroups][kSlotsTm][1024]
  float *grad_state0;
  const float *tables;
  const float *xmax;             // [waves][H][4] (the forward sweep's)
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
};

// (knock-out builds: keep a block product alive without the LDS additions)
__device__ __forceinline__ void ar_sink(const f32x16 &acc) {
  float s_ = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s_ += acc[i];
  asm volatile("" ::"v"(s_));
}
__device__ __forceinline__ void ar_barrier() {
  if (APG_AR_KNOCKOUT & 16) return;
  __syncthreads();
}

struct ArMeta {           // at rMeta; written by plain stores, one slot per wave
  unsigned dmax[5][8];    // max |cotangent| bits of head, fc3, fc2, fc1, conv
};
static_assert(rMeta + (int)sizeof(ArMeta) <= kLdsAll, "LDS map");

__global__ __launch_bounds__(kThreads) void mlp_rollout_bwd_tm_kernel(ArTmArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  char *lds = reinterpret_cast<char *>(lds_f);
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b0 = blockIdx.x * kTrajPerBlock;
  const int b = b0 + wave * 32 + row;
  const int B = A.B;
  const bool live = b < B, st_lo = live && hi == 0;
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pst(A.states, kH * 12, pitchB);
  const Planes Pac(A.actions, kH * 4, pitchB), Prf(A.ref, kH * A.ref_cols, pitchB);
  const Planes Pin(A.in_ref, 2 * kH * kRD, pitchB);
  const Planes Pfe(A.feat, kNF, pitchN), Px1(A.x1, kN1, pitchN), Ph(A.h, 3 * kW, pitchN);
  const Planes Pmk(A.mask, 5, pitchN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  // trajectory-major addressing: lane = plane `row` of a 32-plane block, its 16
  // trajectories start 4 hi into the wave's 32
  const unsigned wcolB = (unsigned)(b0 + wave * 32) * 4u;
  const unsigned vtN = (unsigned)row * pitchN + (unsigned)hi * 16u;
  const unsigned vtB = (unsigned)row * pitchB + (unsigned)hi * 16u;
  // this workgroup's accumulators in global memory ([kSlotsTm][1024] floats)
  const __amdgpu_buffer_rsrc_t part = __builtin_amdgcn_make_buffer_rsrc(
      A.part + (size_t)blockIdx.x * kSlotsTm * 1024, 0, kSlotsTm * 4096, 0x00020000);
  char *lane_blk = lds + lane * 4;
  ArMeta &meta = *reinterpret_cast<ArMeta *>(lds + rMeta);
  const int ns = (int)A.tables[kArTabFloats];
  bool bad = false;         // (workgroup-uniform) a non-finite operand was seen

  zero_region(lds, rX, rMeta - rX);
  {  // the global accumulators start at zero: every flush is an atomic add
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < kSlotsTm * 256; m += kThreads)
      __builtin_amdgcn_raw_buffer_store_b128(z, part, (int)(threadIdx.x * 16u), m * 16, 0);
  }
  fill_lds_issue(lds_f, A.tables, kArTabFloats);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const LdsView16 L16(lds, lane);
  const LdsView L(lds_f, lane);
  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float loss = 0.f;
  int rg = rX, ro = rY;     // the region the current phase adds into / the other one
  int e5 = 0, ecv = 0;      // scales of the blocks whose flush is deferred to the next step
  TBlock tx, tx2;
  const auto post = [&](const f32x16 (&v)[2], int phase) {
    unsigned am = 0u;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) am = umax_abs(am, v[rb][i]);
    am = wave_umax(am);
    if (lane == 0) meta.dmax[phase][wave] = am;
  };
  // Bias gradients: a float per WAVE and row (sum over the wave's 32
  // trajectories), added straight into this wave's own entries of the two bias
  // slots ([4 waves][4 layers][64] each; layer 0 = fc_out's 4 rows, its entries
  // 32..51 the conv bias) - one writer per address, the steps in order; the
  // second stage sums the eight waves in order.  No fixed-point unit involved.
  const unsigned bias_soff = (unsigned)(uBias + (wave >> 2)) * 4096u + (unsigned)(wave & 3) * 1024u;
  // (transposed_operands below adds them)
  // the deferred blocks of a step: fc1's last four (conv columns 96..159) and
  // the conv block
  const auto flush_tail = [&]() {
    flush_add<4 * 1024>(lds, rg, part, (sFc1 + 10) * 4096, e5, bad);
    flush_add<1024>(lds, rConv, part, uConv * 4096, ecv, bad, kFixConv);
  };

#pragma unroll 1
  for (int k = kH - 1; k >= 0; --k) {
    // (knock-out 64: every per-lane plane sits 256 bytes from the next - cache resident)
    const unsigned pB = (APG_AR_KNOCKOUT & 64) ? 256u : opaque(pitchB), pN = opaque(pitchN);
    const unsigned col = (unsigned)b * 4u + (unsigned)k * pitchB;
    const unsigned vn = live ? ((APG_AR_KNOCKOUT & 64) ? (unsigned)lane * 4u : col) : kDead;
    const unsigned wcolN = wcolB + (unsigned)k * pB;   // (scalar) column k B + the wave's first
    // the identity operands of the transpositions: made per step from an opaque
    // lane index (eight registers that would otherwise live through the sweep)
    u32x4 ident[2];
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      ident_operands(lane_o, ident);
    }
    // ------------------------------------------------ dynamics adjoint, head
    float dz[4];
    unsigned m0 = 0u;
    f32x16 d[2], e[2];
    float x3[2][16];      // h3, trajectory-major (both blocks: the head's x, tanh' of fc3)
    {
      float sn[12], sc[12], a[4], rp[3], rv[3];
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        sn[i] = Pst.ld(vb, (k * 12 + i) * pB);
        sc[i] = k > 0 ? Pst.ld(vb, ((k - 1) * 12 + i) * pB) : Ps0.ld(vb, i * pB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = Pac.ld(vb, (k * 4 + j) * pB);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        rp[i] = Prf.ld(vb, (k * A.ref_cols + i) * pB);
        rv[i] = Prf.ld(vb, (k * A.ref_cols + A.vel_col + i) * pB);
      }
      tx.load(Ph, vtN, (unsigned)(2 * kW) * pN + wcolN);        // h3, block 0
      tx2.load(Ph, vtN, (unsigned)(2 * kW + 32) * pN + wcolN);  // h3, block 1
      __builtin_amdgcn_sched_barrier(0);
      float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float dp = sn[i] - rp[i], dv = sn[6 + i] - rv[i], wn = sn[9 + i];
        lp += dp * dp, lv += dv * dv, lw += wn * wn;
        lam[i] += 2.f * A.w.pos * dp;
        lam[6 + i] += 2.f * A.w.vel * dv;
        lam[9 + i] += 2.f * A.w.av * wn;
      }
      const float da0 = a[0] - 0.5f;
      float ga[4];
      ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        const float dd = a[j] - 0.5f;
        lr += dd * dd;
        ga[j] = 2.f * A.w.rates * dd;
      }
      loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
              A.w.thrust * da0 * da0;
      const Trig t = make_trig(&sc[3]);
      quad_step_adjoint(lam, ga, a[0], &sc[9], c, t);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dz[j] = ga[j] * a[j] * (1.f - a[j]);
        m0 = umax_abs(m0, dz[j]);
      }
    }
    tx.get(x3[0]);
    tx2.get(x3[1]);
    tx.load(Ph, vtN, (unsigned)kW * pN + wcolN);   // fc3's first x block (h2)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      // d_pre3 = (W_out^T dL/dz) tanh'(h3): the head on the VALU, h3 in
      // accumulator layout from its trajectory-major block
      Op16 bx[2];
      split16(x3[rb], -kPreX, bx);
      const f32x16 hf = to_feature_major(bx, ident);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v = fmaf(L.T(gTo + ((j * 2 + rb) * 16 + i) * 2), dz[j], v);
        const float h3 = __builtin_amdgcn_ldexpf(hf[i], -kPreX);
        d[rb][i] = v * (1.f - h3 * h3);
      }
    }
    m0 = wave_umax(m0);
    if (lane == 0) meta.dmax[0][wave] = m0;
    post(d, 1);
    // the scales of the unbounded x plane groups of this step (conv outputs,
    // features + the ones row, window values): the workgroup's maxima, left per
    // wave and step by the forward sweep
    unsigned mc = 0u, mf = 0x3f800000u /* the ones row */, mi = 0u;
    {
      const unsigned *q = reinterpret_cast<const unsigned *>(A.xmax) +
                          ((size_t)blockIdx.x * (kThreads / 64) * kH + k) * 4;
#pragma unroll
      for (int w8 = 0; w8 < kThreads / 64; ++w8) {
        mc = q[w8 * kH * 4] > mc ? q[w8 * kH * 4] : mc;
        mf = q[w8 * kH * 4 + 1] > mf ? q[w8 * kH * 4 + 1] : mf;
        mi = q[w8 * kH * 4 + 2] > mi ? q[w8 * kH * 4 + 2] : mi;
      }
    }
    const int fc = bits_exp(mc, bad, true), ff = bits_exp(mf, bad, true),
              fi = bits_exp(mi, bad, true);
    ar_barrier();
    // the previous step's last phase and its conv block (the first step: zeros)
    flush_tail();
    { const int r_ = rg; rg = ro, ro = r_; }

    // ------------------------------------------------ phase 1: head, fc3
    const int e0 = wg_exp(meta.dmax[0], bad), e3 = wg_exp(meta.dmax[1], bad);
    {
      // dL/dz^T by an identity product (scaled into accumulator units), W_out's
      // and b_out's gradient
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v8[j] = (hi == 0 && j < 4) ? __builtin_amdgcn_ldexpf(dz[j < 4 ? j : 0], kPreD - e0) : 0.f;
      const Op16 x0 = split8(v8);
      // k-slot 8 hi + j of column c is 1 where it IS c
      u32x4 idz;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        idz[q] = (8 * hi + 2 * q == row ? 0x3c00u : 0u) |
                 (8 * hi + 2 * q + 1 == row ? 0x3c000000u : 0u);
      f32x16 tzv;
#pragma unroll
      for (int i = 0; i < 16; ++i) tzv[i] = 0.f;
      tzv = mfma16(x0.l, idz, tzv);
      tzv = mfma16(x0.h, idz, tzv);
      float tz[16], sb = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        tz[i] = tzv[i];
        sb += tz[i];
      }
      sb += other_half(sb);
      if (hi == 0 && row < 4)
        gadd(part, (unsigned)row * 4u, bias_soff,
             bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf(sb, e0 - kPreD));
      Op16 az[2];
      split16(tz, 0, az);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        Op16 bx[2];
        split16(x3[nb], -kPreX, bx);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(az[kk], bx[kk], acc);
        if (hi == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) lds_add(lds + rHead + (i * 64 + 32 * nb + row) * 4, acc[i]);
        }
      }
    }
    // The A operands of a layer's weight blocks from its cotangent in the chain's
    // orientation (round 6; until round 5 the chain ran a second time with its
    // operands swapped to get them: tools/transposition_probe.hip prices both,
    // profiles/r06_transposition_probe.jsonl - 7.5 against 5.6 us per layer): the
    // split the chain needs anyway, x[kb] = d 2^-ex[trajectory] as fp16 pairs with
    // the trajectory in the lane, times an identity B operand puts trajectory
    // r(i) + 4 hi of feature `lane & 31` into register i (4 matrix instructions per
    // 32 features, exact: every product is a value times one); the trajectories'
    // exponents arrive in the same layout (texp) and the rescale to the
    // workgroup's unit 2^(e_ - kPreD) is one ldexp per value.  Bias gradient: the
    // float sum of the wave's 32 trajectories per row, as before.
    const auto transposed_operands = [&](const Op16 (&x)[4], int ex, int e_, int bias_id,
                                         Op16 (&ad)[2][2]) {
      int E[16];
      texp(ex, hi, E);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const Op16 pr[2] = {x[2 * mb], x[2 * mb + 1]};
        const f32x16 tz = to_feature_major(pr, ident);
        float v[16], sb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          v[i] = __builtin_amdgcn_ldexpf(tz[i], E[i] - e_ + kPreD);
          sb += v[i];
        }
        sb += other_half(sb);
        if (bias_id >= 0 && hi == 0)
          gadd(part, (unsigned)(bias_id * 64 + 32 * mb + row) * 4u, bias_soff,
               bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf(sb, e_ - kPreD));
        split16(v, 0, ad[mb]);
      }
    };
    // One 64 x 64 layer: dl = its cotangent (accumulator layout), e_ = the
    // workgroup's exponent for it, x = planes [x_plane, +64) of X (the second
    // block and `next_plane`'s first of Xn are requested on the way).  Weight
    // blocks into the region at `rb`, the cotangent of the layer below (tables
    // `tab`; tanh' from the x blocks, brought into accumulator layout by
    // to_feature_major), its maxima into slot `phase + 1`.
    const auto layer64 = [&](f32x16 (&dl)[2], f32x16 (&nx)[2], int e_, int tab, const Planes &X,
                             int x_plane, const Planes &Xn, int next_plane, int bias_id,
                             int phase, char *rb) {
      Op16 x[4];
      const int ex = scaled_split64(dl, x);
      Op16 ad[2][2];
      transposed_operands(x, ex, e_, bias_id, ad);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float xv[16];
        tx.get(xv);
        if (nb == 0) tx.load(X, vtN, (unsigned)(x_plane + 32) * pN + wcolN);
        else tx.load(Xn, vtN, (unsigned)next_plane * pN + wcolN);
        Op16 bx[2];
        split16(xv, -kPreX, bx);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          if (APG_AR_KNOCKOUT & 2) break;
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
          if (APG_AR_KNOCKOUT & 4) ar_sink(acc); else
          add_block(rb + (2 * nb + mb) * 4096, acc);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) nx[nb][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) nx[nb] = mma3(L16.A(gA, tab + 4 * nb + kb), x[kb], nx[nb]);
        const f32x16 hf = to_feature_major(bx, ident);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float hx = __builtin_amdgcn_ldexpf(hf[i], -kPreX);
          nx[nb][i] = __builtin_amdgcn_ldexpf(nx[nb][i], ex) * (1.f - hx * hx);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      post(nx, phase + 1);
    };
    layer64(d, e, e3, a3, Ph, kW, Ph, 0, 1, 1, lane_blk + rg);   // x = h2 -> d_pre2
    ar_barrier();
    if (threadIdx.x < 4 * 64) {   // W_out: [4][64] (waves 0..3)
      const int idx = threadIdx.x;
      int *p = reinterpret_cast<int *>(lds + rHead) + idx;
      const int q = *p;
      *p = 0;
      gadd(part, (unsigned)((idx >> 6) * 64 + (idx & 31)) * 4u,
           (sOut + 2 * ((idx >> 5) & 1)) * 4096,
           bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)q, e0 - kFix));
    }
    flush_add<4 * 1024>(lds, rg, part, sFc3 * 4096, e3, bad);
    { const int r_ = rg; rg = ro, ro = r_; }

    // ------------------------------------------------ phase 2: fc2 (x = h1)
    const int e2 = wg_exp(meta.dmax[2], bad);
    layer64(e, d, e2, a2, Ph, 0, Px1, 0, 2, 2, lane_blk + rg);   // -> d_pre1
    ar_barrier();
    flush_add<4 * 1024>(lds, rg, part, sFc2 * 4096, e2, bad);
    { const int r_ = rg; rg = ro, ro = r_; }

    // ------------- phase 3: fc1 against s1 (x1 planes 0..63), states_in; the
    // conv cotangent feature-major (position cotangent, its exact maximum)
    const int e1 = wg_exp(meta.dmax[3], bad);
    const int es = e1 + ns;                    // a bound of |d_pre_s|
    Op16 x1s[4];    // d_pre1, scaled per trajectory and split: all of fc1^T's parts
    const int ex1 = scaled_split64(d, x1s);
    Op16 ad[2][2];  // d_pre1^T with the workgroup's scale: all of fc1's weight blocks
    transposed_operands(x1s, ex1, e1, 3, ad);
    // fc1's weight blocks of x block `xv` (scaled by 2^-fx), both row blocks
    const auto fc1_blocks = [&](const Op16 (&bx)[2], char *blk) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if (APG_AR_KNOCKOUT & 2) break;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ad[mb][kk], bx[kk], acc);
        if (APG_AR_KNOCKOUT & 4) ar_sink(acc); else
        add_block(blk + mb * 4096, acc);
      }
    };
    Op16 bfeat[2];   // the 15 feature planes + a row of ones (states_in's bias column)
    {
      {
        TBlock tf;
        tf.load(Pfe, row < kNF ? vtN : kDead, wcolN);
        float v[16];
        get_clamped(tf, v, ff);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = row == kNF ? 1.f : v[i];
        split16(v, ff - kPreX, bfeat);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float xv[16];
        tx.get(xv);
        tx.load(Px1, vtN, (unsigned)(32 * (nb + 1)) * pN + wcolN);
        Op16 bx[2];
        split16(xv, -kPreX, bx);
        fc1_blocks(bx, lane_blk + rg + 2 * nb * 4096);
#pragma unroll
        for (int i = 0; i < 16; ++i) e[nb][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) e[nb] = mma3(L16.A(gA, a1s + 4 * nb + kb), x1s[kb], e[nb]);
        // d_pre_s (block nb of its rows)
        const f32x16 hf = to_feature_major(bx, ident);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float hx = __builtin_amdgcn_ldexpf(hf[i], -kPreX);
          e[nb][i] = __builtin_amdgcn_ldexpf(e[nb][i], ex1) * (1.f - hx * hx);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // feature-major: the feature cotangent, the state cotangent
      // (the pre-step state and its trigonometry again: 18 registers that would
      // otherwise live from the dynamics adjoint to here)
      float sc[12];
#pragma unroll
      for (int i = 0; i < 12; ++i)
        sc[i] = k > 0 ? Pst.ld(vb, ((k - 1) * 12 + i) * pB) : Ps0.ld(vb, i * pB);
      f32x16 f;
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = 0.f;
      Op16 xs[4];
      const int exs = scaled_split64(e, xs);
      {  // states_in's weight blocks: d_pre_s^T from the same split (no bias slot:
         // its bias is the ones row of the feature block), unit 2^(es - kPreD)
        Op16 as[2][2];
        transposed_operands(xs, exs, es, -1, as);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) acc = mma3(as[nb][kk], bfeat[kk], acc);
          // 16 columns are real (15 features + the ones row): compact [reg][half][16],
          // 2 KB of high limbs per block, the low limbs 4 KB further
          if (row < 16) {
            char *q = lds + rg + 4 * 4096 + nb * 2048 + (hi * 16 + row) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) lds_add2(q + i * 128, q + 4096 + i * 128, acc[i], kFix);
          }
        }
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) f = mma3(L16.A(gA, aS + kb), xs[kb], f);
      const Trig t = make_trig(&sc[3]);
      float dfeat[kNF], gs[12];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float own = __builtin_amdgcn_ldexpf(f[i], exs), oth = other_half(own);
        dfeat[rrow(i)] = hi ? oth : own;
        if (rrow(i) + 4 < kNF) dfeat[rrow(i) + 4 < kNF ? rrow(i) + 4 : 0] = hi ? own : oth;
      }
      quad_features_adjoint(sc, t, dfeat, gs);
#pragma unroll
      for (int i = 0; i < 12; ++i) lam[i] += gs[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    {  // feature-major conv cotangent: relu', the sum over the positions of a
       // channel -> the cotangent of the drone's position (window columns 0..2
       // are relative), and its largest entry -> the conv block's exact scale
      unsigned mw[5];
#pragma unroll
      for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vn, eb * pN);
      float dpos[3] = {0.f, 0.f, 0.f};
      unsigned cm = 0u;
#pragma unroll 1
      for (int eb = 0; eb < 5; ++eb) {
        const char *tb = L16.b0 + gA + (a1c + 4 * eb) * kBlock16;   // (below 60 KB)
        f32x16 y;
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          Op16 w;
          w.h = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16);
          w.l = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16 + 1024);
          y = mma3(w, x1s[kb], y);
        }
        const unsigned mwe = eb == 0 ? mw[0] : eb == 1 ? mw[1] : eb == 2 ? mw[2]
                             : eb == 3 ? mw[3] : mw[4];
        const unsigned mws = hi ? mwe >> 4 : mwe;  // bit r(i) + 4 hi -> bit r(i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3: channel eb*4 + g, positions ii + 4 hi
          float sg = 0.f;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * g + ii;
            const float yv = ((mws >> rrow(i)) & 1u) ? __builtin_amdgcn_ldexpf(y[i], ex1) : 0.f;
            cm = umax_abs(cm, yv);
            sg += yv;
          }
#pragma unroll
          for (int q = 0; q < 3; ++q)
            dpos[q] = fmaf(lds_f[L.o_0 + gAq + (eb * 4 + g) * 3 + q], sg, dpos[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) lam[q] -= dpos[q] + other_half(dpos[q]);
      cm = wave_umax(cm);
      if (lane == 0) meta.dmax[4][wave] = cm;
    }
    ar_barrier();
    flush_add<4 * 1024>(lds, rg, part, sFc1 * 4096, e1, bad);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {   // states_in: both limbs, block nb
      const int r_ = threadIdx.x, at = (r_ >> 5) * 64 + 32 * ((r_ >> 4) & 1) + (r_ & 15);
      int *q = reinterpret_cast<int *>(lds + rg + 4 * 4096) + nb * 512 + r_;
      const double v = (double)q[0] + (double)q[1024] * (1.0 / (double)(1 << kFix));
      q[0] = 0, q[1024] = 0;
      gadd(part, (unsigned)at * 4u, (sSin + nb) * 4096,
           bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf((float)v, es + ff - kFix));
    }
    { const int r_ = rg; rg = ro, ro = r_; }

    // ------- phases 4, 5: fc1 against the conv outputs (five blocks of 32 x1
    // planes), the conv weights
    const int ec = wg_exp(meta.dmax[4], bad);   // 2^ec above the largest |d conv|
    Op16 binr[3][2];   // the 90 window planes of this step, relative, in three blocks
    {
      const Planes Pp = k > 0 ? Pst : Ps0;
      const unsigned pbase = k > 0 ? (unsigned)((k - 1) * 12) * pB : 0u;
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        const int j = 32 * jb + row;
        TBlock tf, tp;
        tf.load(Pin, j < kH * kRD ? vtB : kDead, (unsigned)(k * kRD + 32 * jb) * pB + wcolB);
        tp.load(Pp, (j < kH * kRD && j % kRD < 3) ? (unsigned)(j % kRD) * pitchB + (unsigned)hi * 16u
                                                 : kDead, pbase + wcolB);
        float v[16], pv[16];
        tf.get(v);
        tp.get(pv);
        const float lim = __builtin_amdgcn_ldexpf(1.f, fi);
#pragma unroll
        for (int i = 0; i < 16; ++i)   // (clamped: see get_clamped)
          v[i] = __builtin_amdgcn_fmed3f(v[i] - pv[i], -lim, lim);
        split16(v, fi - kPreXc, binr[jb]);
      }
    }
#pragma unroll 1
    for (int eb = 0; eb < 5; ++eb) {
      if (eb == 3) {
        ar_barrier();
        flush_add<6 * 1024>(lds, rg, part, (sFc1 + 4) * 4096, e1 + fc, bad);
        const int r_ = rg; rg = ro, ro = r_;
      }
      // x block 2 + eb = the saved conv outputs e = 32 eb + row (channel
      // 4 eb + row / 8, position row % 8)
      float xv[16];
      get_clamped(tx, xv, fc);
      if (eb < 4) tx.load(Px1, vtN, (unsigned)(kW + 32 * (eb + 1)) * pN + wcolN);
      // (lane indices opaque per block: the scatter addresses below are made
      // here, not kept in registers through the whole sweep)
      int row_e = row, hi_e = hi;
      asm volatile("" : "+v"(row_e), "+v"(hi_e));
      {
        Op16 bx[2];
        split16(xv, fc - kPreX, bx);
        fc1_blocks(bx, lane_blk + rg + 2 * (eb < 3 ? eb : eb - 3) * 4096);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the cotangent of this block's conv outputs, trajectory-major, relu' from
      // the saved outputs, then its products against the window planes
      const char *tb = L16.b0 + gA + (a1c + 4 * eb) * kBlock16;   // (below 60 KB)
      f32x16 tt;
#pragma unroll
      for (int i = 0; i < 16; ++i) tt[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        Op16 w;
        w.h = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16);
        w.l = *reinterpret_cast<const u32x4 *>(tb + kb * kBlock16 + 1024);
        tt = mma3(x1s[kb], w, tt);
      }
      int E1[16];
      texp(ex1, hi, E1);
      float v[16], sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        v[i] = xv[i] > 0.f ? __builtin_amdgcn_ldexpf(tt[i], E1[i] - ec) : 0.f;   // / 2^ec
        sum += v[i];
      }
      // the conv block's rows of channels 4 eb .. 4 eb + 3 (accumulator layout:
      // channel ch = register (ch & 3) + 4 (ch >> 3) of half-wave (ch >> 2) & 1)
      char *cblk = lds + rConv + ((4 * (eb >> 1)) * 64 + 32 * (eb & 1)) * 4;
      sum += other_half(sum);
      // conv bias: the eight positions of a channel are eight neighbouring lanes
      // (DPP row shifts inside the group), then as the other biases - not a
      // column of the conv block, whose unit carries the windows' scale 2^fi
      // (single limb: with windows of 3e4 m the bias kept 4 bits there)
      sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
          0, __builtin_bit_cast(int, sum), 0x111 /* row_shr:1 */, 0xf, 0xf, true)) *
             ((row_e & 7) >= 1 ? 1.f : 0.f);
      sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
          0, __builtin_bit_cast(int, sum), 0x112 /* row_shr:2 */, 0xf, 0xf, true)) *
             ((row_e & 7) >= 2 ? 1.f : 0.f);
      sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
          0, __builtin_bit_cast(int, sum), 0x114 /* row_shr:4 */, 0xf, 0xf, true)) *
             ((row_e & 7) >= 4 ? 1.f : 0.f);
      if (hi_e == 0 && (row_e & 7) == 7)   // the group's last lane holds the channel's sum
        gadd(part, (unsigned)(32 + 4 * eb + (row_e >> 3)) * 4u, bias_soff,
             bad ? __builtin_nanf("") : __builtin_amdgcn_ldexpf(sum, ec));
      Op16 ac[2];
      split16(v, -kPreDc, ac);
#pragma unroll
      for (int jb = 0; jb < 3; ++jb) {
        if (APG_AR_KNOCKOUT & 8) break;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = mma3(ac[kk], binr[jb][kk], acc);
        // register 4 g + c of lane (hi, col): conv output row c + 8 g + 4 hi of the
        // block = channel 4 eb + g at position c + 4 hi, against window plane
        // j = 32 jb + col: tap q = j - 9 position of that channel's 27
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int q = 32 * jb + row_e - kRD * (cc + 4 * hi_e);
          if (q >= 0 && q < 27) {
#pragma unroll
            for (int g = 0; g < 4; ++g) lds_add(cblk + (g * 64 + q) * 4, acc[4 * g + cc]);
          }
        }
      }
    }
    e5 = e1 + fc, ecv = ec + fi;
  }
  ar_barrier();
  flush_tail();
  if (st_lo && A.grad_state0)
#pragma unroll
    for (int i = 0; i < 12; ++i) A.grad_state0[(size_t)i * B + b] = lam[i];
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
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

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_quad_mlp_workspace_floats(void) {
  // (+ the packed LearntDynamics weights of the closed-loop evaluation)
  return (kCfLds > kCbLds ? kCfLds : kCbLds) + kLearntFloats;
}

int apg_quad_mlp_loss_partials_count(int B) {
  return B <= 0 ? 0 : ((B + kTrajPerBlock - 1) / kTrajPerBlock) * (kThreads / kWave);
}

int apg_quad_mlp_rollout_fwd(const float *state0, const float *in_ref, float dt,
                             const ApgQuadParams *params,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *states, float *actions, float *feat,
                             float *x1, float *h, unsigned *relu_mask,
                             float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (B == 0) return APG_OK;
  if (!state0 || !in_ref || !states || !actions || !feat || !x1 || !h ||
      !relu_mask || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_rollout_fwd_kernel<false>, kCfLds)) return e;
    attr.set();
  }
  FwdArgs A;
  A.state0 = state0, A.in_ref = in_ref, A.states = states, A.actions = actions;
  A.feat = feat, A.x1 = x1, A.h = h, A.mask = relu_mask;
  A.tables = workspace;
  A.xmax = nullptr;
  A.c = make_const(*params, dt);
  A.B = B;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  hipLaunchKernelGGL(mlp_pack_cfwd_kernel, dim3((kCfLds + 255) / 256), dim3(256),
                     0, (hipStream_t)stream, P);
  hipLaunchKernelGGL(mlp_rollout_fwd_kernel<false>,
                     dim3((B + kTrajPerBlock - 1) / kTrajPerBlock),
                     dim3(kThreads), kCfLds * sizeof(float), (hipStream_t)stream,
                     A);
  return check_launch("quad_mlp_rollout_fwd");
}

int apg_quad_mlp_rollout_bwd(const float *state0, const float *states,
                             const float *actions, const float *ref,
                             int ref_cols, const float *x1, const float *h,
                             const unsigned *relu_mask, float dt,
                             const ApgQuadParams *params,
                             const ApgQuadLossWeights *weights,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *loss_partials, float *loss, float *d_pre,
                             float *d_zout, float *d_conv, float *grad_state0,
                             float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!state0 || !states || !actions || !ref || !x1 || !h || !relu_mask ||
      !loss_partials || !d_pre || !d_zout || !d_conv || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_rollout_bwd_kernel, kCbLds)) return e;
    attr.set();
  }
  BwdArgs A;
  A.state0 = state0, A.states = states, A.actions = actions, A.ref = ref;
  A.x1 = x1, A.h = h, A.mask = relu_mask;
  A.loss_partials = loss_partials, A.d_pre = d_pre, A.d_zout = d_zout;
  A.d_conv = d_conv, A.grad_state0 = grad_state0;
  A.tables = workspace;
  A.c = make_const(*params, dt);
  A.w = *weights;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  hipLaunchKernelGGL(mlp_pack_cbwd_kernel, dim3((kCbLds + 255) / 256), dim3(256),
                     0, st, P);
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  hipLaunchKernelGGL(mlp_rollout_bwd_kernel, dim3(blocks), dim3(kThreads),
                     kCbLds * sizeof(float), st, A);
  if (int e = check_launch("quad_mlp_rollout_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kThreads / kWave),
                                  loss, st);
  return APG_OK;
}

int apg_quad_mlp_closed_loop(const float *traj, int L, float dt,
                             const ApgQuadParams *params,
                             const ApgMlpPolicy *policy, int B, int H,
                             int max_steps, float thresh_div,
                             float thresh_stable, int test_time, float *div,
                             int *steps, float *drone, float *actions,
                             float *start_states, float *workspace,
                             apg_stream_t stream) {
  return apg_quad_mlp_closed_loop_env(traj, L, dt, params, nullptr, policy, B, H, max_steps,
                                      thresh_div, thresh_stable, test_time, div, steps, drone,
                                      actions, start_states, workspace, stream);
}

int apg_quad_mlp_closed_loop_env(const float *traj, int L, float dt,
                                 const ApgQuadParams *params, const ApgLearntResidual *learnt,
                                 const ApgMlpPolicy *policy, int B, int H, int max_steps,
                                 float thresh_div, float thresh_stable, int test_time,
                                 float *div, int *steps, float *drone, float *actions,
                                 float *start_states, float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (learnt && (!learnt->linear_at || !learnt->w1 || !learnt->b1 || !learnt->w2 ||
                 !learnt->b2)) {
    set_error("learnt simulator: weight pointer is NULL");
    return APG_ERR_ARG;
  }
  if (L <= kH || max_steps < 1) {
    set_error("closed loop needs L > %d reference rows and max_steps >= 1", kH);
    return APG_ERR_ARG;
  }
  const int T = max_steps < L + 1 ? max_steps : L + 1;
  if ((long long)B * 4 * ((long long)(T + 1) * 12 > (long long)L * 9
                              ? (long long)(T + 1) * 12 : (long long)L * 9) >=
      (1ll << 32) - 64) {
    set_error("B * steps too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  if (B == 0) return APG_OK;
  if (!traj || !div || !steps || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_closed_loop_kernel<true>, kCfLds + kLearntFloats)) return e;
    if (int e = raise_lds(mlp_closed_loop_kernel<false>, kCfLds)) return e;
    attr.set();
  }
  LoopArgs A;
  A.traj = traj, A.div = div, A.steps = steps, A.drone = drone;
  A.actions = actions, A.start = start_states, A.tables = workspace;
  A.c = make_const(*params, dt);
  A.B = B, A.L = L, A.T = T, A.test_time = test_time;
  A.thresh_div = thresh_div, A.thresh_stable = thresh_stable;
  A.learnt = learnt != nullptr;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_pack_cfwd_kernel, dim3((kCfLds + 255) / 256), dim3(256),
                     0, st, P);
  if (learnt)
    hipLaunchKernelGGL(learnt_pack_kernel, dim3((kLearntFloats + 255) / 256), dim3(256), 0, st,
                       *learnt, workspace + kCfLds);
  const dim3 grid((B + kTrajPerBlock - 1) / kTrajPerBlock);
  if (learnt)
    hipLaunchKernelGGL(mlp_closed_loop_kernel<true>, grid, dim3(kThreads),
                       (kCfLds + kLearntFloats) * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(mlp_closed_loop_kernel<false>, grid, dim3(kThreads), kCfLds * sizeof(float), st, A);
  return check_launch("quad_mlp_closed_loop");
}

int apg_quad_mlp_concurrent_workspace_floats(void) {
  return kCfLds + kCbLds;   // forward and reverse tables, packed by one launch
}

int apg_quad_mlp_concurrent_fwd_bwd(
    const float *feat, const float *in_ref, const float *state0, const float *ref,
    int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H,
    float *x1, float *h, unsigned *relu_mask, float *d_zout, float *d_pre,
    float *d_conv, float *loss_partials, float *loss, float *states,
    float *workspace, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!feat || !in_ref || !state0 || !ref || !x1 || !h || !relu_mask || !d_zout ||
      !d_pre || !d_conv || !loss_partials || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<false>, kCfLds)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_kernel, kCbLds)) return e;
    attr.set();
  }
  ConcArgs A;
  A.feat = feat, A.in_ref = in_ref, A.state0 = state0, A.ref = ref;
  A.x1 = x1, A.h = h, A.mask = relu_mask, A.d_zout = d_zout, A.d_pre = d_pre;
  A.d_conv = d_conv, A.states = states, A.loss_partials = loss_partials;
  A.tables = workspace;
  A.xmax = nullptr;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = kNA;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  const int fwd_blocks = (kCfLds + 255) / 256, bwd_blocks = (kCbLds + 255) / 256;
  hipLaunchKernelGGL(mlp_pack_pair_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0, st,
                     P, fwd_blocks);
  hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<false>, dim3(blocks), dim3(kThreads),
                     kCfLds * sizeof(float), st, A);
  A.tables = workspace + kCfLds;
  hipLaunchKernelGGL(mlp_concurrent_bwd_kernel, dim3(blocks), dim3(kThreads),
                     kCbLds * sizeof(float), st, A);
  if (int e = check_launch("quad_mlp_concurrent_fwd_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kThreads / kWave), loss, st);
  return APG_OK;
}


// [forward tables | in-sweep reverse tables | 4 | map of the resident tables]
int apg_quad_mlp_step_workspace_floats(void) { return kCfLds + kWgTabFloats + 4 + kMapInts; }

long long apg_quad_mlp_step_partials_floats(int B) {
  if (B <= 0) return 0;
  const long long wgs = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  // the workgroups' partials + the chunk sums of the first reduction level
  // + the forward kernel's per-wave x maxima
  const long long need = (wgs + (wgs + kRedChunk - 1) / kRedChunk) * kSlotsTm * 1024 + wgs * 32;
  // (also the scratch of the one-time table map: index parameters, their
  // tables, owners)
  const long long scratch = 2ll * kParamFloats + kCfLds + kWgTabFloats + 8;
  return need > scratch ? need : scratch;
}

namespace {
int concurrent_train_step(
    const ApgBatchRows *rows, const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream);
bool all_set(const ApgMlpPolicyGrads &g) {
  return g.w_s && g.b_s && g.conv_w && g.conv_b && g.w_1 && g.b_1 && g.w_2 && g.b_2 &&
         g.w_3 && g.b_3 && g.w_out && g.b_out;
}
}  // namespace

int apg_quad_mlp_concurrent_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, apg_event_t after_reverse,
    apg_stream_t stream) {
  ApgStepEvents ev = {nullptr, nullptr, after_reverse};
  return apg_quad_mlp_concurrent_train_step(
      state0, ref, ref_cols, dt, params, weights, policy, B, H, acts, relu_mask, d_zout,
      loss_partials, loss, grads, states, workspace, partials, nullptr, &ev, stream);
}

int apg_quad_mlp_concurrent_train_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream) {
  return concurrent_train_step(nullptr, state0, ref, ref_cols, dt, params, weights, policy, B, H,
                               acts, relu_mask, d_zout, loss_partials, loss, grads, states,
                               workspace, partials, update, events, stream);
}

int apg_quad_mlp_concurrent_train_step_rows(
    const ApgBatchRows *rows, int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H, float *acts,
    unsigned *relu_mask, float *d_zout, float *loss_partials, float *loss,
    const ApgMlpPolicyGrads *grads, float *states, float *workspace, float *partials,
    const ApgMlpSgdUpdate *update, const ApgStepEvents *events, apg_stream_t stream) {
  if (!rows) { set_error("rows is NULL"); return APG_ERR_ARG; }
  if (B > 0 && (!rows->index || !rows->normed || !rows->state0 || !rows->in_ref || !rows->ref)) {
    set_error("rows: NULL pointer");
    return APG_ERR_ARG;
  }
  if (rows->n_rows < 1 || rows->ld_normed < kNF || rows->ld_state0 < 12 ||
      rows->ld_in_ref < kH * kRD || rows->ld_ref < kH * ref_cols) {
    set_error("rows: need n_rows >= 1 and row strides of at least 15 / 12 / 90 / H x ref_cols");
    return APG_ERR_ARG;
  }
  const long long widest = rows->ld_in_ref > rows->ld_ref ? rows->ld_in_ref : rows->ld_ref;
  if (rows->n_rows * widest * 4 >= (1ll << 32) - 64 ||
      rows->n_rows * (long long)rows->ld_normed * 4 >= (1ll << 32) - 64 ||
      rows->n_rows * (long long)rows->ld_state0 * 4 >= (1ll << 32) - 64) {
    set_error("rows: a data-set tensor of 4 GiB or more (32-bit row offsets); gather the "
              "batch with apg_to_soa_multi instead");
    return APG_ERR_ARG;
  }
  return concurrent_train_step(rows, nullptr, nullptr, ref_cols, dt, params, weights, policy, B,
                               H, acts, relu_mask, d_zout, loss_partials, loss, grads, states,
                               workspace, partials, update, events, stream);
}

}  // extern "C"

namespace {
int concurrent_train_step(
    const ApgBatchRows *rows, const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (update && (!all_set(update->param) || !all_set(update->momentum_buf))) {
    set_error("update: parameter / momentum pointer is NULL");
    return APG_ERR_ARG;
  }
  if (update && !(update->lr == update->lr && update->momentum == update->momentum)) {
    set_error("update: lr / momentum is NaN");
    return APG_ERR_ARG;
  }
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  if (!grads || !all_set(*grads)) {
    set_error("gradient pointer is NULL");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    // (an update with zero gradients would still move the parameters by the
    // decaying momentum: not implemented)
    if (update) {
      set_error("update with B = 0 is not supported");
      return APG_ERR_ARG;
    }
    // no trajectory: zero gradients, zero loss
    const ApgMlpPolicyGrads &g = *grads;
    float *ptrs[12] = {g.w_s, g.b_s, g.conv_w, g.conv_b, g.w_1, g.b_1,
                       g.w_2, g.b_2, g.w_3, g.b_3, g.w_out, g.b_out};
    const size_t n[12] = {kW * kNF, kW, kNC * 27, kNC, kW * kN1, kW,
                          kW * kW, kW, kW * kW, kW, kNA * kW, kNA};
    for (int i = 0; i < 12; ++i)
      if (hipMemsetAsync(ptrs[i], 0, n[i] * sizeof(float), st) != hipSuccess)
        return check_launch("memset(grads)");
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if ((!rows && (!state0 || !ref)) || !acts || !relu_mask || !d_zout || !loss_partials ||
      !workspace || !partials) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  if ((long long)B * 4 * kActPlanes >= (1ll << 32) - 64) {
    set_error("B too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<false>, kCfLds)) return e;
    if (int e = raise_lds(mlp_concurrent_fwd_kernel<true>, kCfRowsLds)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_tm_kernel<false>, kLdsAll / 4)) return e;
    if (int e = raise_lds(mlp_concurrent_bwd_tm_kernel<true>, kLdsAll / 4)) return e;
    attr.set();
  }
  const size_t plane = (size_t)B;
  ConcArgs A;
  A.feat = acts + pFeat * plane, A.in_ref = acts + pInr * plane;
  A.state0 = state0, A.ref = ref;
  A.x1 = acts + pX1 * plane, A.h = acts + pH1 * plane, A.mask = relu_mask;
  A.d_zout = d_zout, A.d_pre = nullptr, A.d_conv = nullptr;
  A.states = states, A.loss_partials = loss_partials;
  A.tables = workspace;
  // (behind the workgroups' partials and the chunk sums)
  {
    const long long wgs_ = (B + kTrajPerBlock - 1) / kTrajPerBlock;
    A.xmax = partials + (size_t)((wgs_ + (wgs_ + kRedChunk - 1) / kRedChunk) * kSlotsTm * 1024);
  }
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  A.index = nullptr, A.o_feat = A.o_in_ref = nullptr;
  if (rows) {
    A.index = rows->index;
    A.o_feat = acts + pFeat * plane, A.o_in_ref = acts + pInr * plane;
    A.feat = rows->normed, A.in_ref = rows->in_ref, A.state0 = rows->state0, A.ref = rows->ref;
    A.ld_feat = rows->ld_normed, A.ld_in_ref = rows->ld_in_ref;
    A.ld_state0 = rows->ld_state0, A.ld_ref = rows->ld_ref;
    const auto bytes = [&](int ld) { return (unsigned)(rows->n_rows * (long long)ld * 4); };
    A.bytes_feat = bytes(A.ld_feat), A.bytes_in_ref = bytes(A.ld_in_ref);
    A.bytes_state0 = bytes(A.ld_state0), A.bytes_ref = bytes(A.ld_ref);
  }
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = kNA;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  const int fwd_blocks = (kCfLds + 255) / 256, bwd_blocks = (kWgTabFloats + 255) / 256;
  // resident tables (ApgMlpSgdUpdate.resident): 2 - the workspace holds the tables
  // of exactly these parameters, left by the previous call's second stage: no
  // pack launch; 1 - pack, and build the map the second stage scatters by; 3 -
  // pack (somebody else wrote the parameters), the workspace's map is still good
  const int resident = update ? update->resident : 0;
  if (resident < 0 || resident > 3) {
    set_error("update: resident must be 0, 1, 2 or 3");
    return APG_ERR_ARG;
  }
  int *map = reinterpret_cast<int *>(workspace + kCfLds + kWgTabFloats + 4);
  if (resident == 1) {
    // index-valued parameters -> their tables -> owners -> map (scratch: partials,
    // overwritten by the step afterwards)
    float *par = partials, *tab = partials + kParamFloats;
    int *owner = reinterpret_cast<int *>(tab + kCfLds + kWgTabFloats + 4);
    // (the pack kernel leaves the gaps of the float tables alone: whatever the
    // scratch held there would be read as parameter indices)
    // (... and the owner table, so that nothing in this path indexes by garbage)
    if (hipMemsetAsync(tab, 0,
                       (size_t)(kCfLds + kWgTabFloats + 4 + kParamFloats) * sizeof(float),
                       st) != hipSuccess)
      return check_launch("memset(table map scratch)");
    hipLaunchKernelGGL(tabmap_iota_kernel, dim3((kParamFloats + 255) / 256), dim3(256), 0, st,
                       par);
    PackArgs Q;
    const ApgMlpPolicyGrads f = params_in(par);
    Q.pol = ApgMlpPolicy{f.w_s, f.b_s, f.conv_w, f.conv_b, f.w_1, f.b_1,
                         f.w_2, f.b_2, f.w_3, f.b_3, f.w_out, f.b_out};
    Q.dst = tab, Q.head_rows = kNA;
    hipLaunchKernelGGL(mlp_pack_step_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0,
                       st, Q, fwd_blocks);
    hipLaunchKernelGGL(tabmap_owner_kernel, dim3(kSlotsTm * 4), dim3(256), 0, st, par, owner,
                       map, kSlotsTm, uBias);
    hipLaunchKernelGGL(tabmap_invert_kernel, dim3((kCfLds + kWgTabFloats + 255) / 256),
                       dim3(256), 0, st, tab, owner, map);
  }
  if (resident != 2)
    hipLaunchKernelGGL(mlp_pack_step_kernel, dim3(fwd_blocks + bwd_blocks), dim3(256), 0,
                       st, P, fwd_blocks);
  // (the tables are packed while the caller's producer of acts / state0 / ref -
  // a gather on another stream - may still be running)
  if (events && events->inputs_ready &&
      hipStreamWaitEvent(st, (hipEvent_t)events->inputs_ready, 0) != hipSuccess)
    return check_launch("hipStreamWaitEvent(inputs_ready)");
  if (rows)
    hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<true>, dim3(blocks), dim3(kThreads),
                       kCfRowsLds * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(mlp_concurrent_fwd_kernel<false>, dim3(blocks), dim3(kThreads),
                       kCfLds * sizeof(float), st, A);
  if (events && events->after_forward &&
      hipEventRecord((hipEvent_t)events->after_forward, st) != hipSuccess)
    return check_launch("hipEventRecord(after_forward)");
  WgArgs W;
  W.acts = acts, W.mask = relu_mask, W.d_zout = d_zout, W.part = partials;
  W.tables = workspace + kCfLds, W.B = B, W.xmax = A.xmax;
  W.index = nullptr, W.r_feat = W.r_in_ref = nullptr;
  if (rows && B % kTrajPerBlock) {
    // A ragged last workgroup reads its dead trajectories' x^T entries past the
    // end of a plane - the head of the next plane: finite numbers, times a zero
    // cotangent.  Behind the LAST activation plane that is the window region,
    // which nobody writes in this mode: keep its head finite.
    const size_t head = (size_t)kTrajPerBlock * 4;
    const size_t region = (size_t)kH * kRD * plane * 4;
    if (hipMemsetAsync(acts + pInr * plane, 0, head < region ? head : region, st) != hipSuccess)
      return check_launch("memset(window planes' head)");
  }
  if (rows) {
    W.index = rows->index, W.r_feat = rows->normed, W.r_in_ref = rows->in_ref;
    W.ld_feat = A.ld_feat, W.ld_in_ref = A.ld_in_ref;
    W.bytes_feat = A.bytes_feat, W.bytes_in_ref = A.bytes_in_ref;
    hipLaunchKernelGGL(mlp_concurrent_bwd_tm_kernel<true>, dim3(blocks), dim3(kThreads), kLdsAll,
                       st, W);
  } else {
    hipLaunchKernelGGL(mlp_concurrent_bwd_tm_kernel<false>, dim3(blocks), dim3(kThreads), kLdsAll,
                       st, W);
  }
  // the inputs (activation planes, state0, ref) are not read past this point:
  // a caller that pipelines batches may start refilling the NEXT batch's
  // buffers behind this event while the second stage and the update run
  if (events && events->after_reverse &&
      hipEventRecord((hipEvent_t)events->after_reverse, st) != hipSuccess)
    return check_launch("hipEventRecord(after_reverse)");
  WgReduceArgs R;
  R.part = partials, R.g = *grads, R.loss_partials = loss_partials, R.loss = loss;
  R.loss_sum = rows && loss ? rows->running_loss : nullptr;
  R.ws = reinterpret_cast<char *>(workspace), R.map = resident ? map : nullptr;
  R.wgs = blocks, R.n_partials = blocks * (kThreads / kWave);
  R.n_slots = kSlotsTm, R.bias_slot = uBias, R.conv_src = 1, R.bias_src = 8;
  R.head_rows = kNA, R.conv_bias_here = false;
  const int columns = (R.n_slots * 1024 + 255) / 256;
  R.update = update != nullptr;
  R.param = update ? update->param : *grads, R.mom = update ? update->momentum_buf : *grads;
  R.lr = update ? update->lr : 0.0, R.momentum = update ? update->momentum : 0.0;
  if (blocks > kRedChunk) {
    const int chunks = (blocks + kRedChunk - 1) / kRedChunk;
    float *chunk_sums = partials + (size_t)blocks * R.n_slots * 1024;
    hipLaunchKernelGGL(mlp_wgrad_reduce1_kernel, dim3(columns, chunks), dim3(256), 0, st,
                       partials, chunk_sums, blocks, R.n_slots);
    R.part = chunk_sums, R.wgs = chunks;
  }
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(columns), dim3(256), 0, st, R);
  return check_launch("quad_mlp_concurrent_step");
}
}  // namespace

extern "C" {

int apg_quad_mlp_rollout_step_workspace_floats(void) { return kCfLds + kArTabFloats + 4; }

long long apg_quad_mlp_rollout_step_partials_floats(int B) {
  if (B <= 0) return 0;
  const long long wgs = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  // the workgroups' accumulators + the chunk sums of the first reduction level
  // + the forward sweep's per-wave, per-step x maxima
  return (wgs + (wgs + kRedChunk - 1) / kRedChunk) * kSlotsTm * 1024 +
         wgs * (kThreads / kWave) * kH * 4;
}

int apg_quad_mlp_rollout_train_step(
    const float *state0, const float *in_ref, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *states, float *actions, float *acts,
    unsigned *relu_mask, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *grad_state0, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    apg_stream_t stream) {
  if (int e = check_mlp(params, policy, B, H)) return e;
  if (update && (!all_set(update->param) || !all_set(update->momentum_buf))) {
    set_error("update: parameter / momentum pointer is NULL");
    return APG_ERR_ARG;
  }
  if (update && !(update->lr == update->lr && update->momentum == update->momentum)) {
    set_error("update: lr / momentum is NaN");
    return APG_ERR_ARG;
  }
  if (update && update->resident != 0) {
    set_error("update: resident operand tables are the concurrent step's (resident must be 0)");
    return APG_ERR_ARG;
  }
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  if (!grads || !all_set(*grads)) {
    set_error("gradient pointer is NULL");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (update) {
      set_error("update with B = 0 is not supported");
      return APG_ERR_ARG;
    }
    const ApgMlpPolicyGrads &g = *grads;
    float *ptrs[12] = {g.w_s, g.b_s, g.conv_w, g.conv_b, g.w_1, g.b_1,
                       g.w_2, g.b_2, g.w_3, g.b_3, g.w_out, g.b_out};
    const size_t n[12] = {kW * kNF, kW, kNC * 27, kNC, kW * kN1, kW,
                          kW * kW, kW, kW * kW, kW, 4 * kW, 4};
    for (int i = 0; i < 12; ++i)
      if (hipMemsetAsync(ptrs[i], 0, n[i] * sizeof(float), st) != hipSuccess)
        return check_launch("memset(grads)");
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!state0 || !in_ref || !ref || !states || !actions || !acts || !relu_mask ||
      !loss_partials || !workspace || !partials) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(mlp_rollout_fwd_kernel<true>, kCfLds)) return e;
    if (int e = raise_lds(mlp_rollout_bwd_tm_kernel, kLdsAll / 4)) return e;
    attr.set();
  }
  const size_t N = (size_t)B * kH;
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  float *xmax = partials + (size_t)(apg_quad_mlp_rollout_step_partials_floats(B) -
                                    (long long)blocks * (kThreads / kWave) * kH * 4);
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  const int fwd_blocks = (kCfLds + 255) / 256, bwd_blocks = (kArTabFloats + 255) / 256;
  hipLaunchKernelGGL(mlp_pack_ar_kernel, dim3(fwd_blocks + bwd_blocks + 1), dim3(256), 0, st,
                     P, fwd_blocks);
  FwdArgs F;
  F.state0 = state0, F.in_ref = in_ref, F.states = states, F.actions = actions;
  F.feat = acts, F.x1 = acts + kNF * N, F.h = acts + (kNF + kN1) * N, F.mask = relu_mask;
  F.tables = workspace;
  F.xmax = xmax;
  F.c = make_const(*params, dt);
  F.B = B;
  hipLaunchKernelGGL(mlp_rollout_fwd_kernel<true>, dim3(blocks), dim3(kThreads),
                     kCfLds * sizeof(float), st, F);
  ArTmArgs A;
  A.state0 = state0, A.states = states, A.actions = actions, A.ref = ref, A.in_ref = in_ref;
  A.feat = F.feat, A.x1 = F.x1, A.h = F.h, A.mask = relu_mask;
  A.loss_partials = loss_partials, A.part = partials, A.grad_state0 = grad_state0;
  A.tables = workspace + kCfLds, A.xmax = xmax;
  A.c = F.c;
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  hipLaunchKernelGGL(mlp_rollout_bwd_tm_kernel, dim3(blocks), dim3(kThreads), kLdsAll, st, A);
  WgReduceArgs R;
  R.part = partials, R.g = *grads, R.loss_partials = loss_partials, R.loss = loss, R.loss_sum = nullptr;
  R.ws = nullptr, R.map = nullptr;
  R.wgs = blocks, R.n_partials = blocks * (kThreads / kWave);
  R.n_slots = kSlotsTm, R.bias_slot = uBias, R.conv_src = 1, R.bias_src = 8, R.head_rows = 4;
  R.conv_bias_here = true;
  const int columns = (R.n_slots * 1024 + 255) / 256;
  R.update = update != nullptr;
  R.param = update ? update->param : *grads, R.mom = update ? update->momentum_buf : *grads;
  R.lr = update ? update->lr : 0.0, R.momentum = update ? update->momentum : 0.0;
  if (blocks > kRedChunk) {
    const int chunks = (blocks + kRedChunk - 1) / kRedChunk;
    float *chunk_sums = partials + (size_t)blocks * R.n_slots * 1024;
    hipLaunchKernelGGL(mlp_wgrad_reduce1_kernel, dim3(columns, chunks), dim3(256), 0, st,
                       partials, chunk_sums, blocks, R.n_slots);
    R.part = chunk_sums, R.wgs = chunks;
  }
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(columns), dim3(256), 0, st, R);
  return check_launch("quad_mlp_rollout_train_step");
}

}  // extern "C"
