// lstm.hip - K7: the LSTM-policy quadrotor unroll with the policy INSIDE the
// kernel, on the matrix cores (BASELINE config 5: quadrotor, LSTM recurrent
// mode, H = 10).
//
// Replaces, for train_mode == "LSTM", the loop of
//   TrainDrone.train_recurrent_model       scripts/train_drone.py:113-173
//   LSTM_NEW.forward (conv branch)         neural_control/models/rnn.py:35-51
//   state_preprocessing                    neural_control/dataset.py:207-220
//   FlightmareDynamics / quad_mpc_loss     (see quad.hip)
// with two launches: a forward sweep (policy + dynamics for all H steps) and a
// reverse sweep (BPTT through head, LSTM cell, conv window, feature
// construction and dynamics).  Pinned window semantics:
// window_k = in_ref[k : k+H], its position columns made relative to the
// CURRENT position, copied - never in place.
//
// LSTM_NEW(15, 10, 9, 4, conv=1): x = [15 state features, relu(conv1d(9 -> 20,
// k = 3)) (160)], gates = W_ih x + W_hh h + b (32 = i, f, g, o x 8 units),
// c' = sig(f) c + sig(i) tanh(g), h' = sig(o) tanh(c'), a = sig(W_out h' + b).
// The 32 x 183 gate projection and the conv are GEMM-shaped with the batch as
// N: they run on the matrix cores in the layout of policy_mfma.h (one wave =
// 32 trajectories, lane l works for trajectory l & 31) - rounds 1-2 as
// v_mfma_f32_32x32x2_f32, since round 3 as v_mfma_f32_32x32x16_f16 products of
// fp16-split operands (policy_mfma16.h: fp32 accuracy, 90 + 42 instructions
// of 32 cycles per step and wave instead of 228 + 112 of 64).  The gate
// accumulator puts (i, f, g, o) of hidden unit u = r + 4 (l >> 5) into
// registers r, 4 + r, 8 + r, 12 + r of ONE lane, so the cell update is
// lane-local, and h' (4 registers per lane) is directly the B operand of the
// next step's W_hh product: the recurrence needs no data movement at all.
//
// Weight gradients: the reverse sweep writes the per-(step, trajectory)
// cotangent planes of the gate and head pre-activations ([feature][H*B]) and the
// conv cotangents summed along the window diagonals; rounds 1-5 reduced them
// against the forward's saved inputs with apg_planes_gemm, since round 6 two
// trajectory-major kernels below do (lstm_gate_wgrad_kernel - which recomputes the
// conv inputs instead of reading 160 stored planes - and lstm_conv_wgrad_kernel).
#include "apg_device.h"
#include "policy_mfma.h"
#include "policy_mfma16.h"
#include "policy_tm.h"
#include "quad_math.h"
#include "learnt_residual.h"

namespace apg {
namespace {

constexpr int kH = 10;            // horizon == window length
constexpr int kRD = 9;            // reference columns fed to the policy
constexpr int kNF = 15;           // state features
constexpr int kNC = 20;           // conv output channels
constexpr int kNP = kH - 2;       // conv output positions (kernel 3)
constexpr int kNX = kNF + kNC * kNP;  // LSTM input width (175)
// What the reverse sweep leaves for the conv weight gradient.  The window of
// (step k, position pos, tap t) is reference row k + pos + t, so
//   dW[ch][c][t] = sum_{k,pos,n} d[ch][pos][k][n] R[k+pos+t][c][n]
//                = sum_{sigma,n} G[ch][sigma][n] R[sigma+t][c][n],
//   G[ch][sigma] = sum_{k+pos=sigma} d[ch][pos][k]:
// 17 diagonal sums per channel instead of 80 (pos, k) planes.  A lane holds
// the positions pos = 4 hi + ii of a channel, so each half-wave keeps its own
// diagonals tau = k + ii (13 of them, sigma = tau + 4 hi) in four sliding
// registers per channel and stores a diagonal when its last term is in.
//   planes [0, kConvP):  G[ch][hi][tau]   (kNC x 2 x 13, B floats each)
//   planes [kConvP, +kNC*kH): P[ch][k] = sum_pos d[ch][pos][k] (for the
//                        relative-position shift of window columns 0..2)
constexpr int kTau = kH + 3;               // diagonals per half-wave (13)
constexpr int kConvP = kNC * 2 * kTau;     // 520
constexpr int kConvPlanes = kConvP + kNC * kH;  // 720
constexpr int kNH = 8;            // hidden units
constexpr int kNG = 4 * kNH;      // gate pre-activations (i, f, g, o)
constexpr int kThreads = 256;
constexpr int kTrajPerBlock = kThreads / 2;

__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.f + __expf(-x));
}

// ------------------------------------------------------------ forward sweep
struct PackArgs {
  ApgLstmPolicy pol;
  float *dst;
};

// Forward tables (fp16 split operands, policy_mfma16.h): the small fp32 tables
// indexed by the half-wave - head weights for the VALU [4 j][4 r][2], gate
// bias b_ih + b_hh [16][2], conv bias [16][2], head bias [4] - then 16
// A-operand blocks of 2 KB:
// W_ih on the features, W_hh, conv [kb], W_ih on the conv outputs of a
// position pair [pp][kb].
constexpr int hTo = 0, hTbg = 32, hTbc = 64, hBo = 96;   // floats
constexpr int hA = 512;                                  // bytes: first A block
constexpr int nF = 0, nH = 1, nC = 2, nG = 4, nBlocks16 = 16;
constexpr int kFwd16Lds = (hA + nBlocks16 * kBlock16) / 4;  // 8 320 floats = 33 280 B

// weight behind k-slot j (of 8) of A block n for half-wave hi, gate row `row`
__device__ __forceinline__ float fwd16_weight(const ApgLstmPolicy &p, int n, int row, int j,
                                              int hi) {
  if (n == nF) {                       // features 8 hi + j
    const int k = 8 * hi + j;
    return k < kNF ? p.w_ih[row * kNX + k] : 0.f;
  }
  if (n == nH)                         // hidden units j + 4 hi, 4 of the 8 slots
    return j < 4 ? p.w_hh[row * kNH + j + 4 * hi] : 0.f;
  if (n < nG) {                        // conv: slot s = (column j', tap), 15 of 16
    const int sl = (n - nC) * 8 + j, jc = sl / 3, tap = sl % 3, q = hi ? 4 + jc : jc;
    return (sl < 15 && row < kNC && (hi || jc < 4)) ? p.conv_w[row * 27 + q * 3 + tap] : 0.f;
  }
  const int m = n - nG, pp = m / 3, sl = (m % 3) * 8 + j;
  const int pos = 2 * pp + sl / 12, ch = rrow(sl % 12) + 4 * hi;
  return ch < kNC ? p.w_ih[row * kNX + kNF + ch * kNP + pos] : 0.f;
}

// the A-operand blocks of a table: four entries per round, every weight of a round
// read before the first is split and stored (a pack by ONE workgroup - the step's
// tail - is a chain of dependent loads otherwise)
template <typename W>
__device__ __forceinline__ void pack_blocks16(unsigned *dst, int base, int blocks, int tid, int T,
                                              W weight) {
  const int n_idx = blocks * 64 * 4;
  for (int idx0 = tid; idx0 < n_idx; idx0 += 4 * T) {
    float w[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = idx0 + r * T;
      const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
      w[r][0] = idx < n_idx ? weight(n, l & 31, 2 * q, l >> 5) : 0.f;
      w[r][1] = idx < n_idx ? weight(n, l & 31, 2 * q + 1, l >> 5) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = idx0 + r * T;
      if (idx >= n_idx) break;
      const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
      unsigned h, lo;
      split_pair(w[r][0], w[r][1], h, lo);
      dst[(base + n * kBlock16) / 4 + l * 4 + q] = h;
      dst[(base + n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
    }
  }
}

__device__ __forceinline__ void pack_fwd16(const PackArgs &A, int tid, int T) {
  const ApgLstmPolicy &p = A.pol;
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  pack_blocks16(dst, hA, nBlocks16, tid, T,
                [&](int n, int row, int j, int hi) { return fwd16_weight(p, n, row, j, hi); });
  for (int idx = tid; idx < 32; idx += T) {
    const int hi = idx & 1, r = (idx >> 1) & 3, j = idx >> 3;
    A.dst[hTo + idx] = p.w_out[j * kNH + r + 4 * hi];
    const int i = idx >> 1, row = rrow(i) + 4 * hi;
    A.dst[hTbg + idx] = p.b_ih[row] + p.b_hh[row];
    A.dst[hTbc + idx] = row < kNC ? p.conv_b[row] : 0.f;
  }
  for (int idx = tid; idx < 4; idx += T) A.dst[hBo + idx] = p.b_out[idx];
}
__global__ __launch_bounds__(256) void lstm_pack_fwd16_kernel(PackArgs A) {
  pack_fwd16(A, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

struct FwdArgs {
  const float *state0, *in_ref, *h0, *c0;
  float *states, *actions, *x, *gates, *hc, *hnew;
  unsigned *mask;        // [5][N] relu bits of the conv outputs
  const float *tables;   // packed operand tables (lstm_pack_fwd16_kernel)
  QuadConst c;
  int B;
  // ROWS: the minibatch is named by row numbers of the whole data set's tensors
  // (TrainBase.run_epoch's batch selection, scripts/train_base.py:191-194) and
  // this sweep reads state0 [n][12] and in_ref [n][>= 2H][9] through them; it
  // WRITES the state0 / in_ref planes its followers read (no gather pass)
  const long long *index;
  const float *r_state0, *r_in_ref;
  unsigned bytes_state0, bytes_in_ref;   // n_rows x ld x 4
  int ld_state0, ld_in_ref;
};

// LEGACY (forward only): the loop of scripts/train_drone.py:138-142 AS SHIPPED -
// the window is a view of the batch and the relative-position subtraction
// writes through it: every step shifts the rows its window holds AGAIN
// (SURVEY.md 8a A4 `legacy_inplace_ref`; the pinned semantics copy the window)
// APG_LF_KNOCKOUT (experiment builds only, results WRONG on purpose): 1 the rows
// sweep does not store the state0 / in_ref planes its followers read, 2 its row
// numbers are the batch positions (consecutive rows, no index load)
#ifndef APG_LF_KNOCKOUT
#define APG_LF_KNOCKOUT 0
#endif
template <bool ROWS, bool LEGACY = false>
__global__ __launch_bounds__(kThreads) void lstm_rollout_fwd_kernel(FwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds_issue(lds, A.tables, kFwd16Lds);   // (waited for behind the first loads)
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;             // see Planes: dead lanes never branch
  const bool st_lo = live && hi == 0;  // per-trajectory stores: lower half only
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pin(A.in_ref, 2 * kH * kRD, pitchB);
  const Planes Ph0(A.h0, kNH, pitchB), Pc0(A.c0, kNH, pitchB);
  const Planes Pst(A.states, kH * 12, pitchB), Pac(A.actions, kH * 4, pitchB);
  const Planes Px(A.x, kNF, pitchN), Pg(A.gates, kNG, pitchN);
  const Planes Phc(A.hc, 2 * kNH, pitchN), Phn(A.hnew, kNH, pitchN);
  const Planes Pmk(A.mask, 5, pitchN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vb_lo = st_lo ? vb : kDead;
  const unsigned vb_u = live ? vb + (hi ? 4u * pitchB : 0u) : kDead;  // unit r + 4 hi

  // ROWS: the lane's data-set row (range-checked buffers: a row number beyond
  // the data set reads zeros); the planes of what it reads are this sweep's output
  const Planes Rs0(A.r_state0, 1, ROWS ? A.bytes_state0 : 0u);
  const Planes Rin(A.r_in_ref, 1, ROWS ? A.bytes_in_ref : 0u);
  unsigned vr_s = kDead, vr_in = kDead;
  if (ROWS && live) {
    const unsigned rown = (APG_LF_KNOCKOUT & 2) ? (unsigned)b : (unsigned)A.index[b];
    vr_s = rown * (unsigned)A.ld_state0 * 4u;
    vr_in = rown * (unsigned)A.ld_in_ref * 4u + (hi ? 16u : 0u);   // column + 4 hi
  }
  // reference row r, columns 4 hi .. + 4 of the lane's trajectory (ROWS: 20
  // contiguous bytes of its data-set row as 16 + 4, and their planes for the
  // followers: the upper half's first column is the lower half's last)
  auto window_row = [&](int r, unsigned pB, float (&v)[5]) {
    if (!ROWS) {
#pragma unroll
      for (int j = 0; j < 5; ++j) v[j] = Pin.ld(vb_u, (r * kRD + j) * pB);
      return;
    }
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(Rin.rsrc, (int)vr_in, r * kRD * 4, 0);
    const f32x4_ f = __builtin_bit_cast(f32x4_, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f[j];
    v[4] = Rin.ld(vr_in, (r * kRD + 4) * 4);
    if (!(APG_LF_KNOCKOUT & 1))
#pragma unroll
      for (int j = 0; j < 5; ++j) Pin.st(hi && j == 0 ? kDead : vb_u, (r * kRD + j) * pB, v[j]);
  };
  float s[12], h[4], cell[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    if (!ROWS) {
      s[i] = Ps0.ld(vb, i * pitchB);
    } else if (i % 4 == 0) {   // 48 contiguous bytes: three 16-byte loads
      const f32x4_ f = __builtin_bit_cast(
          f32x4_, __builtin_amdgcn_raw_buffer_load_b128(Rs0.rsrc, (int)vr_s, i * 4, 0));
      s[i] = f[0], s[i + 1] = f[1], s[i + 2] = f[2], s[i + 3] = f[3];
    }
  }
  if (ROWS && !(APG_LF_KNOCKOUT & 1))
#pragma unroll
    for (int i = 0; i < 12; ++i) Ps0.st(vb_lo, i * pitchB, s[i]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h[r] = Ph0.ld(vb_u, r * pitchB);
    cell[r] = Pc0.ld(vb_u, r * pitchB);
  }
  // sliding reference window, raw values: columns 0..4 in the lower half,
  // 4..8 in the upper half (see fwd16_weight)
  float w[kH][5];
#pragma unroll
  for (int r = 0; r < kH; ++r) window_row(r, pitchB, w[r]);
  if (ROWS) {   // (the last row: no step reads it, the planes are whole all the same)
    float last[5];
    window_row(2 * kH - 1, pitchB, last);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // the tables

#pragma unroll 1
  for (int k = 0; k < kH; ++k) {
    const unsigned pB = opaque(pitchB), pN = opaque(pitchN);
    const unsigned col = (unsigned)b * 4u + (unsigned)k * pitchB;  // column k*B + b
    const unsigned vn_lo = st_lo ? col : kDead;
    const unsigned vr = live ? col + (hi ? 4u * pitchN : 0u) : kDead;   // + row 4 hi
    const unsigned vm = live ? col + (hi ? pitchN : 0u) : kDead;        // + mask word hi
    // the row that slides in for the next step: requested first (ROWS: a scattered
    // read through the index - a step of latency to hide)
    // (the plane variant fetches it at the end of the step: five more live
    // registers put that kernel over the 256 of two waves per SIMD)
    float wn[5];
    if (ROWS && k + 1 < kH) window_row(k + kH, pB, wn);
    const Trig t = make_trig(&s[3]);
    float feat[kNF];
    quad_features(s, t, feat);
#pragma unroll
    for (int j = 0; j < kNF; ++j) Px.st(vn_lo, j * pN, feat[j]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Phc.st(vr, r * pN, h[r]);
      Phc.st(vr, (kNH + r) * pN, cell[r]);
    }
    // gates on the 16-bit matrix pipe (policy_mfma16.h): every operand as two
    // fp16 terms, three products per k-block; two accumulators keep the
    // matrix instructions of neighbouring blocks independent
    f32x16 g0, g1;
#pragma unroll
    for (int i = 0; i < 16; ++i) g0[i] = L.T(hTbg + i * 2), g1[i] = 0.f;
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
      g0 = mma3(L16.A(hA, nF), split8(v), g0);
      const float vh[8] = {h[0], h[1], h[2], h[3], 0.f, 0.f, 0.f, 0.f};
      g1 = mma3(L16.A(hA, nH), split8(vh), g1);
    }
    // conv (one 32-row block per window position) feeding the gates; the
    // window relative to the current position is split once per step
    unsigned mbits[3] = {0u, 0u, 0u};
    float sub[3] = {hi ? 0.f : s[0], hi ? 0.f : s[1], hi ? 0.f : s[2]};
    if (LEGACY) {   // the shift stays in the window
#pragma unroll
      for (int r = 0; r < kH; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) w[r][j] -= sub[j];
      sub[0] = sub[1] = sub[2] = 0.f;
    }
#pragma unroll
    for (int pp = 0; pp < kNP / 2; ++pp) {
      float rv[24];  // relu(conv) of positions 2 pp, 2 pp + 1: registers 0..11 each
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pos = 2 * pp + e;
        f32x16 cv;
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[i] = L.T(hTbc + i * 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          // window slot s = (column s / 3, tap s % 3) relative to the current
          // position, two slots per split: high and low terms land in the operand
          // registers directly (round 6; rounds 3-5 split every window value on its
          // own - five instructions - and paired the halves with two v_perm per
          // slot pair: 66 instructions per position, now 41; the same numbers)
          Op16 x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int s0 = kb * 8 + 2 * q, s1 = s0 + 1;
            const float v0 =
                s0 / 3 < 3 ? w[pos + s0 % 3][s0 / 3] - sub[s0 / 3] : w[pos + s0 % 3][s0 / 3];
            float v1 = 0.f;
            if (s1 < 15)
              v1 = s1 / 3 < 3 ? w[pos + s1 % 3][s1 / 3] - sub[s1 / 3] : w[pos + s1 % 3][s1 / 3];
            unsigned h_, l_;
            split_pair(v0, v1, h_, l_);
            x.h[q] = h_, x.l[q] = l_;
          }
          cv = mma3(L16.A(hA, nC + kb), x, cv);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {  // rows r(i) + 4 hi < 20 are real channels
          float v = cv[i];
          mbits[i >> 2] |= (v > 0.f ? 1u : 0u) << ((i & 3) * 8 + pos);
          // (round 6: relu(conv) is NOT stored - 160 of the 236 planes this
          // sweep wrote; lstm_gate_wgrad_kernel recomputes it from the window)
          rv[e * 12 + i] = relu1(v);
        }
      }
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
        if (kb & 1) g1 = mma3(L16.A(hA, nG + pp * 3 + kb), split8(v), g1);
        else g0 = mma3(L16.A(hA, nG + pp * 3 + kb), split8(v), g0);
      }
    }
    // relu mask, trajectory-indexed: bit e = ch*8 + pos of word e >> 5
#pragma unroll
    for (int g = 0; g < 3; ++g) Pmk.stu(g < 2 ? vm : vn_lo, 2 * g * pN, mbits[g]);
    // cell update: (i, f, g, o) of unit r + 4 hi are registers r, 4+r, 8+r, 12+r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = sigmoid_fast(g0[r] + g1[r]);
      const float gf = sigmoid_fast(g0[4 + r] + g1[4 + r]);
      const float gg = tanh_fast(g0[8 + r] + g1[8 + r]);
      const float go = sigmoid_fast(g0[12 + r] + g1[12 + r]);
      Pg.st(vr, r * pN, gi);
      Pg.st(vr, (kNH + r) * pN, gf);
      Pg.st(vr, (2 * kNH + r) * pN, gg);
      Pg.st(vr, (3 * kNH + r) * pN, go);
      cell[r] = fmaf(gf, cell[r], gi * gg);
      h[r] = go * tanh_fast(cell[r]);
      Phn.st(vr, r * pN, h[r]);
    }
    // head on the VALU: each half sums its 4 of the 8 units
    float act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) z = fmaf(L.T(hTo + (j * 4 + r) * 2), h[r], z);
      z += other_half(z);
      act[j] = sigmoidf_(z + L.U(hBo + j));
      Pac.st(vb_lo, (k * 4 + j) * pB, act[j]);
    }
    quad_step(s, act, c, t);
#pragma unroll
    for (int i = 0; i < 12; ++i) Pst.st(vb_lo, (k * 12 + i) * pB, s[i]);
    if (k + 1 < kH) {  // slide the window: drop row 0, fetch in_ref row k + H
#pragma unroll
      for (int r = 0; r + 1 < kH; ++r)
#pragma unroll
        for (int j = 0; j < 5; ++j) w[r][j] = w[r + 1][j];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[kH - 1][j] = wn[j];
      if (!ROWS) window_row(k + kH, pB, w[kH - 1]);
    }
  }
}

// ------------------------------------------------------ closed-loop evaluation
// N2 for the LSTM controller: QuadEvaluator.follow_trajectory
// ("rand", scripts/evaluate_drone.py:81-194) for a batch of reference
// trajectories - see mlp_closed_loop_kernel (mlp_rollout.hip) for the loop; here the
// hidden / cell state is carried through all steps (it is reset once per
// evaluator, evaluate_drone.py:56-58, never on a divergence).
struct LoopArgs {
  const float *traj;  // [L][9][B] (position, euler, velocity) rows
  const float *h0, *c0;  // [8][B]
  float *div;         // [T][B]
  int *steps;         // [B] iterations executed
  float *drone;       // [T+1][12][B] or NULL: states after each step
  float *actions;     // [T][4][B] or NULL
  float *start;       // [T][12][B] or NULL: states the policy saw
  const float *tables;
  QuadConst c;
  int B, L, T, test_time;
  float thresh_div, thresh_stable;
  int learnt;         // LearntDynamics environment (learnt_residual.h)
};

// LEARNT: the environment is a LearntDynamics (a second instantiation, so that the
// analytic loop keeps its registers)
template <bool LEARNT>
__global__ __launch_bounds__(kThreads) void lstm_closed_loop_kernel(LoopArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kFwd16Lds + (LEARNT ? kLearntFloats : 0));
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
  const Planes Ph0(A.h0, kNH, pitchB), Pc0(A.c0, kNH, pitchB);
  const Planes Pdr(A.drone, A.drone ? (T + 1) * 12 : 0, pitchB);
  const Planes Pac(A.actions, A.actions ? T * 4 : 0, pitchB);
  const Planes Pss(A.start, A.start ? T * 12 : 0, pitchB);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vb_u = live ? vb + (hi ? 4u * pitchB : 0u) : kDead;
  // window columns of this half-wave: lower (x, y, z, vx, -), upper (vy, vz,
  // vx, vy, vz) - policy channels 0-3 / 4-8; trajectory columns 6..8 = velocity
  unsigned vcol[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int lo = j < 3 ? j : 6, up = j < 2 ? 7 + j : 4 + j;
    vcol[j] = live ? vb + (unsigned)(hi ? up : lo) * pitchB : kDead;
  }
  float s[12], h[4], cell[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = i < 3 ? Ptr.ld(vb, i * pitchB) : 0.f;  // zero_reset
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h[r] = Ph0.ld(vb_u, r * pitchB);
    cell[r] = Pc0.ld(vb_u, r * pitchB);
  }
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
    // gates and conv on the 16-bit matrix pipe (policy_mfma16.h), as the
    // forward training sweep, nothing saved
    f32x16 g0, g1;
#pragma unroll
    for (int i = 0; i < 16; ++i) g0[i] = L.T(hTbg + i * 2), g1[i] = 0.f;
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = hi ? (8 + j < kNF ? feat[8 + j < kNF ? 8 + j : 0] : 0.f) : feat[j];
      g0 = mma3(L16.A(hA, nF), split8(v), g0);
      const float vh[8] = {h[0], h[1], h[2], h[3], 0.f, 0.f, 0.f, 0.f};
      g1 = mma3(L16.A(hA, nH), split8(vh), g1);
    }
    // lower: position columns relative to the drone; upper: the last three
    // columns are reference velocity minus drone velocity (prepare_data)
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
        for (int i = 0; i < 12; ++i) rv[e * 12 + i] = relu1(cv[i]);
      }
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rv[kb * 8 + j];
        if (kb & 1) g1 = mma3(L16.A(hA, nG + pp * 3 + kb), split8(v), g1);
        else g0 = mma3(L16.A(hA, nG + pp * 3 + kb), split8(v), g0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = sigmoid_fast(g0[r] + g1[r]);
      const float gf = sigmoid_fast(g0[4 + r] + g1[4 + r]);
      const float gg = tanh_fast(g0[8 + r] + g1[8 + r]);
      const float go = sigmoid_fast(g0[12 + r] + g1[12 + r]);
      cell[r] = fmaf(gf, cell[r], gi * gg);
      h[r] = go * tanh_fast(cell[r]);
    }
    float act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) z = fmaf(L.T(hTo + (j * 4 + r) * 2), h[r], z);
      z += other_half(z);
      act[j] = fminf(fmaxf(sigmoidf_(z + L.U(hBo + j)), 0.f), 1.f);  // np.clip
      Pac.st(vrec, (k * 4 + j) * pB, act[j]);
    }
    if (LEARNT) learnt_quad_step(s, act, c, t, lds + kFwd16Lds, hi);
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
// Reverse tables of the training sweep (fp16 split operands): the small fp32
// tables (rTo, rAq content), then 14 transposed A-operand blocks: W_hh^T [kb],
// W_ih^T feature rows [kb], W_ih^T conv rows [32-row block eb of 5][kb]; the
// k-slots are the 32 gate rows in accumulator order (k-block kb = registers
// 8 kb .. 8 kb + 7: row rrow(8 kb + j) + 4 hi).
constexpr int gTo = 0, gAq = 32;                         // floats
constexpr int gA = 512;                                  // bytes
constexpr int mH = 0, mF = 2, mC = 4, mBlocks16 = 14;
constexpr int kBwd16Lds = (gA + mBlocks16 * kBlock16) / 4;  // 7 296 floats = 29 184 B
static_assert(gAq + kNC * 3 <= gA / 4, "LDS map");

__device__ __forceinline__ float bwd16_weight(const ApgLstmPolicy &p, int n, int row, int j,
                                              int hi) {
  const int kb = n & 1, k = rrow(8 * kb + j) + 4 * hi;   // gate row of this slot
  if (n < mF) return row < kNH ? p.w_hh[k * kNH + row] : 0.f;
  if (n < mC) return row < kNF ? p.w_ih[k * kNX + row] : 0.f;
  return p.w_ih[k * kNX + kNF + ((n - mC) >> 1) * 32 + row];
}

__device__ __forceinline__ void pack_bwd16(const PackArgs &A, int tid, int T) {
  const ApgLstmPolicy &p = A.pol;
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  pack_blocks16(dst, gA, mBlocks16, tid, T,
                [&](int n, int row, int j, int hi) { return bwd16_weight(p, n, row, j, hi); });
  for (int idx = tid; idx < 32; idx += T) {
    const int hi = idx & 1, r = (idx >> 1) & 3, j = idx >> 3;
    A.dst[gTo + idx] = p.w_out[j * kNH + r + 4 * hi];
  }
  for (int idx = tid; idx < kNC * 3; idx += T) {
    const int ch = idx / 3, q = idx % 3;
    A.dst[gAq + idx] = p.conv_w[ch * 27 + q * 3] + p.conv_w[ch * 27 + q * 3 + 1] +
                       p.conv_w[ch * 27 + q * 3 + 2];
  }
}
__global__ __launch_bounds__(256) void lstm_pack_bwd16_kernel(PackArgs A) {
  pack_bwd16(A, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
// both table sets in one launch (the first blocks the forward tables)
__global__ __launch_bounds__(256) void lstm_pack_both_kernel(PackArgs F, PackArgs R, int fwd_blocks) {
  if ((int)blockIdx.x < fwd_blocks)
    pack_fwd16(F, blockIdx.x * blockDim.x + threadIdx.x, fwd_blocks * blockDim.x);
  else
    pack_bwd16(R, (blockIdx.x - fwd_blocks) * blockDim.x + threadIdx.x,
               (gridDim.x - fwd_blocks) * blockDim.x);
}

// ------------------------------------------------------------ the step's tail
// Round 6: what followed the weight-gradient products of an
// LSTM training step as SIX launches - conv_ref.weight -= the position part, the
// two copies out of [dW_ih | dW_hh], torch's fused SGD, the loss reduction, and at
// the head of the next step the two table packs - is ONE workgroup here (6 516
// parameters): gradients into their tensors, momentum SGD with torch's rounding
// (scripts/train_base.py:130-150: optim.SGD(lr, momentum = 0.9)), the operand
// tables of the NEXT step's sweeps from the updated parameters, the loss.
struct TailArgs {
  ApgLstmStepTail t;
};
constexpr int kTailThreads = 1024;
__global__ __launch_bounds__(kTailThreads) void lstm_step_tail_kernel(TailArgs A) {
  const ApgLstmStepTail &t = A.t;
  const int tid = threadIdx.x;
  // parameter e of tensor q: gradient source, destination, parameter, momentum
  const int sizes[8] = {kNC * 27, kNC, kNG * kNX, kNG * kNH, kNG, kNG, 4 * kNH, 4};
  float *const par[8] = {t.param.conv_w, t.param.conv_b, t.param.w_ih, t.param.w_hh,
                         t.param.b_ih,   t.param.b_hh,   t.param.w_out, t.param.b_out};
  float *const mom[8] = {t.mom.conv_w, t.mom.conv_b, t.mom.w_ih, t.mom.w_hh,
                         t.mom.b_ih,   t.mom.b_hh,   t.mom.w_out, t.mom.b_out};
  float *const grd[8] = {t.grad.conv_w, t.grad.conv_b, t.grad.w_ih, t.grad.w_hh,
                         t.grad.b_ih,   t.grad.b_hh,   t.grad.w_out, t.grad.b_out};
  // one flat index space over the eight tensors, every load of a thread's (at most
  // seven) elements requested before the first is used: ONE round trip to memory
  // instead of one per tensor (the first build walked the tensors one after the
  // other: 22 us for 6 516 parameters)
  constexpr int kTotal = kNC * 27 + kNC + kNG * kNX + kNG * kNH + kNG + kNG + 4 * kNH + 4;
  constexpr int kPer = (kTotal + kTailThreads - 1) / kTailThreads;
  // the step's parameters as the pack below reads them: this workgroup's LDS (the
  // pack read them back from global memory through ~10 dependent loads per thread:
  // 14 us for the launch, profiles/r06_step_LSTM_timeline.txt)
  __shared__ float sp[kTotal];
  int qs[kPer], es[kPer];
  float g[kPer], m_old[kPer], p_old[kPer];
  const bool pack = t.tables_fwd && t.tables_bwd;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    int e = tid + k * kTailThreads, q = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (q == j && e >= sizes[j]) e -= sizes[j], q = j + 1;
    const bool on = e < sizes[q < 8 ? q : 7] && tid + k * kTailThreads < kTotal;
    qs[k] = on ? q : -1, es[k] = e;
    g[k] = m_old[k] = p_old[k] = 0.f;
    if (!on) continue;
    if (q == 0) {          // conv_ref.weight [20][9][3]: windows - (c < 3) positions
      const int ch = e / 27, c = (e % 27) / 3;
      g[k] = grd[0][e] - (c < 3 ? t.conv_pos[ch * 3 + c] : 0.f);
    } else if (q == 2) {   // lstm.weight_ih out of [dW_ih | dW_hh] [32][183]
      g[k] = t.ih_hh[(e / kNX) * (kNX + kNH) + e % kNX];
    } else if (q == 3) {
      g[k] = t.ih_hh[(e / kNH) * (kNX + kNH) + kNX + e % kNH];
    } else if (q == 5) {   // lstm.bias_hh: the same sums as bias_ih
      g[k] = grd[4][e];
    } else {
      g[k] = grd[q][e];
    }
    if (t.update) m_old[k] = mom[q][e];
    if (t.update || pack) p_old[k] = par[q][e];
  }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = qs[k], e = es[k];
    if (q < 0) continue;
    if (q == 0 || q == 2 || q == 3 || (q == 5 && grd[5] != grd[4])) grd[q][e] = g[k];
    float p_new = p_old[k];
    if (t.update) {        // torch.optim.SGD in double, one rounding each (mlp_common.h)
      const float buf = (float)(t.momentum * (double)m_old[k] + (double)g[k]);
      mom[q][e] = buf;
      par[q][e] = p_new = (float)((double)p_old[k] - t.lr * (double)buf);
    }
    sp[tid + k * kTailThreads] = p_new;
  }
  if (t.loss) {              // fixed-shape sum of the loss partials
    __shared__ double sm[kTailThreads / 64];
    double acc = 0.0;
    for (int k = tid; k < t.n_partials; k += kTailThreads) acc += (double)t.loss_partials[k];
#pragma unroll
    for (int s_ = 32; s_ >= 1; s_ >>= 1) acc += __shfl_xor(acc, s_, 64);
    if ((tid & 63) == 0) sm[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < kTailThreads / 64; ++w) tot += sm[w];
      *t.loss = (float)tot;
      if (t.loss_sum) *t.loss_sum += (float)tot;
    }
  }
  if (pack) {
    __syncthreads();         // every parameter of this step is in `sp`
    PackArgs P;
    const float *q_ = sp;
    const float *at[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) at[j] = q_, q_ += sizes[j];
    P.pol.conv_w = at[0], P.pol.conv_b = at[1], P.pol.w_ih = at[2], P.pol.w_hh = at[3];
    P.pol.b_ih = at[4], P.pol.b_hh = at[5], P.pol.w_out = at[6], P.pol.b_out = at[7];
    P.dst = t.tables_fwd;
    pack_fwd16(P, tid, kTailThreads);
    P.dst = t.tables_bwd;
    pack_bwd16(P, tid, kTailThreads);
  }
}

// `applied` (apg_quad_lstm_wgrads finished the gradients and the update): what is
// left of the tail - the next step's tables on as many workgroups as the pack
// kernels use, the loss on one more
__global__ __launch_bounds__(256) void lstm_tail_pack_kernel(TailArgs A, int fwd_blocks,
                                                             int bwd_blocks) {
  const ApgLstmStepTail &t = A.t;
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < fwd_blocks + bwd_blocks) {
    PackArgs P;
    P.pol.conv_w = t.param.conv_w, P.pol.conv_b = t.param.conv_b;
    P.pol.w_ih = t.param.w_ih, P.pol.w_hh = t.param.w_hh;
    P.pol.b_ih = t.param.b_ih, P.pol.b_hh = t.param.b_hh;
    P.pol.w_out = t.param.w_out, P.pol.b_out = t.param.b_out;
    if (blk < fwd_blocks) {
      P.dst = t.tables_fwd;
      pack_fwd16(P, blk * 256 + tid, fwd_blocks * 256);
    } else {
      P.dst = t.tables_bwd;
      pack_bwd16(P, (blk - fwd_blocks) * 256 + tid, bwd_blocks * 256);
    }
    return;
  }
  // fixed-shape sum of the loss partials (the same tree as lstm_step_tail_kernel's:
  // 1024 strided sums, 64-lane butterflies, 16 wave sums in order)
  __shared__ double sm[kTailThreads / 64];
  double part[kTailThreads / 256];
#pragma unroll
  for (int j = 0; j < kTailThreads / 256; ++j) {
    double acc = 0.0;
    for (int k = tid + 256 * j; k < t.n_partials; k += kTailThreads)
      acc += (double)t.loss_partials[k];
    part[j] = acc;
  }
  // thread `tid + 256 j` of the 1024-thread form is lane tid & 63 of wave
  // (tid >> 6) + 4 j
#pragma unroll
  for (int j = 0; j < kTailThreads / 256; ++j) {
    double acc = part[j];
#pragma unroll
    for (int s_ = 32; s_ >= 1; s_ >>= 1) acc += __shfl_xor(acc, s_, 64);
    if ((tid & 63) == 0) sm[(tid >> 6) + 4 * j] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < kTailThreads / 64; ++w) tot += sm[w];
    *t.loss = (float)tot;
    if (t.loss_sum) *t.loss_sum += (float)tot;
  }
}

struct BwdArgs {
  const float *state0, *states, *actions, *ref;
  const unsigned *mask;
  const float *gates, *hc;
  float *loss_partials;
  float *d_gates;  // [32][N]  dL/d gate pre-activations
  float *d_zout;   // [4][N]   dL/d head pre-activations
  float *d_conv;   // [720][B] conv cotangents, summed along the window diagonals
                   // (see kConvG / kConvP below)
  float *grad_state0, *grad_h0, *grad_c0;
  float *cot_amax;   // [groups of 32 trajectories][2]: largest |d_gates|, |d_zout| (or NULL)
  const float *tables;
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
  // ROWS: ref [n][>= H][ref_cols] of the whole data set, read through `index`
  const long long *index;
  const float *r_ref;
  unsigned bytes_ref;
  int ld_ref;
};

template <bool ROWS>
__global__ __launch_bounds__(kThreads) void lstm_rollout_bwd_kernel(BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kBwd16Lds);
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const bool st_lo = live && hi == 0;
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const QuadConst c = A.c;
  const Planes Ps0(A.state0, 12, pitchB), Pst(A.states, kH * 12, pitchB);
  const Planes Pac(A.actions, kH * 4, pitchB), Prf(A.ref, kH * A.ref_cols, pitchB);
  const Planes Pg(A.gates, kNG, pitchN), Phc(A.hc, 2 * kNH, pitchN);
  const Planes Pmk(A.mask, 5, pitchN), Pdg(A.d_gates, kNG, pitchN);
  const Planes Pdz(A.d_zout, 4, pitchN), Pdc(A.d_conv, kConvPlanes, pitchB);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const Planes Rrf(A.r_ref, 1, ROWS ? A.bytes_ref : 0u);
  const unsigned vr_ref =
      ROWS && live ? (unsigned)A.index[b] * (unsigned)A.ld_ref * 4u : kDead;

  float lam[12], dh[4], dc[4];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) dh[r] = 0.f, dc[r] = 0.f;
  float loss = 0.f;
  float run_g = 0.f, run_z = 0.f;   // largest |d_gates|, |d_zout| of the lane so far
  // sliding diagonal sums of the conv cotangents: dgn[ch][ii] = diagonal
  // tau = k + ii of this half-wave's positions (see kConvP)
  float dgn[kNC][4];
#pragma unroll
  for (int ch = 0; ch < kNC; ++ch)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) dgn[ch][ii] = 0.f;
  const unsigned vg = live ? (unsigned)b * 4u + (hi ? kTau * pitchB : 0u) : kDead;
  const unsigned vb_lo = st_lo ? (unsigned)b * 4u : kDead;

  auto ref_value = [&](int k, int i, unsigned pB) {
    return Prf.ld(vb, (k * A.ref_cols + i) * pB);
  };
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
    if (ROWS) {   // 12 contiguous bytes each of the lane's data-set row
      typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
      typedef float f32x3 __attribute__((ext_vector_type(3)));
      const f32x3 p3 = __builtin_bit_cast(f32x3, (u32x3)__builtin_amdgcn_raw_buffer_load_b96(
          Rrf.rsrc, (int)vr_ref, k * A.ref_cols * 4, 0));
      const f32x3 v3 = __builtin_bit_cast(f32x3, (u32x3)__builtin_amdgcn_raw_buffer_load_b96(
          Rrf.rsrc, (int)vr_ref, (k * A.ref_cols + A.vel_col) * 4, 0));
#pragma unroll
      for (int i = 0; i < 3; ++i) rp[i] = p3[i], rv[i] = v3[i];
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        rp[i] = ref_value(k, i, pB), rv[i] = ref_value(k, A.vel_col + i, pB);
    }
    unsigned mw[5];
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) mw[eb] = Pmk.ldu(vn, eb * pN);
    float gt[16], cp[4];  // activated gates and c_prev of units r + 4 hi
#pragma unroll
    for (int i = 0; i < 16; ++i) gt[i] = Pg.ld(vr, ((i >> 2) * kNH + (i & 3)) * pN);
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[r] = Phc.ld(vr, (kNH + r) * pN);
    __builtin_amdgcn_sched_barrier(0);
    // loss terms of step k and their seeds (drone_loss.py:22-34)
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
    quad_step_adjoint(lam, ga, a[0], &sc[9], c, t);  // lam: dL/ds_k (dynamics)

    // head: a = sigmoid(W_out h' + b_out)
    float dz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dz[j] = ga[j] * a[j] * (1.f - a[j]);
      Pdz.st(vn_lo, j * pN, dz[j]);
      run_z = fmaxf(run_z, fabsf(dz[j]));
    }
    // LSTM cell, lane-local for the units r + 4 hi
    f32x16 dG;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = gt[r], gf = gt[4 + r], gg = gt[8 + r], go = gt[12 + r];
      const float tc = tanh_fast(fmaf(gf, cp[r], gi * gg));
      float dht = dh[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) dht = fmaf(L.T(gTo + (j * 4 + r) * 2), dz[j], dht);
      const float dct = dc[r] + dht * go * (1.f - tc * tc);
      dG[r] = dct * gg * gi * (1.f - gi);
      dG[4 + r] = dct * cp[r] * gf * (1.f - gf);
      dG[8 + r] = dct * gi * (1.f - gg * gg);
      dG[12 + r] = dht * tc * go * (1.f - go);
      dc[r] = dct * gf;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Pdg.st(vr, ((i >> 2) * kNH + (i & 3)) * pN, dG[i]);
    // dL/dh_prev = W_hh^T dG and dL/dfeatures = W_ih[:, :15]^T dG
    // (16-bit matrix pipe, policy_mfma16.h: the gate cotangents scaled per
    // trajectory, two fp16 terms, three products per k-block)
    Op16 xg[2];
    int ex;
    {
      float amax = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) amax = fmaxf(amax, fabsf(dG[i]));
      run_g = fmaxf(run_g, amax);
      ex = scale_exponent(amax);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_ldexpf(dG[8 * kb + j], -ex);
        xg[kb] = split8(v);
      }
    }
    f32x16 yh, yf;
#pragma unroll
    for (int i = 0; i < 16; ++i) yh[i] = 0.f, yf[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      yh = mma3(L16.A(gA, mH + kb), xg[kb], yh);
      yf = mma3(L16.A(gA, mF + kb), xg[kb], yf);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      yh[i] = __builtin_amdgcn_ldexpf(yh[i], ex);
      yf[i] = __builtin_amdgcn_ldexpf(yf[i], ex);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] = yh[r];  // rows r + 4 hi = the lane's units
    float dfeat[kNF], gs[12];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float own = yf[i], oth = other_half(own);
      dfeat[rrow(i)] = hi ? oth : own;
      if (rrow(i) + 4 < kNF) dfeat[rrow(i) + 4 < kNF ? rrow(i) + 4 : 0] = hi ? own : oth;
    }
    quad_features_adjoint(sc, t, dfeat, gs);
#pragma unroll
    for (int i = 0; i < 12; ++i) lam[i] += gs[i];
    // conv outputs: five 32-row blocks over e = ch*8 + pos; only the position
    // columns of the window carry a gradient (rel = ref - pos)
    float dpos[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int eb = 0; eb < 5; ++eb) {
      f32x16 y0;
#pragma unroll
      for (int i = 0; i < 16; ++i) y0[i] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) y0 = mma3(L16.A(gA, mC + eb * 2 + kb), xg[kb], y0);
#pragma unroll
      for (int i = 0; i < 16; ++i) y0[i] = __builtin_amdgcn_ldexpf(y0[i], ex);
      const unsigned mws = hi ? mw[eb] >> 4 : mw[eb];  // bit r(i) + 4 hi -> bit r(i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // registers 4g..4g+3: channel eb*4 + g,
        const int ch = eb * 4 + g;     // positions ii + 4 hi
        float sum = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * g + ii;
          const float dcp = ((mws >> rrow(i)) & 1u) ? y0[i] : 0.f;
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
  if (live && A.grad_h0)
#pragma unroll
    for (int r = 0; r < 4; ++r) A.grad_h0[(size_t)(r + 4 * hi) * B + b] = dh[r];
  if (live && A.grad_c0)
#pragma unroll
    for (int r = 0; r < 4; ++r) A.grad_c0[(size_t)(r + 4 * hi) * B + b] = dc[r];
  write_wave_partial(A.loss_partials, st_lo ? loss : 0.f);
  if (A.cot_amax) {   // what apg_quad_lstm_gate_wgrad scales its fp16 operands by
    float mg = run_g, mz = run_z;
#pragma unroll
    for (int s_ = 32; s_ >= 1; s_ >>= 1) {
      mg = fmaxf(mg, __shfl_xor(mg, s_, 64));
      mz = fmaxf(mz, __shfl_xor(mz, s_, 64));
    }
    const int group = blockIdx.x * (kThreads / 64) + wave;
    if (lane == 0 && group * 32 < B) A.cot_amax[2 * group] = mg, A.cot_amax[2 * group + 1] = mz;
  }
}

// --------------------------------------------------- gate weight gradients
// Round 6.  [dW_ih | dW_hh] = d_gates [x ; h_prev]^T was a
// stream product over 183 planes of H*B columns (480 MB at B = 65 536, 112 us)
// of which the 160 relu(conv) columns are a function of the reference window -
// 90 numbers per column that the forward sweep read from planes a tenth that
// size.  This kernel forms the same sums WITHOUT those columns in memory, and
// the forward sweep no longer writes them (it was bound by exactly these
// stores).  The reduction index of a weight gradient is the trajectory, so the
// products run trajectory-major (policy_tm.h):
//   * conv with the operands of the forward sweep SWAPPED: the window slots of
//     the lane's trajectory as A operand, the conv weight blocks of the forward
//     tables as B - the accumulator then holds channel (lane & 31) of the 16
//     trajectories r(i) + 4 hi in its registers: after bias + relu it IS the
//     B operand (k-slot = trajectory) of the product, no transposition;
//   * d_gates as A operand: gate row (lane & 31), the trajectories as four
//     16-byte loads of that plane (TBlock), scaled by a power of two and split
//     into two fp16 terms.  The scale is the largest |d_gates| of the wave's
//     trajectories over the whole unroll, which the reverse sweep leaves per
//     group of 32 (BwdArgs::cot_amax): ONE exponent per wave, so the
//     accumulators are touched by matrix instructions only.  (A running
//     exponent with a rescale branch made the compiler copy all 160
//     accumulator registers between the two register files every step.)
// A workgroup is 8 waves x 2 groups of 32 trajectories, all ten steps of each;
// the window positions are shared between two workgroups (PH = blockIdx & 1:
// positions 4 PH .. 4 PH + 3: four 32 x 32 accumulators per lane) so that two
// waves per SIMD fit the register file.  The fifth accumulator takes the columns
// that ARE in memory: PH 0 [15 features | h_prev | 1] against d_gates (dW_ih's
// first 15 columns, dW_hh, db), PH 1 [h_new | 1] against d_zout (the head's
// dW_out, db_out - the product that was a launch of its own).  The waves of a
// workgroup put their accumulators side by side in LDS and the workgroup adds
// them in wave order, lstm_gate_wgrad_reduce_kernel sums the workgroups in index
// order: bit-reproducible.
// What bounds it (tools/issue_probe3.hip, tools/ab_gate_wgrad.sh):
// the fp16-split arithmetic - v_cvt_pk_f16_f32, v_fma_mix_f32, integer max, v_ldexp -
// does NOT issue in the gaps of a second wave the way v_fma does: ~1 200 such
// instructions per 32 trajectories and step at 4.3 cycles each; the matrix pipe
// (108 instructions of 32 cycles) hides behind them.  A build with ONE wave per
// SIMD doing all eight positions (d_gates split once, the window in registers,
// matrix instructions spaced by hand with scheduling fences) ran no faster: the
// compiler moves the matrix results between the two register files for the relu
// (tools/patches/lstm_gate_wgrad_one_wave.patch).
constexpr int kGwThreads = 512, kGwWaves = kGwThreads / 64, kGwGroups = 2;
constexpr int kGwBlocks = 5;                      // accumulators per lane: 4 positions + 1
constexpr int kGwPart = kGwBlocks * 16 * 64;      // one workgroup's partial: 5 120 floats
constexpr int kGwTab = (hA + (nC + 2) * kBlock16) / 4;  // forward tables through the conv blocks
constexpr int kGwActs = kNF + 3 * kNH;            // acts planes: 15 | 8 + 8 | 8
#if !defined(APG_EXPERIMENT_BUILD) && defined(APG_GW_KNOCKOUT)
#error "experiment macro in a product build (variants: tools/build_policy_variant.sh)"
#endif
#ifndef APG_GW_KNOCKOUT
#define APG_GW_KNOCKOUT 0   // timing experiments: 2 no conv / product per position,
                            // 8 one step per group
#endif

struct GwArgs {
  const float *state0, *states, *in_ref;
  const float *acts;      // [39][N]: features | h_prev, c_prev | h_new
  const float *d_gates, *d_zout;
  const float *cot_amax;  // [groups][2] (lstm_rollout_bwd_kernel)
  const float *tables;    // forward tables (lstm_pack_fwd16_kernel)
  float *partials;        // [workgroups][kGwPart]
  int B;
};

// the values of trajectories >= nvalid of a trajectory-major block: somebody
// else's columns (TBlock) - they must meet zeros
__device__ __forceinline__ void mask_tail(float (&v)[16], int hi, int nvalid) {
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (rrow(i) + 4 * hi >= nvalid) v[i] = 0.f;
}

// exponent above a (finite, non-negative) maximum; 0 for zero or non-finite
__device__ __forceinline__ int amax_exponent(float m) {
  return m > 0.f && m < __builtin_inff() ? __builtin_amdgcn_frexp_expf(m) : 0;
}

// what a step reads: issued one step ahead, right after the previous step's
// blocks have been split
struct GwLoads {
  TBlock tg, ta, tz;
  float pos[3];
};
// The sliding reference window of a lane (6 rows x 5 columns, raw) lives in LDS,
// [slot of 7][column][wave][lane], written by direct-to-LDS loads: the row that
// slides in for step k + 1 is fetched during step k into the seventh slot and
// costs no registers on the way.  Row k + r (relative to 4 PH) is in slot
// (k + r) % 7.
constexpr int kGwSlots = 7;
constexpr int kGwWin = kGwSlots * 5 * kGwThreads;   // floats
// Trajectory-major blocks (d_gates; PH 0: features + h_prev) reach the registers
// THROUGH LDS: the plane layout makes a block load by lane 64 pieces of 16 bytes in 32
// cache lines (tools/ab_gate_wgrad.sh: 12 us of this kernel); instead 64 lanes fetch
// 16 consecutive bytes each - eight lanes a whole line of one plane - straight into
// LDS (4 instructions, 8 lines each), the chunks of a line swizzled by its plane
// number so that the 16-byte reads in block orientation hit no bank twice.
constexpr int kGwBlk = kGwWaves * 2 * 1024;        // floats: [wave][d_gates | fifth][4 KB]

template <int PH>
__device__ __forceinline__ void gate_wgrad_body(const GwArgs &A) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) float lds[];   // tables | window / sum
  float *const win = lds + kGwTab;
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const LdsView16 L16(lds, lane);
  // this wave's two staged blocks; a lane's share of a staged load: line (plane)
  // lane >> 3 of the instruction's eight, chunk (lane & 7) ^ (that plane & 7)
  float *const blk = lds + kGwTab + kGwWin + wave * 2048;
  const unsigned st_plane = (unsigned)lane >> 3, st_chunk = (((unsigned)lane & 7u) ^ st_plane) * 16u;
  // ... and of a block read: chunk 2 g + hi of plane `row`
  auto staged = [&](int which, TBlock &t) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      t.q[g] = *reinterpret_cast<const u32x4 *>(
          blk + which * 1024 + (row * 8 + ((2 * g + hi) ^ (row & 7))) * 4);
  };
  const int B = A.B;
  const unsigned pitchB = (unsigned)B * 4u, pitchN = pitchB * kH;
  const Planes Ps0(A.state0, 12, pitchB), Pst(A.states, kH * 12, pitchB);
  const Planes Pin(A.in_ref, 2 * kH * kRD, pitchB);
  const Planes Pac(A.acts, kGwActs, pitchN), Pdg(A.d_gates, kNG, pitchN);
  const Planes Pdz(A.d_zout, 4, pitchN);
  // conv bias of THIS lane's channel (the table is in accumulator order)
  const float cb = lds[hTbc + 2 * ((row & 3) + 4 * (row >> 3)) + ((row >> 2) & 1)];
  // the fifth block's B operand: plane of column `row`, or the ones column
  constexpr int kAuxCols = PH == 0 ? kNF + kNH : kNH, kAuxPlane0 = PH == 0 ? 0 : kNF + 2 * kNH;
  const bool aux_row = row < kAuxCols;
  const float a_one = row == kAuxCols ? 1.f : 0.f;

  f32x16 acc[kGwBlocks];
#pragma unroll
  for (int n = 0; n < kGwBlocks; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  // the two waves of a SIMD (w, w + 4) run the same instruction stream: started
  // together they want the matrix pipe and the VALU at the same moments.  The
  // second one starts ~770 cycles late (measured, same box: 104-105.5 us in step,
  // 101-102 with a delay of 256 .. 2 560 cycles; tools/ab_gate_wgrad.sh ps<n>)
#ifndef APG_GW_PHASE_SLEEP
#define APG_GW_PHASE_SLEEP 12
#endif
  if (wave >= 4) __builtin_amdgcn_s_sleep(APG_GW_PHASE_SLEEP);
  // the wave's scales: 2^-E d_gates, 2^-E2 d_zout are at most 1
  const int g0 = ((blockIdx.x >> 1) * kGwWaves + wave) * kGwGroups;
  int E, E2;
  {
    float mg = 0.f, mz = 0.f;
#pragma unroll
    for (int gi = 0; gi < kGwGroups; ++gi)
      if ((g0 + gi) * 32 < B) {
        mg = fmaxf(mg, A.cot_amax[2 * (g0 + gi)]);
        mz = fmaxf(mz, A.cot_amax[2 * (g0 + gi) + 1]);
      }
    E = amax_exponent(mg), E2 = amax_exponent(mz);
  }
  const Op16 wc[2] = {L16.A(hA, nC), L16.A(hA, nC + 1)};   // the conv weights: B operand

#pragma unroll 1
  for (int gi = 0; gi < kGwGroups; ++gi) {
    const int b0 = (g0 + gi) * 32;
    if (b0 >= B) break;   // (wave-uniform; the wave still takes part in the sum below)
    const int nvalid = B - b0;
    const int b = b0 + row;
    const bool live = b < B;
    const unsigned vb = live ? (unsigned)b * 4u : kDead;
    const unsigned vb_u = live ? vb + (hi ? 4u * pitchB : 0u) : kDead;  // window column + 4 hi
    // trajectory-major blocks: the lane's plane, trajectories b0 + 4 hi ..; the WHOLE
    // offset sits in the VGPR, so the buffer's range check covers the ragged tail
    const unsigned tcol = (unsigned)b0 * 4u + (unsigned)hi * 16u;
    const unsigned vg = (unsigned)row * pitchN + tcol;
    const unsigned va = (unsigned)(kAuxPlane0 + (aux_row ? row : 0)) * pitchN + tcol;
    // staged loads: plane (8 i + lane >> 3) of instruction i, this lane's chunk
    const unsigned vs = st_plane * pitchN + (unsigned)b0 * 4u + st_chunk;

    // window row r (relative to 4 PH) -> its slot, straight into LDS
    auto row_in = [&](int r, unsigned pB, int j0, int j1) {
      const int slot = r % kGwSlots;
#pragma unroll
      for (int j = j0; j < j1; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            Pin.rsrc, (lds_ptr_t)(win + ((slot * 5 + j) * kGwWaves + wave) * 64), 4, (int)vb_u,
            (int)(((r + 4 * PH) * kRD + j) * pB), 0, 0);
    };
    GwLoads ld;
    // the loads of step k in four parts: a block load is 64 cache-line accesses
    // (16 bytes of a line per lane) and holds the wave at issue while the
    // workgroup's others queue behind it - spread over the step, between the
    // positions, the address unit works while the matrix pipe and the VALU do
    auto issue = [&](int k, int part) {
      const unsigned pB = opaque(pitchB), pN8 = 8u * opaque(pitchN);
      const unsigned kcol = (unsigned)k * pB;
      // d_gates: planes 8 part .. + 7 into this wave's first staged block
      __builtin_amdgcn_raw_ptr_buffer_load_lds(Pdg.rsrc, (lds_ptr_t)(blk + part * 256), 16,
                                               (int)(vs + kcol), (int)(part * pN8), 0, 0);
      if (PH == 0) {   // features, h_prev: planes 0 .. 22
        const bool on = 8 * part + (int)st_plane < kAuxCols;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Pac.rsrc, (lds_ptr_t)(blk + 1024 + part * 256),
                                                 16, (int)(on ? vs + kcol : kDead),
                                                 (int)(part * pN8), 0, 0);
      } else {         // h_new, d_zout: a few planes - by lane
        const unsigned o = 32u * (unsigned)part;
        ld.ta.q[part] = __builtin_amdgcn_raw_buffer_load_b128(
            Pac.rsrc, (int)(aux_row ? va + kcol : kDead), (int)o, APG_PLANES_LD_AUX);
        ld.tz.q[part] = __builtin_amdgcn_raw_buffer_load_b128(
            Pdz.rsrc, (int)(row < 4 ? vg + kcol : kDead), (int)o, APG_PLANES_LD_AUX);
      }
      if (part == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)   // the current position: the state BEFORE step k
          ld.pos[j] = k > 0 ? Pst.ld(vb, ((k - 1) * 12 + j) * pB) : Ps0.ld(vb, j * pB);
      } else if (k > 0) {   // the window row that slides in: columns 0-1 | 2-3 | 4
        row_in(k + 5, pB, 2 * (part - 1), part == 3 ? 5 : 2 * part);
      }
    };
    for (int r = 0; r < 6; ++r) row_in(r, pitchB, 0, 5);
#pragma unroll
    for (int part = 0; part < 4; ++part) issue(0, part);

    int s0 = 0;   // k % 7
#pragma unroll 1
    for (int k = 0; k < ((APG_GW_KNOCKOUT & 8) ? 1 : kH); ++k) {
      // everything issued for this step has landed (registers and LDS)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      typedef const __attribute__((address_space(3))) float *win_ptr_t;
      unsigned wo[6];   // LDS byte offset of window row r, column 0, this lane
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int slot = s0 + r >= kGwSlots ? s0 + r - kGwSlots : s0 + r;
        wo[r] = (unsigned)(kGwTab + (slot * 5 * kGwWaves + wave) * 64 + lane) * 4u;
      }
      s0 = s0 + 1 == kGwSlots ? 0 : s0 + 1;
      const float sub[3] = {hi ? 0.f : ld.pos[0], hi ? 0.f : ld.pos[1], hi ? 0.f : ld.pos[2]};
      // the step's blocks out of the load registers, the next step's loads into
      // them (a whole step ahead of their use), then: d_gates scaled and split,
      // the fifth block's operands
      Op16 ad[2], bx[2], az[2];
      {
        float dv[16], av[16], zv[16];
        staged(0, ld.tg);
        if (PH == 0) staged(1, ld.ta);
        ld.tg.get(dv);
        ld.ta.get(av);
        if (PH == 1) ld.tz.get(zv);
        // (the staged blocks are in registers: the next step's may land)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (k + 1 < kH) issue(k + 1, 0);
        if (nvalid < 32) mask_tail(dv, hi, nvalid);
        split16(dv, E, ad);
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] += a_one;   // (the ones column's plane reads 0)
        split16(av, 0, bx);
        if (PH == 1) {
          if (nvalid < 32) mask_tail(zv, hi, nvalid);
          split16(zv, E2, az);
        }
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc[4] = mma3(PH == 0 ? ad[kk] : az[kk], bx[kk], acc[4]);

      // conv of position 4 PH + e4 with the operands swapped: window slot s =
      // (column s / 3, tap s % 3) of the lane's trajectory, relative to the
      // current position, split in pairs straight into the operand registers
      auto conv = [&](int e4) {
        // the 15 window values of the position in one batch of LDS reads; the
        // row offsets are made opaque per position: a value shared by three
        // positions is read three times instead of being kept in a register
        float wv[15];
        {
          unsigned o[3] = {wo[e4], wo[e4 + 1], wo[e4 + 2]};
          asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]));
#pragma unroll
          for (int s_ = 0; s_ < 15; ++s_)
            wv[s_] = *(win_ptr_t)((const __attribute__((address_space(3))) char *)(lds_ptr_t)lds +
                                  (o[s_ % 3] + (s_ / 3) * kGwThreads * 4));
        }
        f32x16 cv;
#pragma unroll
        for (int i = 0; i < 16; ++i) cv[i] = cb;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          Op16 x;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int s0_ = kb * 8 + 2 * q, s1 = s0_ + 1;
            const float v0 = s0_ / 3 < 3 ? wv[s0_] - sub[s0_ / 3] : wv[s0_];
            float v1 = 0.f;
            if (s1 < 15) v1 = s1 / 3 < 3 ? wv[s1] - sub[s1 / 3] : wv[s1];
            unsigned h, l;
            split_pair(v0, v1, h, l);
            x.h[q] = h, x.l[q] = l;
          }
          cv = mma3(x, wc[kb], cv);   // [trajectory][channel]
        }
        return cv;
      };
#pragma unroll
      for (int e4 = 0; e4 < ((APG_GW_KNOCKOUT & 2) ? 0 : 4); ++e4) {
        const f32x16 cv = conv(e4);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = relu1(cv[8 * kk + j]);
          acc[e4] = mma3(ad[kk], split8(v), acc[e4]);
        }
        if (e4 < 3 && k + 1 < kH) issue(k + 1, e4 + 1);
      }
    }
  }
  // the workgroup's sum at true scale, two accumulator blocks at a time: every
  // wave puts them into its own LDS region (16-byte writes; the window is done
  // with), then the 512 threads add the eight regions in wave order and write the
  // partial
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 *const sum = reinterpret_cast<f32x4 *>(win);
  f32x4 *dst = reinterpret_cast<f32x4 *>(A.partials + (size_t)blockIdx.x * kGwPart);
  constexpr int kRound = 2, kQuads = kRound * 4 * 64;   // blocks / float4 per wave and round
  static_assert(kGwWaves * kQuads * 4 <= kGwWin, "the sum regions fit the window's LDS");
#pragma unroll
  for (int n0 = 0; n0 < kGwBlocks; n0 += kRound) {
    __syncthreads();   // (the window / the previous round have been read)
#pragma unroll
    for (int n = n0; n < n0 + kRound && n < kGwBlocks; ++n)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int e = (n == 4 && PH == 1) ? E2 : E;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_ldexpf(acc[n][4 * i4 + c], e);
        sum[wave * kQuads + ((n - n0) * 4 + i4) * 64 + lane] = v;
      }
    __syncthreads();
    const int quads = (n0 + kRound <= kGwBlocks ? kRound : kGwBlocks - n0) * 4 * 64;
    for (int idx = threadIdx.x; idx < quads; idx += kGwThreads) {
      f32x4 v = sum[idx];
#pragma unroll
      for (int wv = 1; wv < kGwWaves; ++wv) v += sum[wv * kQuads + idx];
      dst[n0 * 4 * 64 + idx] = v;
    }
  }
}

__global__ __launch_bounds__(kGwThreads) void lstm_gate_wgrad_kernel(GwArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kGwTab);
  if (blockIdx.x & 1) gate_wgrad_body<1>(A);
  else gate_wgrad_body<0>(A);
}

// The owner thread of a gradient element (the one that holds the final sum in
// lstm_wgrad_reduce_kernel) can finish the step for it: the gradient into its
// tensor and torch's momentum SGD (double, one rounding each: mlp_common.h) - what
// lstm_step_tail_kernel did on ONE workgroup behind the sums (`applied`).
struct FuseArgs {
  int on, update;
  double lr, momentum;
  ApgLstmPolicyGrads grad, param, mom;
};
__device__ __forceinline__ void fuse_elem(const FuseArgs &U, float *grad, float *par,
                                          float *mom, int e, float g) {
  grad[e] = g;
  if (!U.update) return;
  const float buf = (float)(U.momentum * (double)mom[e] + (double)g);
  mom[e] = buf;
  par[e] = (float)((double)par[e] - U.lr * (double)buf);
}

// sum of the workgroups' partials (index order) into the gradients: thread
// (part, o) adds every 8th workgroup of output o's parity, part 0 adds the eight
struct GwReduceArgs {
  const float *partials;
  int chunks;            // workgroup pairs
  float *ih_hh;          // [32][183]
  float *b_ih;           // [32]
  float *w_out, *b_out;  // [4][8], [4]
};
template <bool FUSE>
__device__ __forceinline__ void gate_wgrad_reduce_body(const GwReduceArgs &A, int block,
                                                       const FuseArgs &U) {
  __shared__ float part[8][32];
  const int ol = threadIdx.x & 31, pt = threadIdx.x >> 5;
  const int o = block * 32 + ol;               // < 2 kGwPart
  const int ph = o / kGwPart, r = o - ph * kGwPart;
  float s = 0.f;
  for (int c = pt; c < A.chunks; c += 8) s += A.partials[(size_t)(2 * c + ph) * kGwPart + r];
  part[pt][ol] = s;
  __syncthreads();
  if (pt) return;
#pragma unroll
  for (int q = 1; q < 8; ++q) s += part[q][ol];
  // element r of a partial: block n, register 4 i4 + c of lane `lane`
  const int n = r >> 10, i = 4 * ((r >> 8) & 3) + (r & 3), lane = (r >> 2) & 63;
  const int g = rrow(i) + 4 * (lane >> 5), col = lane & 31;
  const bool fu = FUSE && U.on;
  if (n < 4) {
    if (col < kNC) {
      const int x = kNF + col * kNP + 4 * ph + n;
      A.ih_hh[g * (kNX + kNH) + x] = s;
      if (fu) fuse_elem(U, U.grad.w_ih, U.param.w_ih, U.mom.w_ih, g * kNX + x, s);
    }
  } else if (ph == 0) {
    if (col < kNF) {
      A.ih_hh[g * (kNX + kNH) + col] = s;
      if (fu) fuse_elem(U, U.grad.w_ih, U.param.w_ih, U.mom.w_ih, g * kNX + col, s);
    } else if (col < kNF + kNH) {
      A.ih_hh[g * (kNX + kNH) + kNX + col - kNF] = s;
      if (fu)
        fuse_elem(U, U.grad.w_hh, U.param.w_hh, U.mom.w_hh, g * kNH + col - kNF, s);
    } else if (col == kNF + kNH) {
      A.b_ih[g] = s;
      if (fu) {   // lstm.bias_hh: the same sums as bias_ih (its gradient may alias it)
        fuse_elem(U, U.grad.b_ih, U.param.b_ih, U.mom.b_ih, g, s);
        fuse_elem(U, U.grad.b_hh, U.param.b_hh, U.mom.b_hh, g, s);
      }
    }
  } else if (g < 4) {
    if (col < kNH) {
      A.w_out[g * kNH + col] = s;
      if (fu) fuse_elem(U, U.grad.w_out, U.param.w_out, U.mom.w_out, g * kNH + col, s);
    } else if (col == kNH) {
      A.b_out[g] = s;
      if (fu) fuse_elem(U, U.grad.b_out, U.param.b_out, U.mom.b_out, g, s);
    }
  }
}
constexpr int kGwReduceBlocks = 2 * kGwPart / 32;
__global__ __launch_bounds__(256) void lstm_gate_wgrad_reduce_kernel(GwReduceArgs A) {
  gate_wgrad_reduce_body<false>(A, blockIdx.x, FuseArgs{});
}

// --------------------------------------------------- conv weight gradient
// Round 6.  dconv_w[ch][c][t] = sum G[ch][hi][tau] . R[4 hi + tau + t][c]
//                               - (c < 3) sum_k P[ch][k] . pos_k[c],
// dconv_b[ch] = sum_k P[ch][k]  (the diagonal sums the reverse sweep leaves, see
// kConvP) were two segmented planes_gemm products + their reduce: 75-80 us for
// 190 MB of cotangents, the LDS-tile kernel on the fp32 matrix instruction.  Here:
// one wave per 32 trajectories walks the 26 diagonals and the 10 position sums;
// both operands of a segment are 32 x 32 blocks straight out of planes - G / P
// rows as A (channel in the lane), window rows / positions as B ((column, tap) in
// the lane: plane (4 hi + tau + t) 9 + c) - fetched by coalesced direct-to-LDS loads
// one segment ahead (chunks swizzled by plane, see kGwBlk), split into fp16 terms
// (the cotangents scaled by a running power of two per wave: two accumulators to
// rescale when it grows) and multiplied trajectory-major.  The waves of a workgroup
// add up in LDS in wave order, lstm_conv_wgrad_reduce_kernel adds the workgroups in
// index order: bit-reproducible.
constexpr int kCwThreads = 512, kCwWaves = kCwThreads / 64;
constexpr int kCwSegG = 2 * kTau, kCwSeg = kCwSegG + kH;   // 26 diagonals + 10 position sums
constexpr int kCwPart = 2 * 16 * 64;                       // a workgroup's partial: 2 048 floats
constexpr int kCwLds = kCwWaves * 4 * 1024;                // floats: [wave][2 buffers][A | B][4 KB]
constexpr int kCwTaps = kRD * 3;                           // 27 (column, tap) pairs

struct CwArgs {
  const float *d_conv, *in_ref, *st_all;   // [720][B], [2H*9][B], [(H+1)*12][B]
  float *partials;                         // [workgroups][kCwPart]
  int B;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
  return o > v ? o : v;
}
// maximum of a wave's unsigned values on the VALU alone (DPP row shifts, then row
// broadcasts; policy_tm.h's wave_umax is six ds_bpermute round trips)
__device__ __forceinline__ unsigned wave_umax_dpp(unsigned v) {
  v = dpp_umax<0x111, 0xf>(v);   // row_shr:1
  v = dpp_umax<0x112, 0xf>(v);   // row_shr:2
  v = dpp_umax<0x114, 0xf>(v);   // row_shr:4
  v = dpp_umax<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of each row: the row's maximum
  v = dpp_umax<0x142, 0xa>(v);   // row_bcast:15 -> lanes 31, 63: two rows
  v = dpp_umax<0x143, 0xc>(v);   // row_bcast:31 -> lane 63: the wave
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(kCwThreads) void lstm_conv_wgrad_kernel(CwArgs A) {
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, hi = lane >> 5, row = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int B = A.B, b0 = (blockIdx.x * kCwWaves + wave) * 32;
  const unsigned pitchB = (unsigned)B * 4u;
  const Planes Pdc(A.d_conv, kConvPlanes, pitchB), Pin(A.in_ref, 2 * kH * kRD, pitchB);
  const Planes Pst(A.st_all, (kH + 1) * 12, pitchB);
  float *const buf = lds + wave * 4096;   // [buffer][A | B][1024]
  // a lane's share of a staged load: line (plane) 8 i + lane >> 3 of instruction i,
  // chunk (lane & 7) ^ (that plane's number & 7)
  const unsigned sp = (unsigned)lane >> 3, sc = (((unsigned)lane & 7u) ^ sp) * 16u;
  const unsigned col0 = (unsigned)b0 * 4u + sc;
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  int E[2] = {-100000, -100000};

  if (b0 < B) {   // (wave-uniform; a dead wave still takes part in the sum below)
    const int nvalid = B - b0;
    // the two operands of segment s into buffer q (8 instructions)
    auto issue = [&](int s, int q) {
      const unsigned pB = opaque(pitchB);
      const bool diag = s < kCwSegG;
      const int shi = s >= kTau ? 1 : 0, tau = s - shi * kTau, k = s - kCwSegG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned ch = 8u * i + sp;   // A: channel rows
        const unsigned pa = diag ? ch * (unsigned)kCwSegG + (unsigned)s
                                 : (unsigned)kConvP + ch * (unsigned)kH + (unsigned)k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            Pdc.rsrc, (lds_ptr_t)(buf + q * 2048 + i * 256), 16,
            (int)(ch < (unsigned)kNC ? pa * pB + col0 : kDead), 0, 0, 0);
        const unsigned j = ch;             // B: (column, tap) = (j / 3, j % 3) | position j
        if (diag) {
          const unsigned pb = ((unsigned)(4 * shi + tau) + j % 3u) * (unsigned)kRD + j / 3u;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              Pin.rsrc, (lds_ptr_t)(buf + q * 2048 + 1024 + i * 256), 16,
              (int)(j < (unsigned)kCwTaps ? pb * pB + col0 : kDead), 0, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              Pst.rsrc, (lds_ptr_t)(buf + q * 2048 + 1024 + i * 256), 16,
              (int)(j < 3u ? ((unsigned)k * 12u + j) * pB + col0 : kDead), 0, 0, 0);
        }
      }
    };
    auto staged = [&](int q, int which, float (&v)[16]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_ f = *reinterpret_cast<const f32x4_ *>(
            buf + q * 2048 + which * 1024 + (row * 8 + ((2 * g + hi) ^ (row & 7))) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * g + c] = f[c];
      }
    };
    // one segment: wait for its operands (the next one's are in flight), multiply
    auto segment = [&](int s, f32x16 &acc_, int &E_, bool ones) {
      const int q = s & 1;
      if (s + 1 < kCwSeg) {
        issue(s + 1, q ^ 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this segment's eight have landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      float av[16], bv[16];
      staged(q, 0, av);
      staged(q, 1, bv);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (before the buffer is refilled)
      if (nvalid < 32) mask_tail(av, hi, nvalid);
      // the cotangents' block exponent; the accumulator follows when it grows
      unsigned m = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) m = umax_abs(m, av[i]);
      m = wave_umax_dpp(m);
      bool bad = false;
      const int e = bits_exp(m, bad, false);
      if (e > E_) {
        if (E_ > -100000) {
          const int d = E_ - e;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_[i] = __builtin_amdgcn_ldexpf(acc_[i], d);
        }
        E_ = e;
      }
      if (ones)   // the ones column of the position block: the bias gradient
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] += row == 3 ? 1.f : 0.f;
      Op16 ad[2], bx[2];
      split16(av, E_, ad);
      split16(bv, 0, bx);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc_ = mma3(ad[kk], bx[kk], acc_);
    };
    issue(0, 0);
#pragma unroll 1
    for (int s = 0; s < kCwSegG; ++s) segment(s, acc[0], E[0], false);
#pragma unroll 1
    for (int s = kCwSegG; s < kCwSeg; ++s) segment(s, acc[1], E[1], true);
  }
  // the workgroup's sum at true scale: every wave into its own region, then all
  // threads add the eight regions in wave order
  f32x4_ *const sum = reinterpret_cast<f32x4_ *>(lds);
  constexpr int kQuads = 2 * 4 * 64;
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      f32x4_ v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_ldexpf(acc[n][4 * i4 + c], E[n]);
      sum[wave * kQuads + (n * 4 + i4) * 64 + lane] = v;
    }
  __syncthreads();
  f32x4_ *dst = reinterpret_cast<f32x4_ *>(A.partials + (size_t)blockIdx.x * kCwPart);
  for (int idx = threadIdx.x; idx < kQuads; idx += kCwThreads) {
    f32x4_ v = sum[idx];
#pragma unroll
    for (int wv = 1; wv < kCwWaves; ++wv) v += sum[wv * kQuads + idx];
    dst[idx] = v;
  }
}

struct CwReduceArgs {
  const float *partials;
  int wgs;
  float *conv_w, *conv_pos, *conv_b;   // [20][27], [20][3], [20]
};
// 620 of a partial's 2 048 floats are gradients (20 channels x 27 + 20 x 4): a thread
// per (gradient, 1 of 32 slices of the workgroups), the slices added in a fixed tree
constexpr int kCwOut = kNC * kCwTaps + kNC * 4;
template <bool FUSE>
__device__ __forceinline__ void conv_wgrad_reduce_body(const CwReduceArgs &A, int block,
                                                       const FuseArgs &U) {
  const int pt = threadIdx.x & 31, o = block * 8 + (threadIdx.x >> 5);
  const bool on = o < kCwOut;
  const int oc = on ? o : 0;
  const int n = oc < kNC * kCwTaps ? 0 : 1, q = n ? oc - kNC * kCwTaps : oc;
  const int ch = n ? q >> 2 : q / kCwTaps, col = n ? q & 3 : q % kCwTaps;
  // element of a partial: block n, register i of lane `lane` with ch = r(i) + 4 hi
  const int hi = (ch >> 2) & 1, i = (ch & 3) + 4 * (ch >> 3), lane = hi * 32 + col;
  const int r = ((n * 4 + (i >> 2)) * 64 + lane) * 4 + (i & 3);
  // finishing the step here: a window weight of a position column (c < 3) also adds
  // up its own copy of the position sum it is reduced by - the same slices in the
  // same tree as that sum's owner, i.e. the same float
  const bool fu = FUSE && U.on, pos = fu && n == 0 && col < 9;
  const int r2 = ((4 + (i >> 2)) * 64 + hi * 32 + col / 3) * 4 + (i & 3);
  float s = 0.f, s2 = 0.f;
  if (on)
    for (int c = pt; c < A.wgs; c += 32) {
      s += A.partials[(size_t)c * kCwPart + r];
      if (pos) s2 += A.partials[(size_t)c * kCwPart + r2];
    }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {   // (fixed order)
    s += __shfl_xor(s, d, 32);
    s2 += __shfl_xor(s2, d, 32);
  }
  if (!on || pt) return;
  if (n == 0) {
    if (fu)   // (grad.conv_w may be A.conv_w: the final value is the one that stays)
      fuse_elem(U, U.grad.conv_w, U.param.conv_w, U.mom.conv_w, ch * kCwTaps + col,
                pos ? s - s2 : s);
    else
      A.conv_w[ch * kCwTaps + col] = s;
  } else if (col < 3) {
    A.conv_pos[ch * 3 + col] = s;
  } else {
    A.conv_b[ch] = s;
    if (fu) fuse_elem(U, U.grad.conv_b, U.param.conv_b, U.mom.conv_b, ch, s);
  }
}
constexpr int kCwReduceBlocks = (kCwOut + 7) / 8;
__global__ __launch_bounds__(256) void lstm_conv_wgrad_reduce_kernel(CwReduceArgs A) {
  conv_wgrad_reduce_body<false>(A, blockIdx.x, FuseArgs{});
}
// both sums in one launch (apg_quad_lstm_wgrads: the two product kernels run back to
// back, their partials are added side by side)
__global__ __launch_bounds__(256) void lstm_wgrad_reduce_kernel(GwReduceArgs G, CwReduceArgs C,
                                                                FuseArgs U) {
  if ((int)blockIdx.x < kGwReduceBlocks) gate_wgrad_reduce_body<true>(G, blockIdx.x, U);
  else conv_wgrad_reduce_body<true>(C, blockIdx.x - kGwReduceBlocks, U);
}

// pol NULL: the caller holds packed tables instead of the parameters
int check_lstm(const ApgQuadParams *params, const ApgLstmPolicy *pol, int B, int H,
               bool packed = false) {
  if (!params || (!pol && !packed)) { set_error("params / policy is NULL"); return APG_ERR_ARG; }
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if ((long long)B * kH * 4 * kNX >= (1ll << 32) - 64) {
    set_error("B too large for 32-bit plane offsets (max %d); split the batch",
              (int)(((1ll << 32) - 64) / (kH * 4 * kNX)));
    return APG_ERR_ARG;
  }
  if (H != kH) {
    set_error("the fused LSTM rollout is built for horizon %d (got %d)", kH, H);
    return APG_ERR_ARG;
  }
  if (pol && (!pol->conv_w || !pol->conv_b || !pol->w_ih || !pol->w_hh || !pol->b_ih ||
              !pol->b_hh || !pol->w_out || !pol->b_out)) {
    set_error("policy weight pointer is NULL");
    return APG_ERR_ARG;
  }
  return APG_OK;
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_quad_lstm_workspace_floats(void) {
  // (+ the packed LearntDynamics weights of the closed-loop evaluation)
  return (kFwd16Lds > kBwd16Lds ? kFwd16Lds : kBwd16Lds) + kLearntFloats;
}

int apg_quad_lstm_loss_partials_count(int B) {
  return B <= 0 ? 0 : ((B + kTrajPerBlock - 1) / kTrajPerBlock) * (kThreads / kWave);
}

// `policy` given: its tables are packed into `workspace` first; NULL: `workspace`
// holds them already (apg_quad_lstm_pack_tables / apg_quad_lstm_step_tail)
// rows given: the `_rows` entry points - state0 / in_ref (forward) and ref
// (reverse) are read through rows->index; `state0`, `in_ref` are then OUTPUTS
static int check_rows(const ApgBatchRows *rows, int B, int H, int ref_cols, bool reverse) {
  if (B > 0 && (!rows->index || (reverse ? !rows->ref : (!rows->state0 || !rows->in_ref)))) {
    set_error("rows: NULL index / tensor");
    return APG_ERR_ARG;
  }
  const long long need = reverse ? (long long)H * ref_cols : 2ll * H * kRD;
  const long long ld = reverse ? rows->ld_ref : rows->ld_in_ref;
  if (rows->n_rows < 1 || ld < need || (!reverse && rows->ld_state0 < 12)) {
    set_error("rows: need n_rows >= 1 and row strides of at least 12 / 2H x 9 / H x ref_cols");
    return APG_ERR_ARG;
  }
  if (rows->n_rows * ld * 4 >= (1ll << 32) - 64 ||
      (!reverse && rows->n_rows * (long long)rows->ld_state0 * 4 >= (1ll << 32) - 64)) {
    set_error("rows: the data set's tensors must stay below 4 GiB (32-bit byte offsets)");
    return APG_ERR_ARG;
  }
  return APG_OK;
}

static int lstm_fwd(const ApgBatchRows *rows, const float *state0, const float *in_ref,
                    const float *h0,
                    const float *c0, float dt, const ApgQuadParams *params,
                    const ApgLstmPolicy *policy, int B, int H, float *states,
                    float *actions, float *x, float *gates, float *hc, float *hnew,
                    unsigned *relu_mask, float *workspace, apg_stream_t stream,
                    bool legacy_inplace_ref = false) {
  if (int e = check_lstm(params, policy, B, H, true)) return e;
  if (rows)
    if (int e = check_rows(rows, B, H, 0, false)) return e;
  if (B == 0) return APG_OK;
  if (!state0 || !in_ref || !h0 || !c0 || !states || !actions || !x || !gates ||
      !hc || !hnew || !relu_mask || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  FwdArgs A;
  A.state0 = state0, A.in_ref = in_ref, A.h0 = h0, A.c0 = c0;
  A.states = states, A.actions = actions, A.x = x, A.gates = gates, A.hc = hc;
  A.hnew = hnew;
  A.mask = relu_mask;
  A.tables = workspace;
  A.c = make_const(*params, dt);
  A.B = B;
  A.index = nullptr, A.r_state0 = A.r_in_ref = nullptr;
  A.bytes_state0 = A.bytes_in_ref = 0u, A.ld_state0 = A.ld_in_ref = 0;
  if (rows) {
    A.index = rows->index, A.r_state0 = rows->state0, A.r_in_ref = rows->in_ref;
    A.ld_state0 = rows->ld_state0, A.ld_in_ref = rows->ld_in_ref;
    A.bytes_state0 = (unsigned)(rows->n_rows * (long long)rows->ld_state0 * 4);
    A.bytes_in_ref = (unsigned)(rows->n_rows * (long long)rows->ld_in_ref * 4);
  }
  hipStream_t st = (hipStream_t)stream;
  if (policy) {
    PackArgs P;
    P.pol = *policy, P.dst = workspace;
    hipLaunchKernelGGL(lstm_pack_fwd16_kernel, dim3((kFwd16Lds + 255) / 256), dim3(256),
                       0, st, P);
  }
  const dim3 grid((B + kTrajPerBlock - 1) / kTrajPerBlock);
  if (legacy_inplace_ref)
    hipLaunchKernelGGL((lstm_rollout_fwd_kernel<false, true>), grid, dim3(kThreads),
                       kFwd16Lds * sizeof(float), st, A);
  else if (rows)
    hipLaunchKernelGGL(lstm_rollout_fwd_kernel<true>, grid, dim3(kThreads),
                       kFwd16Lds * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(lstm_rollout_fwd_kernel<false>, grid, dim3(kThreads),
                       kFwd16Lds * sizeof(float), st, A);
  return check_launch("quad_lstm_rollout_fwd");
}

int apg_quad_lstm_rollout_fwd(const float *state0, const float *in_ref,
                              const float *h0, const float *c0, float dt,
                              const ApgQuadParams *params,
                              const ApgLstmPolicy *policy, int B, int H,
                              float *states, float *actions, float *x,
                              float *gates, float *hc, float *hnew,
                              unsigned *relu_mask, float *workspace,
                              apg_stream_t stream) {
  if (!policy) { set_error("policy is NULL"); return APG_ERR_ARG; }
  return lstm_fwd(nullptr, state0, in_ref, h0, c0, dt, params, policy, B, H, states, actions, x,
                  gates,
                  hc, hnew, relu_mask, workspace, stream);
}

int apg_quad_lstm_rollout_fwd_inplace_ref(const float *state0, const float *in_ref,
                                          const float *h0, const float *c0, float dt,
                                          const ApgQuadParams *params,
                                          const ApgLstmPolicy *policy, int B, int H,
                                          float *states, float *actions, float *x,
                                          float *gates, float *hc, float *hnew,
                                          unsigned *relu_mask, float *workspace,
                                          apg_stream_t stream) {
  if (!policy) { set_error("policy is NULL"); return APG_ERR_ARG; }
  return lstm_fwd(nullptr, state0, in_ref, h0, c0, dt, params, policy, B, H, states, actions, x,
                  gates, hc, hnew, relu_mask, workspace, stream, true);
}

int apg_quad_lstm_rollout_fwd_packed(const float *state0, const float *in_ref,
                                     const float *h0, const float *c0, float dt,
                                     const ApgQuadParams *params, const float *tables_fwd,
                                     int B, int H, float *states, float *actions, float *x,
                                     float *gates, float *hc, float *hnew,
                                     unsigned *relu_mask, apg_stream_t stream) {
  return lstm_fwd(nullptr, state0, in_ref, h0, c0, dt, params, nullptr, B, H, states, actions, x,
                  gates,
                  hc, hnew, relu_mask, const_cast<float *>(tables_fwd), stream);
}

static int lstm_bwd(const ApgBatchRows *rows, const float *state0, const float *states,
                    const float *actions,
                    const float *ref, int ref_cols, const unsigned *relu_mask,
                    const float *gates, const float *hc, float dt,
                    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
                    const ApgLstmPolicy *policy, int B, int H, float *loss_partials,
                    float *loss, float *d_gates, float *d_zout, float *d_conv,
                    float *grad_state0, float *grad_h0, float *grad_c0, float *cot_amax,
                    float *workspace, apg_stream_t stream) {
  if (int e = check_lstm(params, policy, B, H, true)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 or 6");
    return APG_ERR_ARG;
  }
  if (rows)
    if (int e = check_rows(rows, B, H, ref_cols, true)) return e;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!state0 || !states || !actions || (!ref && !rows) || !relu_mask || !gates || !hc ||
      !loss_partials || !d_gates || !d_zout || !d_conv || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  BwdArgs A;
  A.state0 = state0, A.states = states, A.actions = actions, A.ref = ref;
  A.mask = relu_mask, A.gates = gates, A.hc = hc;
  A.loss_partials = loss_partials, A.d_gates = d_gates, A.d_zout = d_zout;
  A.d_conv = d_conv, A.grad_state0 = grad_state0, A.grad_h0 = grad_h0;
  A.grad_c0 = grad_c0;
  A.cot_amax = cot_amax;
  A.tables = workspace;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  A.index = nullptr, A.r_ref = nullptr, A.bytes_ref = 0u, A.ld_ref = 0;
  if (rows) {
    A.index = rows->index, A.r_ref = rows->ref, A.ld_ref = rows->ld_ref;
    A.bytes_ref = (unsigned)(rows->n_rows * (long long)rows->ld_ref * 4);
  }
  if (policy) {
    PackArgs P;
    P.pol = *policy, P.dst = workspace;
    hipLaunchKernelGGL(lstm_pack_bwd16_kernel, dim3((kBwd16Lds + 255) / 256), dim3(256),
                       0, st, P);
  }
  const int blocks = (B + kTrajPerBlock - 1) / kTrajPerBlock;
  if (rows)
    hipLaunchKernelGGL(lstm_rollout_bwd_kernel<true>, dim3(blocks), dim3(kThreads),
                       kBwd16Lds * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(lstm_rollout_bwd_kernel<false>, dim3(blocks), dim3(kThreads),
                       kBwd16Lds * sizeof(float), st, A);
  if (int e = check_launch("quad_lstm_rollout_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kThreads / kWave), loss, st);
  return APG_OK;
}

int apg_quad_lstm_rollout_bwd(const float *state0, const float *states,
                              const float *actions, const float *ref,
                              int ref_cols, const unsigned *relu_mask,
                              const float *gates, const float *hc, float dt,
                              const ApgQuadParams *params,
                              const ApgQuadLossWeights *weights,
                              const ApgLstmPolicy *policy, int B, int H,
                              float *loss_partials, float *loss, float *d_gates,
                              float *d_zout, float *d_conv, float *grad_state0,
                              float *grad_h0, float *grad_c0, float *cot_amax,
                              float *workspace, apg_stream_t stream) {
  if (!policy) { set_error("policy is NULL"); return APG_ERR_ARG; }
  return lstm_bwd(nullptr, state0, states, actions, ref, ref_cols, relu_mask, gates, hc, dt,
                  params, weights, policy, B, H, loss_partials, loss, d_gates, d_zout, d_conv,
                  grad_state0, grad_h0, grad_c0, cot_amax, workspace, stream);
}

int apg_quad_lstm_rollout_bwd_packed(const float *state0, const float *states,
                                     const float *actions, const float *ref, int ref_cols,
                                     const unsigned *relu_mask, const float *gates,
                                     const float *hc, float dt, const ApgQuadParams *params,
                                     const ApgQuadLossWeights *weights,
                                     const float *tables_bwd, int B, int H,
                                     float *loss_partials, float *loss, float *d_gates,
                                     float *d_zout, float *d_conv, float *grad_state0,
                                     float *grad_h0, float *grad_c0, float *cot_amax,
                                     apg_stream_t stream) {
  return lstm_bwd(nullptr, state0, states, actions, ref, ref_cols, relu_mask, gates, hc, dt,
                  params, weights, nullptr, B, H, loss_partials, loss, d_gates, d_zout, d_conv,
                  grad_state0, grad_h0, grad_c0, cot_amax, const_cast<float *>(tables_bwd), stream);
}

static int gw_blocks(int B) {
  const int groups = (B + 31) / 32, per = kGwWaves * kGwGroups;
  return 2 * ((groups + per - 1) / per);
}

int apg_quad_lstm_cot_amax_floats(int B) { return B <= 0 ? 0 : 2 * ((B + 31) / 32); }

int apg_quad_lstm_gate_wgrad_partials_floats(int B) {
  return B <= 0 ? 0 : gw_blocks(B) * kGwPart;
}

// `defer`: the sum of the partials is left to the caller (its arguments are handed
// back, chunks 0: nothing to add up)
static int gate_wgrad_impl(const float *state0, const float *states, const float *in_ref,
                           const float *acts, const float *d_gates, const float *d_zout,
                           const float *cot_amax, const ApgLstmPolicy *policy,
                           float *tables_fwd, int B, int H,
                           float *partials, float *ih_hh, float *b_ih, float *w_out,
                           float *b_out, apg_stream_t stream, apg::GwReduceArgs *defer) {
  using namespace apg;
  if (defer) defer->chunks = 0;
  if (H != kH) {
    set_error("the fused LSTM rollout is built for horizon %d (got %d)", kH, H);
    return APG_ERR_ARG;
  }
  if (B < 0 || (long long)B * kH * 4 * kNX >= (1ll << 32) - 64) {
    set_error("apg_quad_lstm_gate_wgrad: B out of range (%d)", B);
    return APG_ERR_ARG;
  }
  if (!ih_hh || !b_ih || !w_out || !b_out) {
    set_error("apg_quad_lstm_gate_wgrad: NULL gradient buffer");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (hipMemsetAsync(ih_hh, 0, sizeof(float) * kNG * (kNX + kNH), st) != hipSuccess ||
        hipMemsetAsync(b_ih, 0, sizeof(float) * kNG, st) != hipSuccess ||
        hipMemsetAsync(w_out, 0, sizeof(float) * 4 * kNH, st) != hipSuccess ||
        hipMemsetAsync(b_out, 0, sizeof(float) * 4, st) != hipSuccess)
      return check_launch("memset(gate gradients)");
    return APG_OK;
  }
  if (!state0 || !states || !in_ref || !acts || !d_gates || !d_zout || !cot_amax ||
      !tables_fwd || !partials) {
    set_error("apg_quad_lstm_gate_wgrad: NULL buffer");
    return APG_ERR_ARG;
  }
  if (policy) {
    if (!policy->conv_w || !policy->conv_b || !policy->w_ih || !policy->w_hh ||
        !policy->b_ih || !policy->b_hh || !policy->w_out || !policy->b_out) {
      set_error("policy weight pointer is NULL");
      return APG_ERR_ARG;
    }
    PackArgs P;
    P.pol = *policy, P.dst = tables_fwd;
    hipLaunchKernelGGL(lstm_pack_fwd16_kernel, dim3((kFwd16Lds + 255) / 256), dim3(256), 0, st,
                       P);
  }
  GwArgs A;
  A.state0 = state0, A.states = states, A.in_ref = in_ref, A.acts = acts;
  A.d_gates = d_gates, A.d_zout = d_zout, A.cot_amax = cot_amax;
  A.tables = tables_fwd, A.partials = partials;
  A.B = B;
  const int blocks = gw_blocks(B);
  {   // (145 KB of dynamic LDS: above the 64 KB a kernel gets unasked)
    static PerDeviceOnce attr;
    if (!attr.test()) {
      if (hipFuncSetAttribute((const void *)lstm_gate_wgrad_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((kGwTab + kGwWin + kGwBlk) * sizeof(float))) != hipSuccess)
        return check_launch("hipFuncSetAttribute(lstm_gate_wgrad)");
      attr.set();
    }
  }
  hipLaunchKernelGGL(lstm_gate_wgrad_kernel, dim3(blocks), dim3(kGwThreads),
                     (kGwTab + kGwWin + kGwBlk) * sizeof(float), st, A);
  if (int e = check_launch("quad_lstm_gate_wgrad")) return e;
  GwReduceArgs R;
  R.partials = partials, R.chunks = blocks / 2;
  R.ih_hh = ih_hh, R.b_ih = b_ih, R.w_out = w_out, R.b_out = b_out;
  if (defer) { *defer = R; return APG_OK; }
  hipLaunchKernelGGL(lstm_gate_wgrad_reduce_kernel, dim3(kGwReduceBlocks), dim3(256), 0, st, R);
  return check_launch("quad_lstm_gate_wgrad_reduce");
}

int apg_quad_lstm_gate_wgrad(const float *state0, const float *states, const float *in_ref,
                             const float *acts, const float *d_gates, const float *d_zout,
                             const float *cot_amax, const ApgLstmPolicy *policy,
                             float *tables_fwd, int B, int H,
                             float *partials, float *ih_hh, float *b_ih, float *w_out,
                             float *b_out, apg_stream_t stream) {
  return gate_wgrad_impl(state0, states, in_ref, acts, d_gates, d_zout, cot_amax, policy,
                         tables_fwd, B, H, partials, ih_hh, b_ih, w_out, b_out, stream,
                         nullptr);
}

int apg_quad_lstm_rollout_fwd_rows(const ApgBatchRows *rows, const float *h0, const float *c0,
                                   float dt, const ApgQuadParams *params,
                                   const float *tables_fwd, int B, int H, float *state0,
                                   float *in_ref, float *states, float *actions, float *x,
                                   float *gates, float *hc, float *hnew, unsigned *relu_mask,
                                   apg_stream_t stream) {
  if (!rows) { set_error("rows is NULL"); return APG_ERR_ARG; }
  return lstm_fwd(rows, state0, in_ref, h0, c0, dt, params, nullptr, B, H, states, actions, x,
                  gates, hc, hnew, relu_mask, const_cast<float *>(tables_fwd), stream);
}

int apg_quad_lstm_rollout_bwd_rows(const ApgBatchRows *rows, int ref_cols, const float *state0,
                                   const float *states, const float *actions,
                                   const unsigned *relu_mask, const float *gates,
                                   const float *hc, float dt, const ApgQuadParams *params,
                                   const ApgQuadLossWeights *weights, const float *tables_bwd,
                                   int B, int H, float *loss_partials, float *loss,
                                   float *d_gates, float *d_zout, float *d_conv,
                                   float *grad_state0, float *grad_h0, float *grad_c0,
                                   float *cot_amax, apg_stream_t stream) {
  if (!rows) { set_error("rows is NULL"); return APG_ERR_ARG; }
  return lstm_bwd(rows, state0, states, actions, nullptr, ref_cols, relu_mask, gates, hc, dt,
                  params, weights, nullptr, B, H, loss_partials, loss, d_gates, d_zout, d_conv,
                  grad_state0, grad_h0, grad_c0, cot_amax, const_cast<float *>(tables_bwd),
                  stream);
}

int apg_quad_lstm_conv_wgrad_partials_floats(int B) {
  return B <= 0 ? 0 : ((B + 32 * kCwWaves - 1) / (32 * kCwWaves)) * kCwPart;
}

static int conv_wgrad_impl(const float *d_conv, const float *in_ref, const float *st_all, int B,
                           int H, float *partials, float *conv_w, float *conv_pos,
                           float *conv_b, apg_stream_t stream, apg::CwReduceArgs *defer) {
  using namespace apg;
  if (defer) defer->wgs = 0;
  if (H != kH) {
    set_error("the fused LSTM rollout is built for horizon %d (got %d)", kH, H);
    return APG_ERR_ARG;
  }
  if (B < 0 || (long long)B * 4 * kConvPlanes >= (1ll << 32) - 64) {
    set_error("apg_quad_lstm_conv_wgrad: B out of range (%d)", B);
    return APG_ERR_ARG;
  }
  if (!conv_w || !conv_pos || !conv_b) {
    set_error("apg_quad_lstm_conv_wgrad: NULL gradient buffer");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (hipMemsetAsync(conv_w, 0, sizeof(float) * kNC * kCwTaps, st) != hipSuccess ||
        hipMemsetAsync(conv_pos, 0, sizeof(float) * kNC * 3, st) != hipSuccess ||
        hipMemsetAsync(conv_b, 0, sizeof(float) * kNC, st) != hipSuccess)
      return check_launch("memset(conv gradients)");
    return APG_OK;
  }
  if (!d_conv || !in_ref || !st_all || !partials) {
    set_error("apg_quad_lstm_conv_wgrad: NULL buffer");
    return APG_ERR_ARG;
  }
  {
    static PerDeviceOnce attr;
    if (!attr.test()) {
      if (hipFuncSetAttribute((const void *)lstm_conv_wgrad_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kCwLds * sizeof(float))) != hipSuccess)
        return check_launch("hipFuncSetAttribute(lstm_conv_wgrad)");
      attr.set();
    }
  }
  CwArgs A;
  A.d_conv = d_conv, A.in_ref = in_ref, A.st_all = st_all, A.partials = partials, A.B = B;
  const int wgs = (B + 32 * kCwWaves - 1) / (32 * kCwWaves);
  hipLaunchKernelGGL(lstm_conv_wgrad_kernel, dim3(wgs), dim3(kCwThreads),
                     kCwLds * sizeof(float), st, A);
  if (int e = check_launch("quad_lstm_conv_wgrad")) return e;
  CwReduceArgs R;
  R.partials = partials, R.wgs = wgs;
  R.conv_w = conv_w, R.conv_pos = conv_pos, R.conv_b = conv_b;
  if (defer) { *defer = R; return APG_OK; }
  hipLaunchKernelGGL(lstm_conv_wgrad_reduce_kernel, dim3(kCwReduceBlocks), dim3(256), 0, st, R);
  return check_launch("quad_lstm_conv_wgrad_reduce");
}

int apg_quad_lstm_conv_wgrad(const float *d_conv, const float *in_ref, const float *st_all, int B,
                             int H, float *partials, float *conv_w, float *conv_pos,
                             float *conv_b, apg_stream_t stream) {
  return conv_wgrad_impl(d_conv, in_ref, st_all, B, H, partials, conv_w, conv_pos, conv_b, stream,
                         nullptr);
}

int apg_quad_lstm_wgrads(const float *state0, const float *states, const float *in_ref,
                         const float *acts, const float *d_gates, const float *d_zout,
                         const float *cot_amax, const float *d_conv, const float *st_all,
                         const ApgLstmPolicy *policy, float *tables_fwd, int B, int H,
                         float *gate_partials, float *conv_partials, float *ih_hh, float *b_ih,
                         float *w_out, float *b_out, float *conv_w, float *conv_pos,
                         float *conv_b, const ApgLstmStepTail *finish, apg_stream_t stream) {
  using namespace apg;
  GwReduceArgs G;
  CwReduceArgs C;
  FuseArgs U;
  U.on = U.update = 0, U.lr = U.momentum = 0.0;
  U.grad = U.param = U.mom = ApgLstmPolicyGrads{};
  if (finish) {
    const ApgLstmStepTail &t = *finish;
    const float *const need[] = {t.grad.conv_w, t.grad.conv_b, t.grad.w_ih, t.grad.w_hh,
                                 t.grad.b_ih,   t.grad.b_hh,   t.grad.w_out, t.grad.b_out};
    for (const float *q : need)
      if (!q) { set_error("apg_quad_lstm_wgrads: finish: NULL gradient tensor"); return APG_ERR_ARG; }
    if (t.update) {
      const float *const pm[] = {t.param.conv_w, t.param.conv_b, t.param.w_ih, t.param.w_hh,
                                 t.param.b_ih,   t.param.b_hh,   t.param.w_out, t.param.b_out,
                                 t.mom.conv_w,   t.mom.conv_b,   t.mom.w_ih,   t.mom.w_hh,
                                 t.mom.b_ih,     t.mom.b_hh,     t.mom.w_out,  t.mom.b_out};
      for (const float *q : pm)
        if (!q) {
          set_error("apg_quad_lstm_wgrads: finish: NULL parameter / momentum buffer");
          return APG_ERR_ARG;
        }
    }
    U.on = 1, U.update = t.update != 0, U.lr = t.lr, U.momentum = t.momentum;
    U.grad = t.grad, U.param = t.param, U.mom = t.mom;
  }
  if (int e = gate_wgrad_impl(state0, states, in_ref, acts, d_gates, d_zout, cot_amax, policy,
                              tables_fwd, B, H, gate_partials, ih_hh, b_ih, w_out, b_out, stream,
                              &G))
    return e;
  if (int e = conv_wgrad_impl(d_conv, in_ref, st_all, B, H, conv_partials, conv_w, conv_pos,
                              conv_b, stream, &C))
    return e;
  if (!G.chunks || !C.wgs) {                   // (B = 0: the gradients were zeroed)
    if (finish) {
      set_error("apg_quad_lstm_wgrads: finish needs B > 0");
      return APG_ERR_ARG;
    }
    return APG_OK;
  }
  hipLaunchKernelGGL(lstm_wgrad_reduce_kernel, dim3(kGwReduceBlocks + kCwReduceBlocks), dim3(256),
                     0, (hipStream_t)stream, G, C, U);
  return check_launch("quad_lstm_wgrad_reduce");
}

int apg_quad_lstm_tables_floats(int reverse) { return reverse ? kBwd16Lds : kFwd16Lds; }

int apg_quad_lstm_pack_tables(const ApgLstmPolicy *policy, float *tables_fwd,
                              float *tables_bwd, apg_stream_t stream) {
  if (!policy || !tables_fwd || !tables_bwd) {
    set_error("apg_quad_lstm_pack_tables: NULL argument");
    return APG_ERR_ARG;
  }
  if (!policy->conv_w || !policy->conv_b || !policy->w_ih || !policy->w_hh || !policy->b_ih ||
      !policy->b_hh || !policy->w_out || !policy->b_out) {
    set_error("policy weight pointer is NULL");
    return APG_ERR_ARG;
  }
  PackArgs F, R;
  F.pol = *policy, F.dst = tables_fwd;
  R.pol = *policy, R.dst = tables_bwd;
  const int fb = (kFwd16Lds + 255) / 256, rb = (kBwd16Lds + 255) / 256;
  hipLaunchKernelGGL(lstm_pack_both_kernel, dim3(fb + rb), dim3(256), 0, (hipStream_t)stream, F,
                     R, fb);
  return check_launch("quad_lstm_pack_tables");
}

int apg_quad_lstm_step_tail(const ApgLstmStepTail *tail, apg_stream_t stream) {
  if (!tail) { set_error("apg_quad_lstm_step_tail: tail is NULL"); return APG_ERR_ARG; }
  const ApgLstmStepTail &t = *tail;
  const float *const need[] = {t.grad.conv_w, t.grad.conv_b, t.grad.w_ih, t.grad.w_hh,
                               t.grad.b_ih, t.grad.b_hh, t.grad.w_out, t.grad.b_out,
                               t.ih_hh, t.conv_pos};
  for (const float *q : need)
    if (!q && !t.applied) {
      set_error("apg_quad_lstm_step_tail: NULL gradient buffer");
      return APG_ERR_ARG;
    }
  if (t.update || t.tables_fwd || t.tables_bwd) {
    const float *const pm[] = {t.param.conv_w, t.param.conv_b, t.param.w_ih, t.param.w_hh,
                               t.param.b_ih, t.param.b_hh, t.param.w_out, t.param.b_out};
    for (const float *q : pm)
      if (!q) { set_error("apg_quad_lstm_step_tail: NULL parameter"); return APG_ERR_ARG; }
  }
  if (t.update) {
    const float *const pm[] = {t.mom.conv_w, t.mom.conv_b, t.mom.w_ih, t.mom.w_hh,
                               t.mom.b_ih, t.mom.b_hh, t.mom.w_out, t.mom.b_out};
    for (const float *q : pm)
      if (!q) { set_error("apg_quad_lstm_step_tail: NULL momentum buffer"); return APG_ERR_ARG; }
  }
  if ((t.tables_fwd == nullptr) != (t.tables_bwd == nullptr)) {
    set_error("apg_quad_lstm_step_tail: both table sets or none");
    return APG_ERR_ARG;
  }
  if (t.loss && (!t.loss_partials || t.n_partials < 0)) {
    set_error("apg_quad_lstm_step_tail: loss without partials");
    return APG_ERR_ARG;
  }
  TailArgs A;
  A.t = t;
  if (t.applied) {
    const bool pack = t.tables_fwd != nullptr;
    const int fwd_blocks = pack ? (kFwd16Lds + 255) / 256 : 0;
    const int bwd_blocks = pack ? (kBwd16Lds + 255) / 256 : 0;
    const int blocks = fwd_blocks + bwd_blocks + (t.loss ? 1 : 0);
    if (!blocks) return APG_OK;
    hipLaunchKernelGGL(lstm_tail_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       A, fwd_blocks, bwd_blocks);
    return check_launch("quad_lstm_step_tail(applied)");
  }
  hipLaunchKernelGGL(lstm_step_tail_kernel, dim3(1), dim3(kTailThreads), 0, (hipStream_t)stream,
                     A);
  return check_launch("quad_lstm_step_tail");
}

int apg_quad_lstm_closed_loop(const float *traj, int L, const float *h0,
                              const float *c0, float dt,
                              const ApgQuadParams *params,
                              const ApgLstmPolicy *policy, int B, int H,
                              int max_steps, float thresh_div,
                              float thresh_stable, int test_time, float *div,
                              int *steps, float *drone, float *actions,
                              float *start_states, float *workspace,
                              apg_stream_t stream) {
  return apg_quad_lstm_closed_loop_env(traj, L, h0, c0, dt, params, nullptr, policy, B, H,
                                       max_steps, thresh_div, thresh_stable, test_time, div,
                                       steps, drone, actions, start_states, workspace, stream);
}

int apg_quad_lstm_closed_loop_env(const float *traj, int L, const float *h0, const float *c0,
                                  float dt, const ApgQuadParams *params,
                                  const ApgLearntResidual *learnt, const ApgLstmPolicy *policy,
                                  int B, int H, int max_steps, float thresh_div,
                                  float thresh_stable, int test_time, float *div, int *steps,
                                  float *drone, float *actions, float *start_states,
                                  float *workspace, apg_stream_t stream) {
  if (int e = check_lstm(params, policy, B, H)) return e;
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
  const long long planes = (long long)(T + 1) * 12 > (long long)L * 9
                               ? (long long)(T + 1) * 12 : (long long)L * 9;
  if ((long long)B * 4 * planes >= (1ll << 32) - 64) {
    set_error("B * steps too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  if (B == 0) return APG_OK;
  if (!traj || !h0 || !c0 || !div || !steps || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  LoopArgs A;
  A.traj = traj, A.h0 = h0, A.c0 = c0, A.div = div, A.steps = steps;
  A.drone = drone, A.actions = actions, A.start = start_states;
  A.tables = workspace;
  A.c = make_const(*params, dt);
  A.B = B, A.L = L, A.T = T, A.test_time = test_time;
  A.thresh_div = thresh_div, A.thresh_stable = thresh_stable;
  A.learnt = learnt != nullptr;
  PackArgs P;
  P.pol = *policy, P.dst = workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lstm_pack_fwd16_kernel, dim3((kFwd16Lds + 255) / 256), dim3(256),
                     0, st, P);
  if (learnt)
    hipLaunchKernelGGL(learnt_pack_kernel, dim3((kLearntFloats + 255) / 256), dim3(256), 0, st,
                       *learnt, workspace + kFwd16Lds);
  const dim3 grid((B + kTrajPerBlock - 1) / kTrajPerBlock);
  if (learnt)
    hipLaunchKernelGGL(lstm_closed_loop_kernel<true>, grid, dim3(kThreads),
                       (kFwd16Lds + kLearntFloats) * sizeof(float), st, A);
  else
    hipLaunchKernelGGL(lstm_closed_loop_kernel<false>, grid, dim3(kThreads), kFwd16Lds * sizeof(float), st, A);
  return check_launch("quad_lstm_closed_loop");
}

}  // extern "C"
