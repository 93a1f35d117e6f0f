// mlp_rollout.hip (until round 6: mlp.hip) - the AUTOREGRESSIVE quadrotor unroll with the MLP policy inside the
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
#include "mlp_common.h"

namespace apg {
namespace {
// LEGACY (forward only): scripts/train_drone.py:138-142 AS SHIPPED - the window is a
// view of the batch, the relative-position subtraction writes through it and every
// step shifts the rows its window holds again (SURVEY.md 8a A4 `legacy_inplace_ref`)
template <bool XMAX, bool LEGACY = false>
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
    float sub[3] = {hi ? 0.f : s[0], hi ? 0.f : s[1], hi ? 0.f : s[2]};
    if (LEGACY) {   // the shift stays in the window
#pragma unroll
      for (int r = 0; r < kH; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) w[r][j] -= sub[j];
      sub[0] = sub[1] = sub[2] = 0.f;
    }
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
          v = relu1(v);
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
        for (int i = 0; i < 12; ++i) rv[e * 12 + i] = relu1(cv[i]);
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

static int mlp_rollout_fwd(const float *state0, const float *in_ref, float dt,
                           const ApgQuadParams *params, const ApgMlpPolicy *policy, int B, int H,
                           float *states, float *actions, float *feat, float *x1, float *h,
                           unsigned *relu_mask, float *workspace, apg_stream_t stream,
                           bool legacy_inplace_ref);

int apg_quad_mlp_rollout_fwd(const float *state0, const float *in_ref, float dt,
                             const ApgQuadParams *params,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *states, float *actions, float *feat,
                             float *x1, float *h, unsigned *relu_mask,
                             float *workspace, apg_stream_t stream) {
  return mlp_rollout_fwd(state0, in_ref, dt, params, policy, B, H, states, actions, feat, x1, h,
                         relu_mask, workspace, stream, false);
}

int apg_quad_mlp_rollout_fwd_inplace_ref(const float *state0, const float *in_ref, float dt,
                                         const ApgQuadParams *params,
                                         const ApgMlpPolicy *policy, int B, int H,
                                         float *states, float *actions, float *feat,
                                         float *x1, float *h, unsigned *relu_mask,
                                         float *workspace, apg_stream_t stream) {
  return mlp_rollout_fwd(state0, in_ref, dt, params, policy, B, H, states, actions, feat, x1, h,
                         relu_mask, workspace, stream, true);
}

static int mlp_rollout_fwd(const float *state0, const float *in_ref, float dt,
                           const ApgQuadParams *params, const ApgMlpPolicy *policy, int B, int H,
                           float *states, float *actions, float *feat, float *x1, float *h,
                           unsigned *relu_mask, float *workspace, apg_stream_t stream,
                           bool legacy_inplace_ref) {
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
    if (int e = raise_lds(mlp_rollout_fwd_kernel<false, true>, kCfLds)) return e;
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
  const dim3 grid((B + kTrajPerBlock - 1) / kTrajPerBlock);
  if (legacy_inplace_ref)
    hipLaunchKernelGGL((mlp_rollout_fwd_kernel<false, true>), grid, dim3(kThreads),
                       kCfLds * sizeof(float), (hipStream_t)stream, A);
  else
    hipLaunchKernelGGL(mlp_rollout_fwd_kernel<false>, grid, dim3(kThreads),
                       kCfLds * sizeof(float), (hipStream_t)stream, A);
  return check_launch("quad_mlp_rollout_fwd");
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
