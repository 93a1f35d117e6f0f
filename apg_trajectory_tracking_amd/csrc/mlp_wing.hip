// mlp_wing.hip - the fixed-wing controller on the matrix cores (BASELINE config
// 4's full training step: fixed-wing concurrent, H = 20).
//
// Replaces the policy part of the concurrent step of the fixed-wing trainer:
//   TrainBase.run_epoch, concurrent branch   scripts/train_base.py:198-204
//     actions = sigmoid(net(in_state, in_ref_state)) reshaped [B, H, 4]
//   hutter_model.Net(9, 1, 3, 80, conv=False) neural_control/models/hutter_model.py:6-49
//     s1 = tanh(W_s in_state + b_s) (9 -> 64), r1 = tanh(W_r in_ref + b_r) (3 -> 64),
//     h1 = tanh(W_1 [s1, r1] + b_1) (128 -> 64), h2, h3 (64 -> 64), out (64 -> 80)
// and its backward.  The dynamics / loss / adjoint stay in wing.hip
// (apg_wing_rollout_fwd_bwd): the forward kernel writes the actions as
// [H][4][B] planes - exactly what that kernel reads - and the reverse kernel
// starts from its dL/dactions planes.  Same layout as policy_mfma.h: one wave =
// 32 trajectories, layers chain through the accumulator registers; the two
// input branches are ONE block-diagonal 12 -> 128 layer.  Since round 3 the
// layers are v_mfma_f32_32x32x16_f16 products of fp16-split operands
// (policy_mfma16.h: fp32 accuracy, cotangents scaled per trajectory).  Weight
// gradients: cotangent planes x activation planes by apg_planes_gemm_grouped.
#include "apg_device.h"
#include "policy_mfma.h"
#include "policy_mfma16.h"
#include "wing_math.h"
#include "learnt_residual.h"

namespace apg {
namespace {

constexpr int kNS = 9, kNR = 3, kNI = kNS + kNR;  // network inputs (12)
constexpr int kW = 64, kW0 = 2 * kW;              // layer widths (64; first 128)
constexpr int kNA = 80;                           // head width = H x 4 actions
constexpr int kThreads = 512;
constexpr int kTrajPerBlock = kThreads / 2;

// ------------------------------------------------------------------ forward
struct PackArgs {
  ApgWingPolicy pol;
  float *dst;
  int head_rows;  // rows of fc_out behind pol.w_out / b_out that may be read
};

// Tables (fp16 split operands, policy_mfma16.h).  Forward: the bias tables
// [rb][16][2] (first layer 4 row blocks, then fc1, fc2, fc3, head 3 row
// blocks), then 48 A-operand blocks of 2 KB: first layer [rb of 4] (one
// k-block: inputs 8 hi + j of the 12), fc1 [rb][kb of 8], fc2 / fc3 [rb][kb],
// head [rb of 3][kb].
constexpr int hT0 = 0, hT1 = 128, hT2 = 192, hT3 = 256, hTo = 320;   // floats
constexpr int hA = 2048;                                            // bytes
constexpr int n0 = 0, n1 = 4, n2 = 20, n3 = 28, nO = 36, nBlocks16 = 48;
constexpr int kFwd16Lds = (hA + nBlocks16 * kBlock16) / 4;   // 25 088 floats = 100 352 B
static_assert(hTo + 96 <= hA / 4, "LDS map");

// input index of slot j of k-block kb for a layer fed by FOUR row blocks
// (fc1: 128 inputs) or two (64): registers 8 (kb & 1) .. + 7 of block kb >> 1
__device__ __forceinline__ float wing_fwd16_weight(const ApgWingPolicy &p, int n, int row,
                                                   int j, int hi, int head_rows) {
  if (n < n1) {                        // [states_in 0; 0 ref_in], inputs 8 hi + j
    const int m = n * 32 + row, k = 8 * hi + j;
    if (m < kW && k < kNS) return p.w_s[m * kNS + k];
    if (m >= kW && k >= kNS && k < kNI) return p.w_r[(m - kW) * kNR + (k - kNS)];
    return 0.f;
  }
  if (n < n2) {
    const int m = n - n1, rb = m / 8, kb = m % 8;
    return p.w_1[(rb * 32 + row) * kW0 + kin(kb, j, hi)];
  }
  if (n < nO) {
    const int m = (n - n2) % 8, rb = m / 4, k = kin(m % 4, j, hi);
    return n < n3 ? p.w_2[(rb * 32 + row) * kW + k] : p.w_3[(rb * 32 + row) * kW + k];
  }
  const int m = n - nO, rb = m / 4, out = rb * 32 + row;
  return out < head_rows ? p.w_out[out * kW + kin(m % 4, j, hi)] : 0.f;
}

__global__ __launch_bounds__(256) void wing_pack_fwd16_kernel(PackArgs A) {
  const ApgWingPolicy &p = A.pol;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  for (int idx = tid; idx < nBlocks16 * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    unsigned h, lo;
    split_pair(wing_fwd16_weight(p, n, l & 31, 2 * q, l >> 5, A.head_rows),
               wing_fwd16_weight(p, n, l & 31, 2 * q + 1, l >> 5, A.head_rows), h, lo);
    dst[(hA + n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(hA + n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
  for (int idx = tid; idx < 128; idx += T) {
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = idx >> 5;
    const int row = rb * 32 + rrow(i) + 4 * hi;
    A.dst[hT0 + idx] = row < kW ? p.b_s[row] : p.b_r[row - kW];
    if (rb < 2) {
      A.dst[hT1 + idx] = p.b_1[row];
      A.dst[hT2 + idx] = p.b_2[row];
      A.dst[hT3 + idx] = p.b_3[row];
    }
    if (rb < 3) A.dst[hTo + idx] = row < A.head_rows ? p.b_out[row] : 0.f;
  }
}

struct Args {
  const float *feat, *ref_in;   // [9][B], [3][B]
  float *actions;               // [80][B] = [H][4][B]
  const float *grad_actions;    // [80][B] (reverse)
  float *x1, *h;                // [128][B], [192][B]
  float *d_zout, *d_pre;        // [80][B], [320][B]: fc1, fc2, fc3 (64 each), first layer (128)
  const float *tables;
  int B, NA;   // NA = 4 H head rows in use (40 or 80)
};

__global__ __launch_bounds__(kThreads) void wing_policy_fwd_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kFwd16Lds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pfe(A.feat, kNS, pN), Prf(A.ref_in, kNR, pN);
  const Planes Pac(A.actions, A.NA, pN), Px1(A.x1, kW0, pN), Ph(A.h, 3 * kW, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;  // + row 4 hi

  float in[kNI];
#pragma unroll
  for (int j = 0; j < kNS; ++j) in[j] = Pfe.ld(vb, j * pN);
#pragma unroll
  for (int j = 0; j < kNR; ++j) in[kNS + j] = Prf.ld(vb, j * pN);

  // the layers on the 16-bit matrix pipe (policy_mfma16.h): every operand as
  // two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  // first layer: [states_in 0; 0 ref_in], 12 -> 128: one k-block
  f32x16 x[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) x[rb][i] = L.T(hT0 + (rb * 16 + i) * 2);
  {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = hi ? (8 + j < kNI ? in[8 + j < kNI ? 8 + j : 0] : 0.f) : in[j];
    const Op16 xi = split8(v);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) x[rb] = mma3(L16.A(hA, n0 + rb), xi, x[rb]);
  }
  // fc1 (tanh of the first layer applied, and stored, where it is consumed)
  f32x16 u[2], a[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) a[rb][i] = L.T(hT1 + (rb * 16 + i) * 2);
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * (kb & 1) + j;
      v[j] = tanh_fast(x[kb >> 1][i]);
      Px1.st(vr, ((kb >> 1) * 32 + rrow(i)) * pN, v[j]);
    }
    const Op16 xv = split8(v);
    a[0] = mma3(L16.A(hA, n1 + kb), xv, a[0]);
    a[1] = mma3(L16.A(hA, n1 + 8 + kb), xv, a[1]);
  }
  // fc2, fc3
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) u[rb][i] = L.T((layer ? hT3 : hT2) + (rb * 16 + i) * 2);
    dense64_16(u, a, L16, hA, layer ? n3 : n2, [&](int rb, int i, float v) {
      const float tv = tanh_fast(v);
      Ph.st(vr, (layer * kW + rb * 32 + rrow(i)) * pN, tv);
      return tv;
    });
    a[0] = u[0], a[1] = u[1];
  }
  // head: 80 outputs in three row blocks
  f32x16 z[3];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) z[rb][i] = L.T(hTo + (rb * 16 + i) * 2);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * (kb & 1) + j;
      v[j] = tanh_fast(a[kb >> 1][i]);
      Ph.st(vr, (2 * kW + (kb >> 1) * 32 + rrow(i)) * pN, v[j]);
    }
    const Op16 xv = split8(v);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) z[rb] = mma3(L16.A(hA, nO + rb * 4 + kb), xv, z[rb]);
  }
  // actions = sigmoid(z): accumulator rows ARE the action planes
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (rb * 32 + rrow(i) + 4 < kNA)  // rows r(i), r(i) + 4 both < 80; rows >= NA
        Pac.st(vr, (rb * 32 + rrow(i)) * pN, sigmoidf_(z[rb][i]));  // are out of range
}

// ------------------------------------------------------ closed-loop evaluation
// Beyond SURVEY.md §8 (VERDICT r2 "what's missing" #5): FixedWingEvaluator.
// fly_to_point (scripts/evaluate_fixed_wing.py:45-131) for a batch of target
// lists in ONE launch.  Per step: WingDataset.prepare_data
// (neural_control/dataset.py:322-350: ((state - mean) / std)[3:] and the last
// point of the linear reference relative to the aircraft),
// FixedWingNetWrapper.predict_actions (network_wrapper.py:81-98: sigmoid, first
// action of the plan), SimpleWingEnv.step (wing_env.py:44-57: dynamics +
// |roll|, |pitch| < thresh_stable), project_to_line (q_funcs.py:6-18), the
// target switch and either the break (test_time) or the reset onto the line.
// As in the reference, after a reset the POLICY still sees the last simulated
// state (the local `state` of fly_to_point is not refreshed, :124) while the
// environment continues from the reset state.
struct WingLoopArgs {
  const float *targets;  // [n_targets][3][B]
  const float *state0;   // [12][B] or NULL: SimpleWingEnv.zero_reset (:26-28)
  float *div_linear;     // [T][B] distance to the current line after each step
  float *div_pass;       // [T][B] miss distance when a target is passed, else -1
  float *div_fail;       // [T][B] entry appended on divergence, else -1
  int *steps;            // [B] iterations executed = len(drone_traj)
  float *drone;          // [T][16][B] or NULL: state after the step + action
  float *seen;           // [T][15][B] or NULL: state the policy saw + its target
  const float *tables;
  WingGeneralConst k;         // (the full inertia matrix: LearntFixedWingDynamics only)
  float mean[kNS], std[kNS];  // entries 3..11 of the data set's mean / std
  float vec_len, horizon;     // 12 * dt of the data set, its horizon
  float thresh_div, thresh_stable, des_speed;
  int B, T, n_targets, test_time;
};

// a + ab (ab . (p - a)) / |ab|^2, and a itself for a degenerate line
__device__ __forceinline__ void project_to_line(const float (&a)[3], const float (&b)[3],
                                                const float (&p)[3], float (&out)[3]) {
  const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
  const float n2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
  const float d =
      ab[0] * (p[0] - a[0]) + ab[1] * (p[1] - a[1]) + ab[2] * (p[2] - a[2]);
  const float f = n2 > 0.f ? d / n2 : 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) out[j] = a[j] + ab[j] * f;
}

__device__ __forceinline__ float dist3(const float (&a)[3], const float (&b)[3]) {
  const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
  return sqrtf(x * x + y * y + z * z);
}

// LEARNT: the environment steps through LearntFixedWingDynamics.forward
// (neural_control/dynamics/fixed_wing_dynamics.py:270-326): simulate_fixed_wing on
// its CURRENT parameters - the 3 x 3 inertia in full - plus the residual network
// on [state, action] (learnt_residual.h, weights behind the policy tables); what
// SimpleWingEnv(train_dynamics) is after train_dynamics() (scripts/
// train_fixed_wing.py:42-43).  A second instantiation: the analytic loop keeps
// its registers.
template <bool LEARNT>
__global__ __launch_bounds__(kThreads) void wing_closed_loop_kernel(WingLoopArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kFwd16Lds + (LEARNT ? kLearntFloats : 0));
  const LdsView16 L16(lds, threadIdx.x & 63);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B, T = A.T;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  // a NULL tensor becomes an empty buffer: loads give 0, stores are dropped
  const Planes Ptg(A.targets, A.n_targets * 3, pN), Ps0(A.state0, A.state0 ? 12 : 0, pN);
  const Planes Pdl(A.div_linear, T, pN), Pdp(A.div_pass, T, pN), Pdf(A.div_fail, T, pN);
  const Planes Pdr(A.drone, A.drone ? T * 16 : 0, pN);
  const Planes Pse(A.seen, A.seen ? T * 15 : 0, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;

  float s[12], obs[12], line[3], prev[3];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = Ps0.ld(vb, i * pN);
  if (!A.state0) s[3] = 11.5f;
#pragma unroll
  for (int i = 0; i < 12; ++i) obs[i] = s[i];
#pragma unroll
  for (int j = 0; j < 3; ++j) line[j] = prev[j] = s[j];
  int ti = 0, steps = 0;
  bool alive = live;

#pragma unroll 1
  for (int k = 0; k < T; ++k) {
    const unsigned pB = opaque(pN);
    const unsigned vrec = (alive && hi == 0) ? vb : kDead;
    const unsigned vt = live ? vb + (unsigned)(ti * 3) * pB : kDead;
    float tg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) tg[j] = Ptg.ld(vt, j * pB);
#pragma unroll
    for (int i = 0; i < 12; ++i) Pse.st(vrec, (k * 15 + i) * pB, obs[i]);
#pragma unroll
    for (int j = 0; j < 3; ++j) Pse.st(vrec, (k * 15 + 12 + j) * pB, tg[j]);

    // WingDataset.prepare_data
    float in[kNI];
#pragma unroll
    for (int j = 0; j < kNS; ++j) in[j] = (obs[3 + j] - A.mean[j]) / A.std[j];
    {
      const float rel[3] = {tg[0] - obs[0], tg[1] - obs[1], tg[2] - obs[2]};
      const float nrm = sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]);
#pragma unroll
      for (int j = 0; j < 3; ++j)  // last point of _compute_target_pos, then - position
        in[kNS + j] = (obs[j] + (rel[j] / nrm) * A.vec_len * A.horizon) - obs[j];
    }

    // the policy (as wing_policy_fwd_kernel: fp16-split operands on the 16-bit
    // matrix pipe; nothing saved, first head block only)
    f32x16 x[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) x[rb][i] = L.T(hT0 + (rb * 16 + i) * 2);
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = hi ? (8 + j < kNI ? in[8 + j < kNI ? 8 + j : 0] : 0.f) : in[j];
      const Op16 xi = split8(v);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) x[rb] = mma3(L16.A(hA, n0 + rb), xi, x[rb]);
    }
    f32x16 u[2], a[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) a[rb][i] = L.T(hT1 + (rb * 16 + i) * 2);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tanh_fast(x[kb >> 1][8 * (kb & 1) + j]);
      const Op16 xv = split8(v);
      a[0] = mma3(L16.A(hA, n1 + kb), xv, a[0]);
      a[1] = mma3(L16.A(hA, n1 + 8 + kb), xv, a[1]);
    }
    const auto th = [](int, int, float v) { return tanh_fast(v); };
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          u[rb][i] = L.T((layer ? hT3 : hT2) + (rb * 16 + i) * 2);
      dense64_16(u, a, L16, hA, layer ? n3 : n2, th);
      a[0] = u[0], a[1] = u[1];
    }
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = L.T(hTo + i * 2);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tanh_fast(a[kb >> 1][8 * (kb & 1) + j]);
      z = mma3(L16.A(hA, nO + kb), split8(v), z);
    }
    // head rows 0..3 (the first action of the plan) sit in registers 0..3 of
    // the lower half-wave
    float act[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float oth = other_half(z[j]);
      act[j] = sigmoidf_(hi ? oth : z[j]);
    }

    // SimpleWingEnv.step
    if (LEARNT) {
      float x[16];
#pragma unroll
      for (int i = 0; i < 12; ++i) x[i] = s[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[12 + j] = act[j];
      wing_step(s, act, A.k);
      learnt_residual_add(s, x, lds + kFwd16Lds, hi);
    } else {
      wing_step(s, act, static_cast<const WingConst &>(A.k));
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) obs[i] = s[i];
    const bool stable =
        fabsf(s[6]) < A.thresh_stable && fabsf(s[7]) < A.thresh_stable;
    const float pos[3] = {s[0], s[1], s[2]};
    float on_line[3];
    project_to_line(line, tg, pos, on_line);
    const float div = dist3(on_line, pos);
    Pdl.st(vrec, k * pB, div);
#pragma unroll
    for (int i = 0; i < 12; ++i) Pdr.st(vrec, (k * 16 + i) * pB, s[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) Pdr.st(vrec, (k * 16 + 12 + j) * pB, act[j]);

    bool done = false;
    float d_pass = -1.f, d_fail = -1.f;
    if (pos[0] > tg[0]) {  // passed the target: miss distance on the last segment
      float on_seg[3];
      project_to_line(prev, pos, tg, on_seg);
      d_pass = dist3(on_seg, tg);
      if (ti < A.n_targets - 1) {
        ++ti;
#pragma unroll
        for (int j = 0; j < 3; ++j) line[j] = pos[j];
      } else {
        done = true;
      }
    }
    if (!done && (!stable || div > A.thresh_div)) {
      d_fail = A.thresh_div;
      if (A.test_time) {
        d_fail = dist3(pos, tg);
        done = true;
      } else {  // continue on the line, flying towards the (old) target
        const float v[3] = {tg[0] - on_line[0], tg[1] - on_line[1], tg[2] - on_line[2]};
        const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          s[j] = on_line[j];
          s[3 + j] = v[j] / vn * A.des_speed;
          s[6 + j] = 0.f;
          s[9 + j] = 0.f;
        }
      }
    }
    Pdp.st(vrec, k * pB, d_pass);
    Pdf.st(vrec, k * pB, d_fail);
#pragma unroll
    for (int j = 0; j < 3; ++j) prev[j] = pos[j];
    if (alive) steps = k + 1;
    alive = alive && !done;
    if (!__any(alive)) break;
  }
  if (live && hi == 0) A.steps[b] = steps;
}

// ------------------------------------------------------------------ reverse
// k index of head k-pair c (accumulator layout of the 80 outputs: row blocks
// 0 and 1 registers 0..15, row block 2 registers 0..7)
__host__ __device__ constexpr int khead(int c, int hi) {
  return (c >> 4) * 32 + rrow(c & 15) + 4 * hi;
}

// Reverse tables (fp16 split operands): 42 transposed blocks: head^T
// [rb][kb of 5] (this lane's 40 dL/dz rows), fc3^T, fc2^T [rb][kb], fc1^T
// [rb of 4][kb].
constexpr int mOT = 0, m3T = 10, m2T = 18, m1T = 26, mBlocks16 = 42;
constexpr int kBwd16Lds = mBlocks16 * kBlock16 / 4;   // 21 504 floats = 86 016 B

__device__ __forceinline__ float wing_bwd16_weight(const ApgWingPolicy &p, int n, int row,
                                                   int j, int hi, int head_rows) {
  if (n < m3T) {
    const int rb = n / 5, c = (n % 5) * 8 + j;       // dz register c: head row khead(c, hi)
    return khead(c, hi) < head_rows ? p.w_out[khead(c, hi) * kW + rb * 32 + row] : 0.f;
  }
  if (n < m1T) {
    const int m = (n - m3T) % 8, rb = m / 4, k = kin(m % 4, j, hi);
    return n < m2T ? p.w_3[k * kW + rb * 32 + row] : p.w_2[k * kW + rb * 32 + row];
  }
  const int m = n - m1T, rb = m / 4;
  return p.w_1[kin(m % 4, j, hi) * kW0 + rb * 32 + row];
}

__global__ __launch_bounds__(256) void wing_pack_bwd16_kernel(PackArgs A) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  unsigned *dst = reinterpret_cast<unsigned *>(A.dst);
  for (int idx = tid; idx < mBlocks16 * 64 * 4; idx += T) {
    const int q = idx & 3, l = (idx >> 2) & 63, n = idx >> 8;
    unsigned h, lo;
    split_pair(wing_bwd16_weight(A.pol, n, l & 31, 2 * q, l >> 5, A.head_rows),
               wing_bwd16_weight(A.pol, n, l & 31, 2 * q + 1, l >> 5, A.head_rows), h, lo);
    dst[(n * kBlock16) / 4 + l * 4 + q] = h;
    dst[(n * kBlock16 + 1024) / 4 + l * 4 + q] = lo;
  }
}

__global__ __launch_bounds__(kThreads) void wing_policy_bwd_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kBwd16Lds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pac(A.actions, A.NA, pN), Pga(A.grad_actions, A.NA, pN);
  const Planes Px1(A.x1, kW0, pN), Ph(A.h, 3 * kW, pN);
  const Planes Pdz(A.d_zout, A.NA, pN), Pdp(A.d_pre, 3 * kW + kW0, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;

  // dL/dz = dL/da * a (1 - a), 40 k-pairs in accumulator layout (rows beyond
  // NA are outside the buffers: they load 0 and their stores are dropped)
  float dz[40];
#pragma unroll
  for (int c = 0; c < 40; ++c) {
    const float a = Pac.ld(vr, khead(c, 0) * pN), g = Pga.ld(vr, khead(c, 0) * pN);
    dz[c] = g * a * (1.f - a);
    Pdz.st(vr, khead(c, 0) * pN, dz[c]);
  }
  float hv[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) hv[rb][i] = Ph.ld(vr, (2 * kW + rb * 32 + rrow(i)) * pN);
  __builtin_amdgcn_sched_barrier(0);
  // the reverse layers on the 16-bit matrix pipe: cotangents scaled per
  // trajectory, two fp16 terms, three products per k-block
  const LdsView16 L16(lds, lane);
  f32x16 d[2], e[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) d[rb][i] = 0.f;
  int ex;
  {  // dL/dh3 = W_out^T dL/dz: this lane's 40 rows = 5 k-blocks
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 40; ++c) amax = fmaxf(amax, fabsf(dz[c]));
    ex = scale_exponent(amax);
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_ldexpf(dz[kb * 8 + j], -ex);
      const Op16 xv = split8(v);
      d[0] = mma3(L16.A(0, mOT + kb), xv, d[0]);
      d[1] = mma3(L16.A(0, mOT + 5 + kb), xv, d[1]);
    }
  }
  Op16 xs[4];
  // fc3, fc2: v = 2^ex v (1 - act^2), store, multiply by the transposed weights
#pragma unroll
  for (int layer = 2; layer >= 1; --layer) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        d[rb][i] = __builtin_amdgcn_ldexpf(d[rb][i], ex) * (1.f - hv[rb][i] * hv[rb][i]);
        Pdp.st(vr, (layer * kW + rb * 32 + rrow(i)) * pN, d[rb][i]);  // d_pre fc3 / fc2
      }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        hv[rb][i] = Ph.ld(vr, ((layer - 1) * kW + rb * 32 + rrow(i)) * pN);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) e[rb][i] = 0.f;
    ex = scaled_split64(d, xs);
    dense64T_16(e, xs, L16, 0, layer == 2 ? m3T : m2T);
    d[0] = e[0], d[1] = e[1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      d[rb][i] = __builtin_amdgcn_ldexpf(d[rb][i], ex) * (1.f - hv[rb][i] * hv[rb][i]);
      Pdp.st(vr, (rb * 32 + rrow(i)) * pN, d[rb][i]);  // d_pre fc1
    }
  // first layer: 128 outputs, then tanh' from the saved x1
  float xv[4][16];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[rb][i] = Px1.ld(vr, (rb * 32 + rrow(i)) * pN);
  __builtin_amdgcn_sched_barrier(0);
  ex = scaled_split64(d, xs);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) y = mma3(L16.A(0, m1T + rb * 4 + kb), xs[kb], y);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      Pdp.st(vr, (3 * kW + rb * 32 + rrow(i)) * pN,
             __builtin_amdgcn_ldexpf(y[i], ex) * (1.f - xv[rb][i] * xv[rb][i]));  // d_pre first layer
  }
}

int check_wing_horizon(int H) {
  if (H != 10 && H != 20) {
    set_error("the fused fixed-wing policy is built for horizon 10 or 20 (got %d)", H);
    return APG_ERR_ARG;
  }
  return APG_OK;
}

int check_wing_policy(const ApgWingPolicy *pol, int B) {
  if (!pol) { set_error("policy is NULL"); return APG_ERR_ARG; }
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if ((long long)B * 4 * 320 >= (1ll << 32) - 64) {
    set_error("B too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  if (!pol->w_s || !pol->b_s || !pol->w_r || !pol->b_r || !pol->w_1 || !pol->b_1 ||
      !pol->w_2 || !pol->b_2 || !pol->w_3 || !pol->b_3 || !pol->w_out || !pol->b_out) {
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
    return check_launch("hipFuncSetAttribute(wing_policy)");
  return APG_OK;
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_wing_policy_workspace_floats(void) {
  // (+ the packed residual network of a learnt environment's closed loop)
  return (kFwd16Lds > kBwd16Lds ? kFwd16Lds : kBwd16Lds) + kLearntFloats;
}

int apg_wing_policy_fwd(const float *feat, const float *ref_in,
                        const ApgWingPolicy *policy, int B, int H, float *actions,
                        float *x1, float *h, float *workspace,
                        apg_stream_t stream) {
  if (int e = check_wing_policy(policy, B)) return e;
  if (int e = check_wing_horizon(H)) return e;
  if (B == 0) return APG_OK;
  if (!feat || !ref_in || !actions || !x1 || !h || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(wing_policy_fwd_kernel, kFwd16Lds)) return e;
    attr.set();
  }
  Args A = {};
  A.feat = feat, A.ref_in = ref_in, A.actions = actions, A.x1 = x1, A.h = h;
  A.tables = workspace, A.B = B, A.NA = 4 * H;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4 * H;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wing_pack_fwd16_kernel, dim3((kFwd16Lds + 255) / 256), dim3(256), 0,
                     st, P);
  hipLaunchKernelGGL(wing_policy_fwd_kernel,
                     dim3((B + kTrajPerBlock - 1) / kTrajPerBlock), dim3(kThreads),
                     kFwd16Lds * sizeof(float), st, A);
  return check_launch("wing_policy_fwd");
}

int apg_wing_policy_bwd(const float *actions, const float *grad_actions,
                        const float *x1, const float *h,
                        const ApgWingPolicy *policy, int B, int H, float *d_zout,
                        float *d_pre, float *workspace, apg_stream_t stream) {
  if (int e = check_wing_policy(policy, B)) return e;
  if (int e = check_wing_horizon(H)) return e;
  if (B == 0) return APG_OK;
  if (!actions || !grad_actions || !x1 || !h || !d_zout || !d_pre || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(wing_policy_bwd_kernel, kBwd16Lds)) return e;
    attr.set();
  }
  Args A = {};
  A.actions = const_cast<float *>(actions), A.grad_actions = grad_actions;
  A.x1 = const_cast<float *>(x1), A.h = const_cast<float *>(h);
  A.d_zout = d_zout, A.d_pre = d_pre, A.tables = workspace, A.B = B, A.NA = 4 * H;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4 * H;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wing_pack_bwd16_kernel, dim3((kBwd16Lds + 255) / 256), dim3(256), 0,
                     st, P);
  hipLaunchKernelGGL(wing_policy_bwd_kernel,
                     dim3((B + kTrajPerBlock - 1) / kTrajPerBlock), dim3(kThreads),
                     kBwd16Lds * sizeof(float), st, A);
  return check_launch("wing_policy_bwd");
}

int apg_wing_mlp_closed_loop(const float *targets, int n_targets,
                             const float *state0, float dt,
                             const ApgWingParams *params,
                             const ApgWingPolicy *policy, const float *mean,
                             const float *std, float data_dt, int data_horizon,
                             int B, int max_steps, float thresh_div,
                             float thresh_stable, int test_time,
                             float *div_linear, float *div_pass, float *div_fail,
                             int *steps, float *drone, float *seen,
                             float *workspace, apg_stream_t stream) {
  return apg_wing_mlp_closed_loop_env(targets, n_targets, state0, dt, params, nullptr, nullptr,
                                      policy, mean, std, data_dt, data_horizon, B, max_steps,
                                      thresh_div, thresh_stable, test_time, div_linear,
                                      div_pass, div_fail, steps, drone, seen, workspace, stream);
}

int apg_wing_mlp_closed_loop_env(const float *targets, int n_targets, const float *state0,
                                 float dt, const ApgWingParams *params, const float *inertia,
                                 const ApgLearntResidual *learnt, const ApgWingPolicy *policy,
                                 const float *mean, const float *std, float data_dt,
                                 int data_horizon, int B, int max_steps, float thresh_div,
                                 float thresh_stable, int test_time, float *div_linear,
                                 float *div_pass, float *div_fail, int *steps, float *drone,
                                 float *seen, float *workspace, apg_stream_t stream) {
  if (int e = check_wing_policy(policy, B)) return e;
  if ((inertia != nullptr) != (learnt != nullptr)) {
    set_error("learnt environment: `inertia` and `learnt` come together");
    return APG_ERR_ARG;
  }
  if (learnt && (!learnt->w1 || !learnt->b1 || !learnt->w2 || !learnt->b2)) {
    set_error("learnt simulator: weight pointer is NULL");
    return APG_ERR_ARG;
  }
  if (!params || !mean || !std) {
    set_error("params / mean / std is NULL");
    return APG_ERR_ARG;
  }
  if (n_targets < 1 || max_steps < 0 || data_horizon < 1) {
    set_error("n_targets >= 1, max_steps >= 0, data_horizon >= 1 (got %d, %d, %d)",
              n_targets, max_steps, data_horizon);
    return APG_ERR_ARG;
  }
  const long long widest = 16ll * max_steps > 3ll * n_targets ? 16ll * max_steps
                                                              : 3ll * n_targets;
  if ((long long)B * 4 * widest >= (1ll << 32) - 64) {
    set_error("B x max_steps too large for 32-bit plane offsets; split the batch");
    return APG_ERR_ARG;
  }
  if (B == 0 || max_steps == 0) return APG_OK;
  if (!targets || !div_linear || !div_pass || !div_fail || !steps || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static PerDeviceOnce attr;
  if (!attr.test()) {
    if (int e = raise_lds(wing_closed_loop_kernel<false>, kFwd16Lds)) return e;
    if (int e = raise_lds(wing_closed_loop_kernel<true>, kFwd16Lds + kLearntFloats)) return e;
    attr.set();
  }
  WingLoopArgs A = {};
  A.targets = targets, A.state0 = state0, A.div_linear = div_linear;
  A.div_pass = div_pass, A.div_fail = div_fail, A.steps = steps, A.drone = drone;
  A.seen = seen, A.tables = workspace;
  if (learnt) A.k = make_general_const(*params, dt, inertia);
  else static_cast<WingConst &>(A.k) = make_const(*params, dt);
  for (int j = 0; j < kNS; ++j) A.mean[j] = mean[3 + j], A.std[j] = std[3 + j];
  // `ref_vector * vec_len_per_step * (i + 1)`: the Python double 12 * dt enters
  // the float32 tensor product rounded to float32 (dataset.py:312-319)
  A.vec_len = (float)(12.0 * (double)data_dt), A.horizon = (float)data_horizon;
  A.thresh_div = thresh_div, A.thresh_stable = thresh_stable, A.des_speed = 11.5f;
  A.B = B, A.T = max_steps, A.n_targets = n_targets, A.test_time = test_time;
  PackArgs P;
  P.pol = *policy, P.dst = workspace, P.head_rows = 4;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wing_pack_fwd16_kernel, dim3((kFwd16Lds + 255) / 256), dim3(256), 0,
                     st, P);
  const dim3 grid((B + kTrajPerBlock - 1) / kTrajPerBlock);
  if (learnt) {
    hipLaunchKernelGGL(learnt_pack_kernel, dim3((kLearntFloats + 255) / 256), dim3(256), 0, st,
                       *learnt, workspace + kFwd16Lds);
    hipLaunchKernelGGL(wing_closed_loop_kernel<true>, grid, dim3(kThreads),
                       (kFwd16Lds + kLearntFloats) * sizeof(float), st, A);
  } else {
    hipLaunchKernelGGL(wing_closed_loop_kernel<false>, grid, dim3(kThreads),
                       kFwd16Lds * sizeof(float), st, A);
  }
  return check_launch("wing_mlp_closed_loop");
}


}  // extern "C"
