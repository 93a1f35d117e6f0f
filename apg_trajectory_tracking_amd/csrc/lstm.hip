// lstm.hip - K7: the LSTM-policy quadrotor unroll with the policy INSIDE the
// kernel (BASELINE config 5: quadrotor, LSTM recurrent mode, H = 10).
//
// Replaces, for train_mode == "LSTM", the loop of
//   TrainDrone.train_recurrent_model       scripts/train_drone.py:113-173
//   LSTM_NEW.forward (conv branch)         neural_control/models/rnn.py:35-51
//   state_preprocessing                    neural_control/dataset.py:207-220
//   FlightmareDynamics / quad_mpc_loss     (see quad.hip)
// with two launches: a forward sweep that runs policy + dynamics for all H
// steps of a trajectory in one lane, and a reverse sweep (BPTT through LSTM
// cell, conv window, feature construction and dynamics).  Pinned window
// semantics: window_k = in_ref[k : k+H], its position
// columns made relative to the CURRENT position, copied - never in place.
//
// One lane = one trajectory.  The 6.5 k policy weights (26 KB) live in LDS;
// every lane of a wave reads the same weight at the same time (LDS broadcast,
// conflict-free); hidden / cell state, the sliding reference window (90
// floats, one new row per step) and the 32 gate accumulators stay in
// registers.  ~10.3 k FMA per env-step forward: VALU-bound, no MFMA (a
// 64 x 183 x 32 per-wave GEMM per step would fit MFMA, but the recurrence
// serialises the steps and the conv/relu producer is elementwise; left for a
// later round).
//
// Weight gradients are NOT accumulated per lane: the reverse sweep writes the
// per-(step, trajectory) cotangents (gate pre-activations, head
// pre-activations, conv pre-activations) next to the forward's saved inputs,
// all as [feature][H*B] planes, and the host turns them into dW with a few
// plain GEMMs over the H*B rows (rocBLAS through torch.matmul).
#include "apg_device.h"
#include "quad_math.h"

namespace apg {
namespace {

constexpr int kH = 10;            // horizon == window length
constexpr int kRD = 9;            // reference columns fed to the policy
constexpr int kNF = 15;           // state features
constexpr int kNC = 20;           // conv output channels
constexpr int kNP = kH - 2;       // conv output positions (kernel 3)
constexpr int kNX = kNF + kNC * kNP;  // LSTM input width (175)
constexpr int kNH = 8;            // hidden units
constexpr int kNG = 4 * kNH;      // gate pre-activations (i, f, g, o)
constexpr int kBlock = 128;

// LDS image of the policy (floats)
constexpr int oWc = 0;                       // [20][27]
constexpr int oBc = oWc + kNC * 27;          // [20]
constexpr int oWih = oBc + kNC;              // [175][32]  (transposed)
constexpr int oWhh = oWih + kNX * kNG;       // [8][32]    (transposed)
constexpr int oBg = oWhh + kNH * kNG;        // [32] = b_ih + b_hh
constexpr int oWo = oBg + kNG;               // [4][8]
constexpr int oBo = oWo + 4 * kNH;           // [4]
constexpr int oA = oBo + 4;                  // [20][3] = sum_t Wc[ch][c][t], c < 3
constexpr int kLdsFloats = oA + kNC * 3;

__device__ __forceinline__ void load_policy(float *lds, const ApgLstmPolicy &p) {
  for (int i = threadIdx.x; i < kNC * 27; i += blockDim.x) lds[oWc + i] = p.conv_w[i];
  for (int i = threadIdx.x; i < kNC; i += blockDim.x) lds[oBc + i] = p.conv_b[i];
  for (int i = threadIdx.x; i < kNX * kNG; i += blockDim.x) lds[oWih + i] = p.w_ih_t[i];
  for (int i = threadIdx.x; i < kNH * kNG; i += blockDim.x) lds[oWhh + i] = p.w_hh_t[i];
  for (int i = threadIdx.x; i < kNG; i += blockDim.x) lds[oBg + i] = p.b_gates[i];
  for (int i = threadIdx.x; i < 4 * kNH; i += blockDim.x) lds[oWo + i] = p.w_out[i];
  for (int i = threadIdx.x; i < 4; i += blockDim.x) lds[oBo + i] = p.b_out[i];
  for (int i = threadIdx.x; i < kNC * 3; i += blockDim.x) {
    const int ch = i / 3, c = i % 3;
    lds[oA + i] = p.conv_w[ch * 27 + c * 3] + p.conv_w[ch * 27 + c * 3 + 1] +
                  p.conv_w[ch * 27 + c * 3 + 2];
  }
  __syncthreads();
}

__device__ __forceinline__ float sigmoidf_(float x) {
  return 1.0f / (1.0f + expf(-x));
}

// acc[0..31] += row[0..31] * v   (row is wave-uniform: LDS broadcast reads)
__device__ __forceinline__ void axpy32(float (&acc)[kNG], const float *row, float v) {
#pragma unroll
  for (int q = 0; q < kNG / 4; ++q) {
    const float4 w = *reinterpret_cast<const float4 *>(row + 4 * q);
    acc[4 * q + 0] = fmaf(w.x, v, acc[4 * q + 0]);
    acc[4 * q + 1] = fmaf(w.y, v, acc[4 * q + 1]);
    acc[4 * q + 2] = fmaf(w.z, v, acc[4 * q + 2]);
    acc[4 * q + 3] = fmaf(w.w, v, acc[4 * q + 3]);
  }
}

// four independent partial sums: a single accumulator would make the 32 FMAs
// one dependent chain (latency-, not issue-bound, with one wave per SIMD)
__device__ __forceinline__ float dot32(const float4 (&w)[kNG / 4],
                                       const float (&d)[kNG]) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int q = 0; q < kNG / 4; ++q) {
    a0 = fmaf(w[q].x, d[4 * q + 0], a0);
    a1 = fmaf(w[q].y, d[4 * q + 1], a1);
    a2 = fmaf(w[q].z, d[4 * q + 2], a2);
    a3 = fmaf(w[q].w, d[4 * q + 3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ float dot32(const float *row, const float (&d)[kNG]) {
  float4 w[kNG / 4];
#pragma unroll
  for (int q = 0; q < kNG / 4; ++q)
    w[q] = *reinterpret_cast<const float4 *>(row + 4 * q);
  return dot32(w, d);
}

struct FwdArgs {
  const float *state0, *in_ref, *h0, *c0;
  float *states, *actions, *x, *gates, *hc, *hnew;
  unsigned *mask;  // [5][N] relu mask bits of the conv outputs
  ApgLstmPolicy pol;
  QuadConst c;
  int B;
};

__global__ __launch_bounds__(kBlock) void lstm_rollout_fwd_kernel(FwdArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  load_policy(lds, A.pol);
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= A.B) return;
  const int B = A.B;
  const size_t N = (size_t)kH * B;
  const QuadConst c = A.c;

  float s[12], h[kNH], cell[kNH];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = A.state0[(size_t)i * B + b];
#pragma unroll
  for (int m = 0; m < kNH; ++m) h[m] = A.h0[(size_t)m * B + b], cell[m] = A.c0[(size_t)m * B + b];
  float w[kH][kRD];  // sliding reference window, raw (absolute) values
#pragma unroll
  for (int r = 0; r < kH; ++r)
#pragma unroll
    for (int q = 0; q < kRD; ++q) w[r][q] = A.in_ref[((size_t)r * kRD + q) * B + b];

#pragma unroll 1
  for (int k = 0; k < kH; ++k) {
    const size_t n = (size_t)k * B + b;
    const Trig t = make_trig(&s[3]);
    float feat[kNF];
    quad_features(s, t, feat);
    float g[kNG];
#pragma unroll
    for (int q = 0; q < kNG; ++q) g[q] = lds[oBg + q];
#pragma unroll
    for (int j = 0; j < kNF; ++j) {
      A.x[(size_t)j * N + n] = feat[j];
      axpy32(g, &lds[oWih + j * kNG], feat[j]);
    }
#pragma unroll
    for (int m = 0; m < kNH; ++m) {
      A.hc[(size_t)m * N + n] = h[m];
      A.hc[(size_t)(kNH + m) * N + n] = cell[m];
      axpy32(g, &lds[oWhh + m * kNG], h[m]);
    }
    // position columns relative to the current position
    float wr[kH][3];
#pragma unroll
    for (int r = 0; r < kH; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) wr[r][q] = w[r][q] - s[q];
    unsigned mask[5] = {0u, 0u, 0u, 0u, 0u};  // relu mask of the 160 conv outputs
    float4 wrow[kNG / 4];                      // W_ih row of the next element
#pragma unroll
    for (int q = 0; q < kNG / 4; ++q)
      wrow[q] = *reinterpret_cast<const float4 *>(&lds[oWih + kNF * kNG + 4 * q]);
#pragma unroll 1
    for (int ch = 0; ch < kNC; ++ch) {
      float wc[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) wc[i] = lds[oWc + ch * 27 + i];
      const float bias = lds[oBc + ch];
      unsigned bits = 0u;
#pragma unroll
      for (int pos = 0; pos < kNP; ++pos) {
        const int j = kNF + ch * kNP + pos;
        // this element's weights were requested one element ago; request the
        // next row now so that its LDS latency hides behind 59 FMAs
        float4 wcur[kNG / 4];
#pragma unroll
        for (int q = 0; q < kNG / 4; ++q) wcur[q] = wrow[q];
        const int jn = j + 1 < kNX ? j + 1 : j;
#pragma unroll
        for (int q = 0; q < kNG / 4; ++q)
          wrow[q] = *reinterpret_cast<const float4 *>(&lds[oWih + jn * kNG + 4 * q]);
        float v = bias;
#pragma unroll
        for (int q = 0; q < kRD; ++q)
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
            v = fmaf(wc[q * 3 + tt], q < 3 ? wr[pos + tt][q] : w[pos + tt][q], v);
        bits |= (v > 0.f ? 1u : 0u) << pos;
        v = fmaxf(v, 0.f);
        A.x[(size_t)j * N + n] = v;
#pragma unroll
        for (int q = 0; q < kNG / 4; ++q) {
          g[4 * q + 0] = fmaf(wcur[q].x, v, g[4 * q + 0]);
          g[4 * q + 1] = fmaf(wcur[q].y, v, g[4 * q + 1]);
          g[4 * q + 2] = fmaf(wcur[q].z, v, g[4 * q + 2]);
          g[4 * q + 3] = fmaf(wcur[q].w, v, g[4 * q + 3]);
        }
      }
      mask[ch >> 2] |= bits << (8 * (ch & 3));
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
      A.mask[(size_t)i * N + n] = mask[i];
    // LSTMCell (torch gate order i, f, g, o)
    float hn[kNH];
#pragma unroll
    for (int m = 0; m < kNH; ++m) {
      const float gi = sigmoidf_(g[m]), gf = sigmoidf_(g[kNH + m]);
      const float gg = tanhf(g[2 * kNH + m]), go = sigmoidf_(g[3 * kNH + m]);
      A.gates[(size_t)m * N + n] = gi;
      A.gates[(size_t)(kNH + m) * N + n] = gf;
      A.gates[(size_t)(2 * kNH + m) * N + n] = gg;
      A.gates[(size_t)(3 * kNH + m) * N + n] = go;
      cell[m] = gf * cell[m] + gi * gg;
      hn[m] = go * tanhf(cell[m]);
      h[m] = hn[m];
      A.hnew[(size_t)m * N + n] = hn[m];
    }
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float z = lds[oBo + j];
#pragma unroll
      for (int m = 0; m < kNH; ++m) z = fmaf(lds[oWo + j * kNH + m], hn[m], z);
      a[j] = sigmoidf_(z);
      A.actions[((size_t)k * 4 + j) * B + b] = a[j];
    }
    quad_step(s, a, c, t);
#pragma unroll
    for (int i = 0; i < 12; ++i) A.states[((size_t)k * 12 + i) * B + b] = s[i];
    // slide the window: drop row 0, fetch in_ref row k + H
    if (k + 1 < kH) {
#pragma unroll
      for (int r = 0; r + 1 < kH; ++r)
#pragma unroll
        for (int q = 0; q < kRD; ++q) w[r][q] = w[r + 1][q];
#pragma unroll
      for (int q = 0; q < kRD; ++q)
        w[kH - 1][q] = A.in_ref[((size_t)(k + kH) * kRD + q) * B + b];
    }
  }
}

struct BwdArgs {
  const float *state0, *states, *actions, *ref, *gates, *hc;
  const unsigned *mask;
  float *loss_partials, *d_gates, *d_zout, *d_conv;
  float *grad_state0, *grad_h0, *grad_c0;
  ApgLstmPolicy pol;
  QuadConst c;
  ApgQuadLossWeights w;
  int B, ref_cols, vel_col;
};

__global__ __launch_bounds__(kBlock) void lstm_rollout_bwd_kernel(BwdArgs A) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  load_policy(lds, A.pol);
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;
  const int B = A.B;
  const size_t N = (size_t)kH * B;
  const QuadConst c = A.c;

  float lam[12], dh[kNH], dc[kNH];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int m = 0; m < kNH; ++m) dh[m] = 0.f, dc[m] = 0.f;
  float loss = 0.f;

#pragma unroll 1
  for (int k = kH - 1; k >= 0; --k) {
    const size_t n = (size_t)k * B + bb;
    float sn[12], sc[12], a[4], rp[3], rv[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      sn[i] = A.states[((size_t)k * 12 + i) * B + bb];
      sc[i] = k > 0 ? A.states[((size_t)(k - 1) * 12 + i) * B + bb]
                    : A.state0[(size_t)i * B + bb];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = A.actions[((size_t)k * 4 + j) * B + bb];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      rp[i] = A.ref[((size_t)k * A.ref_cols + i) * B + bb];
      rv[i] = A.ref[((size_t)k * A.ref_cols + A.vel_col + i) * B + bb];
    }
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
      if (live) A.d_zout[(size_t)j * N + n] = dz[j];
    }
    // LSTM cell
    float dG[kNG];
#pragma unroll
    for (int m = 0; m < kNH; ++m) {
      const float gi = A.gates[(size_t)m * N + n];
      const float gf = A.gates[(size_t)(kNH + m) * N + n];
      const float gg = A.gates[(size_t)(2 * kNH + m) * N + n];
      const float go = A.gates[(size_t)(3 * kNH + m) * N + n];
      const float cp = A.hc[(size_t)(kNH + m) * N + n];
      const float tc = tanhf(gf * cp + gi * gg);
      float dht = dh[m];
#pragma unroll
      for (int j = 0; j < 4; ++j) dht = fmaf(lds[oWo + j * kNH + m], dz[j], dht);
      const float dct = dc[m] + dht * go * (1.f - tc * tc);
      dG[m] = dct * gg * gi * (1.f - gi);
      dG[kNH + m] = dct * cp * gf * (1.f - gf);
      dG[2 * kNH + m] = dct * gi * (1.f - gg * gg);
      dG[3 * kNH + m] = dht * tc * go * (1.f - go);
      dc[m] = dct * gf;
    }
    if (live) {
#pragma unroll
      for (int q = 0; q < kNG; ++q) A.d_gates[(size_t)q * N + n] = dG[q];
    }
#pragma unroll
    for (int m = 0; m < kNH; ++m) dh[m] = dot32(&lds[oWhh + m * kNG], dG);
    // state features
    float dfeat[kNF], gs[12];
#pragma unroll
    for (int j = 0; j < kNF; ++j) dfeat[j] = dot32(&lds[oWih + j * kNG], dG);
    quad_features_adjoint(sc, t, dfeat, gs);
#pragma unroll
    for (int i = 0; i < 12; ++i) lam[i] += gs[i];
    // conv branch: relu mask from the saved activations; only the position
    // columns of the window carry a gradient (rel = ref - pos)
    unsigned mask[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) mask[i] = A.mask[(size_t)i * N + n];
    float4 wrow[kNG / 4];  // W_ih row of the next conv element (prefetched)
#pragma unroll
    for (int q = 0; q < kNG / 4; ++q)
      wrow[q] = *reinterpret_cast<const float4 *>(&lds[oWih + kNF * kNG + 4 * q]);
#pragma unroll 1
    for (int ch = 0; ch < kNC; ++ch) {
      float sum = 0.f;
      const unsigned bits = (mask[ch >> 2] >> (8 * (ch & 3))) & 0xffu;
#pragma unroll
      for (int pos = 0; pos < kNP; ++pos) {
        const int j = kNF + ch * kNP + pos;
        float4 wcur[kNG / 4];
#pragma unroll
        for (int q = 0; q < kNG / 4; ++q) wcur[q] = wrow[q];
        const int jn = j + 1 < kNX ? j + 1 : j;
#pragma unroll
        for (int q = 0; q < kNG / 4; ++q)
          wrow[q] = *reinterpret_cast<const float4 *>(&lds[oWih + jn * kNG + 4 * q]);
        const float dxe = dot32(wcur, dG);
        const float dcp = ((bits >> pos) & 1u) ? dxe : 0.f;
        if (live) A.d_conv[(size_t)(ch * kNP + pos) * N + n] = dcp;
        sum += dcp;
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) lam[q] -= lds[oA + ch * 3 + q] * sum;
    }
  }
  if (live) {
    if (A.grad_state0)
#pragma unroll
      for (int i = 0; i < 12; ++i) A.grad_state0[(size_t)i * B + b] = lam[i];
    if (A.grad_h0)
#pragma unroll
      for (int m = 0; m < kNH; ++m) A.grad_h0[(size_t)m * B + b] = dh[m];
    if (A.grad_c0)
#pragma unroll
      for (int m = 0; m < kNH; ++m) A.grad_c0[(size_t)m * B + b] = dc[m];
  }
  // one loss partial per wave (kBlock / 64 waves per workgroup)
  write_wave_partial(A.loss_partials, live ? loss : 0.f);
}

int check_lstm(const ApgQuadParams *params, const ApgLstmPolicy *pol, int B, int H) {
  if (!params || !pol) { set_error("params / policy is NULL"); return APG_ERR_ARG; }
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if (H != kH) {
    set_error("the fused LSTM rollout is built for horizon %d (got %d)", kH, H);
    return APG_ERR_ARG;
  }
  if (!pol->conv_w || !pol->conv_b || !pol->w_ih_t || !pol->w_hh_t ||
      !pol->b_gates || !pol->w_out || !pol->b_out) {
    set_error("policy weight pointer is NULL");
    return APG_ERR_ARG;
  }
  return APG_OK;
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_quad_lstm_rollout_fwd(const float *state0, const float *in_ref,
                              const float *h0, const float *c0, float dt,
                              const ApgQuadParams *params,
                              const ApgLstmPolicy *policy, int B, int H,
                              float *states, float *actions, float *x,
                              float *gates, float *hc, float *hnew,
                              unsigned *relu_mask, apg_stream_t stream) {
  if (int e = check_lstm(params, policy, B, H)) return e;
  if (B == 0) return APG_OK;
  if (!state0 || !in_ref || !h0 || !c0 || !states || !actions || !x || !gates ||
      !hc || !hnew || !relu_mask) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  FwdArgs A;
  A.state0 = state0, A.in_ref = in_ref, A.h0 = h0, A.c0 = c0;
  A.states = states, A.actions = actions, A.x = x, A.gates = gates, A.hc = hc;
  A.hnew = hnew;
  A.mask = relu_mask;
  A.pol = *policy;
  A.c = make_const(*params, dt);
  A.B = B;
  hipLaunchKernelGGL(lstm_rollout_fwd_kernel, dim3((B + kBlock - 1) / kBlock),
                     dim3(kBlock), 0, (hipStream_t)stream, A);
  return check_launch("quad_lstm_rollout_fwd");
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
                              float *grad_h0, float *grad_c0,
                              apg_stream_t stream) {
  if (int e = check_lstm(params, policy, B, H)) return e;
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
  if (!state0 || !states || !actions || !ref || !relu_mask || !gates || !hc ||
      !loss_partials || !d_gates || !d_zout || !d_conv) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  BwdArgs A;
  A.state0 = state0, A.states = states, A.actions = actions, A.ref = ref;
  A.mask = relu_mask, A.gates = gates, A.hc = hc;
  A.loss_partials = loss_partials, A.d_gates = d_gates, A.d_zout = d_zout;
  A.d_conv = d_conv, A.grad_state0 = grad_state0, A.grad_h0 = grad_h0;
  A.grad_c0 = grad_c0;
  A.pol = *policy;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  const int blocks = (B + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(lstm_rollout_bwd_kernel, dim3(blocks), dim3(kBlock), 0, st, A);
  if (int e = check_launch("quad_lstm_rollout_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, blocks * (kBlock / kWave), loss, st);
  return APG_OK;
}

}  // extern "C"
