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
// input branches are ONE block-diagonal 12 -> 128 layer.  Weight gradients:
// cotangent planes x activation planes by apg_planes_gemm_grouped.
#include "apg_device.h"
#include "policy_mfma.h"

namespace apg {
namespace {

constexpr int kNS = 9, kNR = 3, kNI = kNS + kNR;  // network inputs (12)
constexpr int kW = 64, kW0 = 2 * kW;              // layer widths (64; first 128)
constexpr int kNA = 80;                           // head width = H x 4 actions
constexpr int kThreads = 512;
constexpr int kTrajPerBlock = kThreads / 2;

// ------------------------------------------------------------------ forward
constexpr int fT0 = 0;                    // [4][16][2] first-layer bias (b_s, b_r)
constexpr int fT1 = fT0 + 128;            // [2][16][2] x 3
constexpr int fT2 = fT1 + 64, fT3 = fT2 + 64;
constexpr int fA0 = 320;                  // [4][6][64]  block-diagonal first layer
constexpr int fA1 = fA0 + 4 * 6 * 64;     // [2][64][64] fc1 (128 inputs)
constexpr int fA2 = fA1 + 2 * 64 * 64;    // [2][32][64]
constexpr int fA3 = fA2 + 2 * 32 * 64;
constexpr int fAo = fA3 + 2 * 32 * 64;    // [3][33][64] head + bias pair
constexpr int kFwdLds = fAo + 3 * 33 * 64;  // 24 576 floats = 98 304 B

struct PackArgs {
  ApgWingPolicy pol;
  float *dst;
};

__global__ __launch_bounds__(256) void wing_pack_fwd_kernel(PackArgs A) {
  const ApgWingPolicy &p = A.pol;
  float *dst = A.dst;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  for (int idx = tid; idx < 4 * 6 * 64; idx += T) {
    const int l = idx & 63, pp = (idx >> 6) % 6, rb = idx / (6 * 64);
    const int m = rb * 32 + (l & 31), k = 2 * pp + (l >> 5);
    float v = 0.f;
    if (m < kW && k < kNS) v = p.w_s[m * kNS + k];
    if (m >= kW && k >= kNS) v = p.w_r[(m - kW) * kNR + (k - kNS)];
    dst[fA0 + idx] = v;
  }
  for (int idx = tid; idx < 2 * 64 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) & 63, rb = idx >> 12;
    dst[fA1 + idx] = p.w_1[(rb * 32 + (l & 31)) * kW0 + kchain(c, l >> 5)];
  }
  for (int idx = tid; idx < 2 * 32 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) & 31, rb = idx >> 11;
    const int m = rb * 32 + (l & 31), k = kchain(c, l >> 5);
    dst[fA2 + idx] = p.w_2[m * kW + k];
    dst[fA3 + idx] = p.w_3[m * kW + k];
  }
  for (int idx = tid; idx < 3 * 33 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) % 33, rb = idx / (33 * 64);
    const int m = rb * 32 + (l & 31);
    float v = 0.f;
    if (m < kNA) v = c < 32 ? p.w_out[m * kW + kchain(c, l >> 5)]
                            : (l < 32 ? p.b_out[m] : 0.f);  // bias pair (1, 0)
    dst[fAo + idx] = v;
  }
  for (int idx = tid; idx < 128; idx += T) {
    const int hi = idx & 1, i = (idx >> 1) & 15, rb = idx >> 5;
    const int row = rb * 32 + rrow(i) + 4 * hi;
    dst[fT0 + idx] = row < kW ? p.b_s[row] : p.b_r[row - kW];
    if (rb < 2) {
      dst[fT1 + idx] = p.b_1[row];
      dst[fT2 + idx] = p.b_2[row];
      dst[fT3 + idx] = p.b_3[row];
    }
  }
}

struct Args {
  const float *feat, *ref_in;   // [9][B], [3][B]
  float *actions;               // [80][B] = [H][4][B]
  const float *grad_actions;    // [80][B] (reverse)
  float *x1, *h;                // [128][B], [192][B]
  float *d_zout, *d_pre;        // [80][B], [320][B]: fc1, fc2, fc3 (64 each), first layer (128)
  const float *tables;
  int B;
};

__global__ __launch_bounds__(kThreads) void wing_policy_fwd_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kFwdLds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pfe(A.feat, kNS, pN), Prf(A.ref_in, kNR, pN);
  const Planes Pac(A.actions, kNA, pN), Px1(A.x1, kW0, pN), Ph(A.h, 3 * kW, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;  // + row 4 hi

  float in[kNI];
#pragma unroll
  for (int j = 0; j < kNS; ++j) in[j] = Pfe.ld(vb, j * pN);
#pragma unroll
  for (int j = 0; j < kNR; ++j) in[kNS + j] = Prf.ld(vb, j * pN);

  // first layer: [states_in 0; 0 ref_in], 12 -> 128
  f32x16 x[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) x[rb][i] = L.T(fT0 + (rb * 16 + i) * 2);
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const float bv = hi ? in[2 * p + 1] : in[2 * p];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) x[rb] = mfma(L.A(fA0 + (rb * 6 + p) * 64), bv, x[rb]);
  }
  // fc1 (tanh of the first layer applied where it is consumed)
  f32x16 u[2], a[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) a[rb][i] = L.T(fT1 + (rb * 16 + i) * 2);
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    const float bv = tanh_fast(x[c >> 4][c & 15]);
    Px1.st(vr, ((c >> 4) * 32 + rrow(c & 15)) * pN, bv);
    a[0] = mfma(L.A(fA1 + (0 * 64 + c) * 64), bv, a[0]);
    a[1] = mfma(L.A(fA1 + (1 * 64 + c) * 64), bv, a[1]);
  }
  // fc2, fc3
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) u[rb][i] = L.T((layer ? fT3 : fT2) + (rb * 16 + i) * 2);
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float bv = tanh_fast(a[c >> 4][c & 15]);
      Ph.st(vr, (layer * kW + (c >> 4) * 32 + rrow(c & 15)) * pN, bv);
      u[0] = mfma(L.A((layer ? fA3 : fA2) + (0 * 32 + c) * 64), bv, u[0]);
      u[1] = mfma(L.A((layer ? fA3 : fA2) + (1 * 32 + c) * 64), bv, u[1]);
    }
    a[0] = u[0], a[1] = u[1];
  }
  // head: 80 outputs in three row blocks, bias as an extra k-pair (1, 0)
  f32x16 z[3];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) z[rb][i] = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float bv = tanh_fast(a[c >> 4][c & 15]);
    Ph.st(vr, (2 * kW + (c >> 4) * 32 + rrow(c & 15)) * pN, bv);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb) z[rb] = mfma(L.A(fAo + (rb * 33 + c) * 64), bv, z[rb]);
  }
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
    z[rb] = mfma(L.A(fAo + (rb * 33 + 32) * 64), hi ? 0.f : 1.f, z[rb]);
  // actions = sigmoid(z): accumulator rows ARE the action planes
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (rb * 32 + rrow(i) + 4 < kNA)  // rows r(i), r(i) + 4 both < 80
        Pac.st(vr, (rb * 32 + rrow(i)) * pN, sigmoidf_(z[rb][i]));
}

// ------------------------------------------------------------------ reverse
constexpr int rAo = 0;                    // [2][40][64] head^T
constexpr int rA3 = rAo + 2 * 40 * 64;    // [2][32][64] fc3^T
constexpr int rA2 = rA3 + 2 * 32 * 64;    // [2][32][64] fc2^T
constexpr int rA1 = rA2 + 2 * 32 * 64;    // [4][32][64] fc1^T (128 outputs)
constexpr int kBwdLds = rA1 + 4 * 32 * 64;  // 21 504 floats = 86 016 B

// k index of head k-pair c (accumulator layout of the 80 outputs: row blocks
// 0 and 1 registers 0..15, row block 2 registers 0..7)
__host__ __device__ constexpr int khead(int c, int hi) {
  return (c >> 4) * 32 + rrow(c & 15) + 4 * hi;
}

__global__ __launch_bounds__(256) void wing_pack_bwd_kernel(PackArgs A) {
  const ApgWingPolicy &p = A.pol;
  float *dst = A.dst;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  for (int idx = tid; idx < 2 * 40 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) % 40, rb = idx / (40 * 64);
    dst[rAo + idx] = p.w_out[khead(c, l >> 5) * kW + rb * 32 + (l & 31)];
  }
  for (int idx = tid; idx < 2 * 32 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) & 31, rb = idx >> 11;
    const int m = rb * 32 + (l & 31), k = kchain(c, l >> 5);
    dst[rA3 + idx] = p.w_3[k * kW + m];
    dst[rA2 + idx] = p.w_2[k * kW + m];
  }
  for (int idx = tid; idx < 4 * 32 * 64; idx += T) {
    const int l = idx & 63, c = (idx >> 6) & 31, rb = idx >> 11;
    dst[rA1 + idx] = p.w_1[kchain(c, l >> 5) * kW0 + rb * 32 + (l & 31)];
  }
}

__global__ __launch_bounds__(kThreads) void wing_policy_bwd_kernel(Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fill_lds(lds, A.tables, kBwdLds);
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const LdsView L(lds, lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = (blockIdx.x * (kThreads / 64) + wave) * 32 + (lane & 31);
  const int B = A.B;
  const bool live = b < B;
  const unsigned pN = (unsigned)B * 4u;
  const Planes Pac(A.actions, kNA, pN), Pga(A.grad_actions, kNA, pN);
  const Planes Px1(A.x1, kW0, pN), Ph(A.h, 3 * kW, pN);
  const Planes Pdz(A.d_zout, kNA, pN), Pdp(A.d_pre, 3 * kW + kW0, pN);
  const unsigned vb = live ? (unsigned)b * 4u : kDead;
  const unsigned vr = live ? vb + (hi ? 4u * pN : 0u) : kDead;

  // dL/dz = dL/da * a (1 - a), 40 k-pairs in accumulator layout
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
  f32x16 d[2], e[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) d[rb][i] = 0.f;
#pragma unroll
  for (int c = 0; c < 40; ++c) {
    d[0] = mfma(L.A(rAo + (0 * 40 + c) * 64), dz[c], d[0]);
    d[1] = mfma(L.A(rAo + (1 * 40 + c) * 64), dz[c], d[1]);
  }
  // fc3, fc2: v *= 1 - act^2, store, multiply by the transposed weights
#pragma unroll
  for (int layer = 2; layer >= 1; --layer) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        d[rb][i] *= 1.f - hv[rb][i] * hv[rb][i];
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
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float bv = d[c >> 4][c & 15];
      e[0] = mfma(L.A((layer == 2 ? rA3 : rA2) + (0 * 32 + c) * 64), bv, e[0]);
      e[1] = mfma(L.A((layer == 2 ? rA3 : rA2) + (1 * 32 + c) * 64), bv, e[1]);
    }
    d[0] = e[0], d[1] = e[1];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      d[rb][i] *= 1.f - hv[rb][i] * hv[rb][i];
      Pdp.st(vr, (rb * 32 + rrow(i)) * pN, d[rb][i]);  // d_pre fc1
    }
  // first layer: 128 outputs, then tanh' from the saved x1
  float xv[4][16];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[rb][i] = Px1.ld(vr, (rb * 32 + rrow(i)) * pN);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c)
      y = mfma(L.A(rA1 + (rb * 32 + c) * 64), d[c >> 4][c & 15], y);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      Pdp.st(vr, (3 * kW + rb * 32 + rrow(i)) * pN,
             y[i] * (1.f - xv[rb][i] * xv[rb][i]));  // d_pre first layer
  }
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
  return kFwdLds > kBwdLds ? kFwdLds : kBwdLds;
}

int apg_wing_policy_fwd(const float *feat, const float *ref_in,
                        const ApgWingPolicy *policy, int B, float *actions,
                        float *x1, float *h, float *workspace,
                        apg_stream_t stream) {
  if (int e = check_wing_policy(policy, B)) return e;
  if (B == 0) return APG_OK;
  if (!feat || !ref_in || !actions || !x1 || !h || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static bool attr = false;
  if (!attr) {
    if (int e = raise_lds(wing_policy_fwd_kernel, kFwdLds)) return e;
    attr = true;
  }
  Args A = {};
  A.feat = feat, A.ref_in = ref_in, A.actions = actions, A.x1 = x1, A.h = h;
  A.tables = workspace, A.B = B;
  PackArgs P;
  P.pol = *policy, P.dst = workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wing_pack_fwd_kernel, dim3((kFwdLds + 255) / 256), dim3(256), 0,
                     st, P);
  hipLaunchKernelGGL(wing_policy_fwd_kernel,
                     dim3((B + kTrajPerBlock - 1) / kTrajPerBlock), dim3(kThreads),
                     kFwdLds * sizeof(float), st, A);
  return check_launch("wing_policy_fwd");
}

int apg_wing_policy_bwd(const float *actions, const float *grad_actions,
                        const float *x1, const float *h,
                        const ApgWingPolicy *policy, int B, float *d_zout,
                        float *d_pre, float *workspace, apg_stream_t stream) {
  if (int e = check_wing_policy(policy, B)) return e;
  if (B == 0) return APG_OK;
  if (!actions || !grad_actions || !x1 || !h || !d_zout || !d_pre || !workspace) {
    set_error("NULL buffer");
    return APG_ERR_ARG;
  }
  static bool attr = false;
  if (!attr) {
    if (int e = raise_lds(wing_policy_bwd_kernel, kBwdLds)) return e;
    attr = true;
  }
  Args A = {};
  A.actions = const_cast<float *>(actions), A.grad_actions = grad_actions;
  A.x1 = const_cast<float *>(x1), A.h = const_cast<float *>(h);
  A.d_zout = d_zout, A.d_pre = d_pre, A.tables = workspace, A.B = B;
  PackArgs P;
  P.pol = *policy, P.dst = workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wing_pack_bwd_kernel, dim3((kBwdLds + 255) / 256), dim3(256), 0,
                     st, P);
  hipLaunchKernelGGL(wing_policy_bwd_kernel,
                     dim3((B + kTrajPerBlock - 1) / kTrajPerBlock), dim3(kThreads),
                     kBwdLds * sizeof(float), st, A);
  return check_launch("wing_policy_bwd");
}

}  // extern "C"
