// cartpole.hip - cart-pole kernels: single step (+VJP) and the fused rollout
// of TrainCartpole.run_epoch's controller branch.
//
// Arithmetic restated from (paths relative to the reference repo):
//   neural_control/dynamics/cartpole_dynamics.py:53-119 (params :23-43)
//   scripts/train_cartpole.py:103-110 (make_reference), :131-150 (unroll)
//   neural_control/drone_loss.py:136-145 (cartpole_loss_mpc)
// BASELINE config 1 (B = 64, H = 5) is launch-latency bound: one wave does
// the whole batch; the kernel exists for parity and API completeness.
#include "apg_device.h"
#include "cartpole_math.h"

namespace apg {
namespace {

template <int LAYOUT>
__global__ __launch_bounds__(256) void cart_step_fwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    CartConst c, int B, float *__restrict__ next) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[4];
  load_state<LAYOUT, 4>(state, B, b, s);
  cart_step(s, action[b], c);
  store_state<LAYOUT, 4>(next, B, b, s);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void cart_step_bwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    CartConst c, int B, const float *__restrict__ grad_next,
    float *__restrict__ grad_state, float *__restrict__ grad_action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[4], lam[4];
  load_state<LAYOUT, 4>(state, B, b, s);
  load_state<LAYOUT, 4>(grad_next, B, b, lam);
  const float xd = s[1], thd = s[3];
  CartAux x = cart_step(s, action[b], c);
  const float ga = cart_step_adjoint(lam, xd, thd, x, c);
  if (grad_state) store_state<LAYOUT, 4>(grad_state, B, b, lam);
  if (grad_action) grad_action[b] = ga;
}

struct CartRolloutArgs {
  const float *state0, *actions;
  float *loss_partials, *grad_actions, *grad_state0, *states_out;
  CartConst c;
  int B, H;
};

// Per-step stash in LDS as [k][4][lane]: the pre-step state.
template <int LAYOUT>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) void cart_rollout_kernel(
    CartRolloutArgs A) {
  extern __shared__ float stash[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;
  const CartConst c = A.c;
  const int H = A.H;
  const float wq[4] = {0.f, 3.f, 10.f, 1.f};  // drone_loss.py:136
  // make_reference: ref_k = s0 * (1 - 1/(H-1) * k), k < H-1; last row zero
  const double inv = H > 1 ? 1.0 / (double)(H - 1) : 0.0;
  auto ST = [&](int k, int i) -> float & {
    return stash[(k * 4 + i) * APG_ROLLOUT_BLOCK + lane];
  };
  float s0[4], s[4];
  load_state<LAYOUT, 4>(A.state0, A.B, bb, s0);
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = s0[i];
  float loss = 0.f;
  for (int k = 0; k < H; ++k) {
    float a[1];
    load_seq<LAYOUT, 1>(A.actions, A.B, H, 1, bb, k, 0, a);
#pragma unroll
    for (int i = 0; i < 4; ++i) ST(k, i) = s[i];
    cart_step(s, a[0], c);
    if (A.states_out && live)
      store_seq<LAYOUT, 4>(A.states_out, A.B, H, 4, b, k, 0, s);
    const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float d = s[i] - s0[i] * f;
      loss += (d * d) * wq[i];
    }
    loss += 0.01f * a[0] * a[0];
  }
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);

  float lam[4] = {0.f, 0.f, 0.f, 0.f}, g0[4] = {0.f, 0.f, 0.f, 0.f};
  float nxt[4] = {s[0], s[1], s[2], s[3]};
  for (int k = H - 1; k >= 0; --k) {
    float a[1], pre[4];
    load_seq<LAYOUT, 1>(A.actions, A.B, H, 1, bb, k, 0, a);
    const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pre[i] = ST(k, i);
      float seed = 2.f * wq[i] * (nxt[i] - s0[i] * f);
      lam[i] += seed;
      g0[i] -= seed * f;  // gradient through make_reference
    }
    float tmp[4] = {pre[0], pre[1], pre[2], pre[3]};
    CartAux x = cart_step(tmp, a[0], c);
    float ga[1] = {cart_step_adjoint(lam, pre[1], pre[3], x, c) +
                   0.02f * a[0]};
    if (live) store_seq<LAYOUT, 1>(A.grad_actions, A.B, H, 1, b, k, 0, ga);
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[i] = pre[i];
  }
  if (A.grad_state0 && live) {
#pragma unroll
    for (int i = 0; i < 4; ++i) lam[i] += g0[i];
    store_state<LAYOUT, 4>(A.grad_state0, A.B, b, lam);
  }
}

// no-grad unroll (evaluation): states only
template <int LAYOUT>
__global__ __launch_bounds__(256) void cart_rollout_fwd_kernel(
    const float *__restrict__ state0, const float *__restrict__ actions,
    CartConst c, int B, int H, float *__restrict__ states_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[4];
  load_state<LAYOUT, 4>(state0, B, b, s);
  for (int k = 0; k < H; ++k) {
    float a[1];
    load_seq<LAYOUT, 1>(actions, B, H, 1, b, k, 0, a);
    cart_step(s, a[0], c);
    store_seq<LAYOUT, 4>(states_out, B, H, 4, b, k, 0, s);
  }
}

inline int grid_for(int B, int block) { return (B + block - 1) / block; }

int check_args(const void *p0, const void *p1, const void *params, int B,
               int layout) {
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if (layout != APG_LAYOUT_SOA && layout != APG_LAYOUT_AOS) {
    set_error("unknown layout %d", layout);
    return APG_ERR_ARG;
  }
  if (!params) { set_error("params is NULL"); return APG_ERR_ARG; }
  if (B > 0 && (!p0 || !p1)) { set_error("NULL input pointer"); return APG_ERR_ARG; }
  return APG_OK;
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_cartpole_step_fwd(const float *state, const float *action, float dt,
                          const ApgCartpoleParams *params, int B, int layout,
                          float *next_state, apg_stream_t stream) {
  if (int e = check_args(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!next_state) { set_error("next_state is NULL"); return APG_ERR_ARG; }
  CartConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(cart_step_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, next_state);
  else
    hipLaunchKernelGGL(cart_step_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, next_state);
  return check_launch("cartpole_step_fwd");
}

int apg_cartpole_step_bwd(const float *state, const float *action, float dt,
                          const ApgCartpoleParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action, apg_stream_t stream) {
  if (int e = check_args(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!grad_next) { set_error("grad_next is NULL"); return APG_ERR_ARG; }
  CartConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(cart_step_bwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, grad_next,
                       grad_state, grad_action);
  else
    hipLaunchKernelGGL(cart_step_bwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, grad_next,
                       grad_state, grad_action);
  return check_launch("cartpole_step_bwd");
}

int apg_cartpole_rollout_fwd_bwd(const float *state0, const float *actions,
                                 float dt, const ApgCartpoleParams *params,
                                 int B, int H, int layout,
                                 float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out, apg_stream_t stream) {
  if (int e = check_args(state0, actions, params, B, layout)) return e;
  if (H < 1 || H > APG_MAX_HORIZON) {
    set_error("H must be in [1, %d] (got %d)", APG_MAX_HORIZON, H);
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!loss_partials || !grad_actions) {
    set_error("loss_partials / grad_actions must not be NULL");
    return APG_ERR_ARG;
  }
  CartRolloutArgs A;
  A.state0 = state0, A.actions = actions;
  A.loss_partials = loss_partials, A.grad_actions = grad_actions;
  A.grad_state0 = grad_state0, A.states_out = states_out;
  A.c = make_const(*params, dt);
  A.B = B, A.H = H;
  const size_t lds = (size_t)H * 4 * APG_ROLLOUT_BLOCK * sizeof(float);
  const dim3 grid(grid_for(B, APG_ROLLOUT_BLOCK)), block(APG_ROLLOUT_BLOCK);
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(cart_rollout_kernel<APG_LAYOUT_SOA>, grid, block, lds, st, A);
  else
    hipLaunchKernelGGL(cart_rollout_kernel<APG_LAYOUT_AOS>, grid, block, lds, st, A);
  if (int e = check_launch("cartpole_rollout_fwd_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
  return APG_OK;
}

int apg_cartpole_rollout_fwd(const float *state0, const float *actions, float dt,
                             const ApgCartpoleParams *params, int B, int H,
                             int layout, float *states_out, apg_stream_t stream) {
  if (int e = check_args(state0, actions, params, B, layout)) return e;
  if (H < 1) { set_error("H must be >= 1 (got %d)", H); return APG_ERR_ARG; }
  if (B == 0) return APG_OK;
  if (!states_out) { set_error("states_out is NULL"); return APG_ERR_ARG; }
  const CartConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(cart_rollout_fwd_kernel<APG_LAYOUT_SOA>,
                       dim3(grid_for(B, 256)), dim3(256), 0, st, state0, actions,
                       c, B, H, states_out);
  else
    hipLaunchKernelGGL(cart_rollout_fwd_kernel<APG_LAYOUT_AOS>,
                       dim3(grid_for(B, 256)), dim3(256), 0, st, state0, actions,
                       c, B, H, states_out);
  return check_launch("cartpole_rollout_fwd");
}

}  // extern "C"
