// quad.hip - quadrotor kernels: single step (+VJP), fused horizon-unrolled
// rollout + quad_mpc_loss + analytic adjoint, loss, policy-input features.
//
// Arithmetic restated from (paths relative to the reference repo):
//   neural_control/dynamics/quad_dynamics_flightmare.py:128-216
//   neural_control/dynamics/quad_dynamics_base.py:59-127
//   neural_control/drone_loss.py:12-39
//   neural_control/dataset.py:207-220
// Closed form of one step (state = [p, att=(phi,theta,psi), v, w]):
//   T    = 15 a0 - 7.5 + 9.81
//   z    = third row of world_to_body(att)
//   acc  = T z + gravity + translational_drag          (mass cancels)
//   p'   = p + 0.5 dt^2 acc + 0.5 dt v                  (sic)
//   v'   = v + dt acc
//   w'   = w + dt (K (a_{1:3} - 0.5 - w) + J^-1 rotational_drag)
//          (the w x Jw term is added and subtracted in the reference)
//   att' = att + dt E(phi,theta) w                      (old w)
// All of it is per-trajectory elementwise work: HBM-bound, no MFMA.
#include "apg_device.h"
#include "quad_math.h"

namespace apg {
namespace {

// ------------------------------------------------------------ single step --
template <int LAYOUT>
__global__ __launch_bounds__(256) void quad_step_fwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    QuadConst c, int B, float *__restrict__ next) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], a[4];
  load_state<LAYOUT, 12>(state, B, b, s);
  load_state<LAYOUT, 4>(action, B, b, a);
  Trig t = make_trig(&s[3]);
  quad_step(s, a, c, t);
  store_state<LAYOUT, 12>(next, B, b, s);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void quad_step_bwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    QuadConst c, int B, const float *__restrict__ grad_next,
    float *__restrict__ grad_state, float *__restrict__ grad_action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], a[4], lam[12];
  load_state<LAYOUT, 12>(state, B, b, s);
  load_state<LAYOUT, 4>(action, B, b, a);
  load_state<LAYOUT, 12>(grad_next, B, b, lam);
  Trig t = make_trig(&s[3]);
  float ga[4] = {0.f, 0.f, 0.f, 0.f};
  quad_step_adjoint(lam, ga, a[0], &s[9], c, t);
  if (grad_state) store_state<LAYOUT, 12>(grad_state, B, b, lam);
  if (grad_action) store_state<LAYOUT, 4>(grad_action, B, b, ga);
}

// ------------------------------------------------------------ fused rollout
struct RolloutArgs {
  const float *state0, *actions, *ref;
  float *loss_partials, *grad_actions, *grad_state0, *states_out;
  QuadConst c;
  ApgQuadLossWeights w;
  ApgDeferredLoss prev;  // prev_partials == NULL: nothing deferred
  int B, H, ref_cols, vel_col;
};

// Compile-time horizon: the whole trajectory lives in registers and the HBM
// stream is software-pipelined against the arithmetic (one wave per SIMD at
// B = 65 536, so nothing else would hide the latency):
//   prologue      : issue state0, all actions and the LAST two reference rows
//                   (64 loads in flight - the vmcnt ceiling).
//   forward sweep : step k needs only state0/actions; after each step one more
//                   reference row is requested, in REVERSE step order.  Stashes
//                   sin/cos(att_k), w_k and (p, v)_{k+1}.
//   reverse sweep : consumes the reference rows in the order they were
//                   requested (k = H-1 .. 0): loss terms, seeds, adjoint of
//                   step k, store dL/da_k.  The tail of the load stream thus
//                   overlaps the adjoint arithmetic instead of preceding it.
// HBM traffic per trajectory: 48 + 16H + 24H read, 16H (+48) written.
template <int LAYOUT, int HT, bool STATES_OUT, bool BUF>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) void quad_rollout_reg_kernel(
    RolloutArgs A) {
  static_assert(!BUF || LAYOUT == APG_LAYOUT_SOA, "buffer path is SoA only");
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;  // keep the wave convergent for the reduce
  const QuadConst c = A.c;
  // Just-in-time request schedule: only state0 and the first kActPre action
  // rows are requested up front (every wave's FIRST data is then near the
  // head of the memory queues and lands ~0.6 us earlier than behind a 64-load
  // burst per wave); each forward step requests one more action row and one
  // reference row (reverse order), so everything is in flight by the end of
  // the forward sweep.
  constexpr int kActPre = HT < 3 ? HT : 3;

  // deferred loss of an earlier launch (ApgDeferredLoss): request its
  // partials before this wave's own inputs, sum them at the very end
  const bool reducer = blockIdx.x == 0 && threadIdx.x < kWave &&
                       A.prev.prev_partials != nullptr;
  PrevPartials pp;
  if (reducer) reduce_prev_head(A.prev, pp);
  __builtin_amdgcn_sched_barrier(0);

  // accessors: buffer-addressed planes (SoA fast path) or flat addresses
  const SoaPlanes b_s0(A.state0, 12, A.B, bb), b_act(A.actions, HT * 4, A.B, bb),
      b_ref(A.ref, HT * A.ref_cols, A.B, bb),
      b_ga(A.grad_actions, HT * 4, A.B, bb),
      b_gs(A.grad_state0, 12, A.B, bb), b_so(A.states_out, HT * 12, A.B, bb);
  auto ld_ref = [&](int k, float(&p)[3], float(&v)[3]) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        p[i] = b_ref.ld(k * A.ref_cols + i);
        v[i] = b_ref.ld(k * A.ref_cols + A.vel_col + i);
      }
    } else {
      load_seq<LAYOUT, 3>(A.ref, A.B, HT, A.ref_cols, bb, k, 0, p);
      load_seq<LAYOUT, 3>(A.ref, A.B, HT, A.ref_cols, bb, k, A.vel_col, v);
    }
  };

  float s[12];
  float act[HT][4];
  float rp[HT][3], rv[HT][3];
  // (sched_barriers pin the request order: the memory system returns loads
  // in order, so program order here IS the arrival order)
  auto ld_act = [&](int k) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) act[k][i] = b_act.ld(k * 4 + i);
    } else {
      load_seq<LAYOUT, 4>(A.actions, A.B, HT, 4, bb, k, 0, act[k]);
    }
  };
  if constexpr (BUF) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = b_s0.ld(i);
  } else {
    load_state<LAYOUT, 12>(A.state0, A.B, bb, s);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < kActPre; ++k) {
    ld_act(k);
    __builtin_amdgcn_sched_barrier(0);
  }

  Trig st_trig[HT];
  float st_w[HT + 1][3];
  float st_pv[HT][6];
#pragma unroll
  for (int k = 0; k < HT; ++k) {
    {  // requests of this step: action row k + kActPre, reference row H-1-k
      const int kr = HT - 1 - k;
      if (k + kActPre < HT) ld_act(k + kActPre);
      ld_ref(kr, rp[kr], rv[kr]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) st_w[k][i] = s[9 + i];
    st_trig[k] = make_trig(&s[3]);
    quad_step(s, act[k], c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) st_pv[k][i] = s[i], st_pv[k][3 + i] = s[6 + i];
    if constexpr (STATES_OUT) {
      if (live) {
        if constexpr (BUF) {
#pragma unroll
          for (int i = 0; i < 12; ++i) b_so.st(k * 12 + i, s[i]);
        } else {
          store_seq<LAYOUT, 12>(A.states_out, A.B, HT, 12, b, k, 0, s);
        }
      }
    }

  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[HT][i] = s[9 + i];

  float loss = 0.f;
  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int k = HT - 1; k >= 0; --k) {
    // loss terms of step k (drone_loss.py:22-34) and their seeds
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rp[k][i];
      const float dv = st_pv[k][3 + i] - rv[k][i];
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
    if (live) {
      if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b_ga.st(k * 4 + i, ga[i]);
      } else {
        store_seq<LAYOUT, 4>(A.grad_actions, A.B, HT, 4, b, k, 0, ga);
      }
    }
  }
  if (A.grad_state0 && live) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 12; ++i) b_gs.st(i, lam[i]);
    } else {
      store_state<LAYOUT, 12>(A.grad_state0, A.B, b, lam);
    }
  }
  write_wave_partial(A.loss_partials, live ? loss : 0.f);
  if (reducer) reduce_prev_tail(A.prev, pp);
}

// Reference (AoS, row-major) tensors, compile-time horizon.  A wave owns 64
// consecutive trajectories, i.e. ONE contiguous slab of each tensor
// (state0 3 KB, actions 16H*64 B, ref 4*RC*H*64 B).  The slabs are pulled
// into LDS with direct-to-LDS buffer loads (`buffer_load_dwordx4 ... lds`,
// 1 KB per wave instruction, fully coalesced, no VGPR staging); every lane
// then reads ITS row from LDS (row strides 48 / 160 / 360 B: at most 2-way
// bank conflicts).  dL/dactions rows are written back into the action slab
// and leave through the same coalesced 16-byte-per-lane pattern.  Compared
// with per-lane row loads straight from HBM this turns 64 partial cache
// lines per wave instruction into one full KB, and frees the ~100 VGPRs the
// SoA kernel spends on in-flight inputs.
template <int HT, int RC, bool STATES_OUT>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) void quad_rollout_aos_kernel(
    RolloutArgs A) {
  constexpr int kS0 = 64 * 12, kAct = 64 * 4 * HT;
  constexpr int kRefChunks = (64 * RC * HT * 4 + 1023) / 1024;
  constexpr int kVel = RC == 9 ? 6 : 3;
  __shared__ __attribute__((aligned(16))) float l_s0[kS0];
  __shared__ __attribute__((aligned(16))) float l_act[kAct];
  __shared__ __attribute__((aligned(16))) float l_ref[kRefChunks * 256];
  typedef __attribute__((address_space(3))) void *lds_ptr;
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < A.B;
  const QuadConst c = A.c;

  const bool reducer = blockIdx.x == 0 && threadIdx.x < kWave &&
                       A.prev.prev_partials != nullptr;
  PrevPartials pp;
  if (reducer) reduce_prev_head(A.prev, pp);

  // slabs -> LDS (out-of-range rows of the last workgroup read as zero)
  const auto r_s0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A.state0), 0, A.B * 12 * 4, 0x00020000);
  const auto r_act = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A.actions), 0, A.B * HT * 4 * 4, 0x00020000);
  const auto r_ref = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(A.ref), 0, A.B * HT * RC * 4, 0x00020000);
  const auto r_ga = __builtin_amdgcn_make_buffer_rsrc(
      A.grad_actions, 0, A.B * HT * 4 * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r_s0, (lds_ptr)(l_s0 + i * 256), 16, lane * 16,
        blockIdx.x * (kS0 * 4) + i * 1024, 0, 0);
#pragma unroll
  for (int i = 0; i < HT; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r_act, (lds_ptr)(l_act + i * 256), 16, lane * 16,
        blockIdx.x * (kAct * 4) + i * 1024, 0, 0);
#pragma unroll
  for (int i = 0; i < kRefChunks; ++i)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r_ref, (lds_ptr)(l_ref + i * 256), 16, lane * 16,
        blockIdx.x * (64 * RC * HT * 4) + i * 1024, 0, 0);
  // state0 + actions have landed once only the reference chunks are pending
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRefChunks) : "memory");

  float s[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float4 v = *reinterpret_cast<const float4 *>(&l_s0[lane * 12 + 4 * i]);
    s[4 * i] = v.x, s[4 * i + 1] = v.y, s[4 * i + 2] = v.z, s[4 * i + 3] = v.w;
  }
  Trig st_trig[HT];
  float st_w[HT + 1][3];
  float st_pv[HT][6];
#pragma unroll
  for (int k = 0; k < HT; ++k) {
    const float4 a4 =
        *reinterpret_cast<const float4 *>(&l_act[lane * 4 * HT + 4 * k]);
    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int i = 0; i < 3; ++i) st_w[k][i] = s[9 + i];
    st_trig[k] = make_trig(&s[3]);
    quad_step(s, a, c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) st_pv[k][i] = s[i], st_pv[k][3 + i] = s[6 + i];
    if constexpr (STATES_OUT)
      if (live)
        store_seq<APG_LAYOUT_AOS, 12>(A.states_out, A.B, HT, 12, b, k, 0, s);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[HT][i] = s[9 + i];

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // reference slab landed

  float loss = 0.f;
  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int k = HT - 1; k >= 0; --k) {
    const float *rrow = &l_ref[lane * RC * HT + k * RC];
    float4 a4 = *reinterpret_cast<const float4 *>(&l_act[lane * 4 * HT + 4 * k]);
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rrow[i];
      const float dv = st_pv[k][3 + i] - rrow[kVel + i];
      const float wn = st_w[k + 1][i];
      lp += dp * dp, lv += dv * dv, lw += wn * wn;
      lam[i] += 2.f * A.w.pos * dp;
      lam[6 + i] += 2.f * A.w.vel * dv;
      lam[9 + i] += 2.f * A.w.av * wn;
    }
    const float a0 = act[0], da0 = a0 - 0.5f;
    float ga[4];
    ga[0] = 2.f * A.w.thrust * da0;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = act[i] - 0.5f;
      lr += d * d;
      ga[i] = 2.f * A.w.rates * d;
    }
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
    quad_step_adjoint(lam, ga, a0, st_w[k], c, st_trig[k]);
    *reinterpret_cast<float4 *>(&l_act[lane * 4 * HT + 4 * k]) =
        make_float4(ga[0], ga[1], ga[2], ga[3]);
  }
  // dL/dactions slab: LDS -> HBM, 1 KB per wave instruction, streaming
  // stores; rows past B fall outside the buffer and are dropped
#pragma unroll
  for (int i = 0; i < HT; ++i) {
    const float4 v = *reinterpret_cast<const float4 *>(&l_act[i * 256 + lane * 4]);
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(
        __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned,
                           (v4f){v.x, v.y, v.z, v.w}),
        r_ga, lane * 16, blockIdx.x * (kAct * 4) + i * 1024, 2);
  }
  if (A.grad_state0 && live)
    store_state<APG_LAYOUT_AOS, 12>(A.grad_state0, A.B, b, lam);
  write_wave_partial(A.loss_partials, live ? loss : 0.f);
  if (reducer) reduce_prev_tail(A.prev, pp);
}

// Run-time horizon: same sweeps, the per-step stash (att, w, seeds: 12 floats)
// is staged in LDS as [k][12][lane] (conflict-free: lane == bank).
template <int LAYOUT, bool STATES_OUT>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) void quad_rollout_lds_kernel(
    RolloutArgs A) {
  extern __shared__ float stash[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;
  const QuadConst c = A.c;
  const int H = A.H;
  auto ST = [&](int k, int i) -> float & {
    return stash[(k * 12 + i) * APG_ROLLOUT_BLOCK + lane];
  };

  float s[12];
  load_state<LAYOUT, 12>(A.state0, A.B, bb, s);
  if (blockIdx.x == 0 && threadIdx.x < kWave && A.prev.prev_partials)
    reduce_prev_partials(A.prev);
  float loss = 0.f;
  for (int k = 0; k < H; ++k) {
    float a[4], rp[3], rv[3];
    load_seq<LAYOUT, 4>(A.actions, A.B, H, 4, bb, k, 0, a);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, 0, rp);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, A.vel_col, rv);
#pragma unroll
    for (int i = 0; i < 3; ++i) ST(k, i) = s[3 + i], ST(k, 3 + i) = s[9 + i];
    Trig t = make_trig(&s[3]);
    quad_step(s, a, c, t);
    if constexpr (STATES_OUT)
      if (live) store_seq<LAYOUT, 12>(A.states_out, A.B, H, 12, b, k, 0, s);
    float lp = 0.f, lv = 0.f, lw = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float dp = s[i] - rp[i], dv = s[6 + i] - rv[i];
      lp += dp * dp, lv += dv * dv, lw += s[9 + i] * s[9 + i];
      ST(k, 6 + i) = 2.f * A.w.pos * dp;
      ST(k, 9 + i) = 2.f * A.w.vel * dv;
    }
    float lr = 0.f;
#pragma unroll
    for (int i = 1; i < 4; ++i) lr += (a[i] - 0.5f) * (a[i] - 0.5f);
    const float da0 = a[0] - 0.5f;
    loss += A.w.pos * lp + A.w.vel * lv + A.w.av * lw + A.w.rates * lr +
            A.w.thrust * da0 * da0;
  }
  write_wave_partial(A.loss_partials, live ? loss : 0.f);

  float lam[12], wn[3] = {s[9], s[10], s[11]};
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  for (int k = H - 1; k >= 0; --k) {
    float a[4], att[3], w[3];
    load_seq<LAYOUT, 4>(A.actions, A.B, H, 4, bb, k, 0, a);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      att[i] = ST(k, i), w[i] = ST(k, 3 + i);
      lam[i] += ST(k, 6 + i);
      lam[6 + i] += ST(k, 9 + i);
      lam[9 + i] += 2.f * A.w.av * wn[i];
    }
    float ga[4] = {2.f * A.w.thrust * (a[0] - 0.5f),
                   2.f * A.w.rates * (a[1] - 0.5f),
                   2.f * A.w.rates * (a[2] - 0.5f),
                   2.f * A.w.rates * (a[3] - 0.5f)};
    Trig t = make_trig(att);
    quad_step_adjoint(lam, ga, a[0], w, c, t);
    if (live) store_seq<LAYOUT, 4>(A.grad_actions, A.B, H, 4, b, k, 0, ga);
#pragma unroll
    for (int i = 0; i < 3; ++i) wn[i] = w[i];
  }
  if (A.grad_state0 && live) store_state<LAYOUT, 12>(A.grad_state0, A.B, b, lam);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void quad_rollout_fwd_kernel(
    const float *__restrict__ state0, const float *__restrict__ actions,
    QuadConst c, int B, int H, float *__restrict__ states_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12];
  load_state<LAYOUT, 12>(state0, B, b, s);
  for (int k = 0; k < H; ++k) {
    float a[4];
    load_seq<LAYOUT, 4>(actions, B, H, 4, b, k, 0, a);
    Trig t = make_trig(&s[3]);
    quad_step(s, a, c, t);
    store_seq<LAYOUT, 12>(states_out, B, H, 12, b, k, 0, s);
  }
}

// ------------------------------------------------------------- loss alone --
template <int LAYOUT>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) void quad_loss_kernel(
    const float *__restrict__ states, const float *__restrict__ ref,
    int ref_cols, int vel_col, const float *__restrict__ actions,
    ApgQuadLossWeights w, int B, int H, float *__restrict__ partials,
    float *__restrict__ grad_states, float *__restrict__ grad_actions) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  float loss = 0.f;
  for (int k = 0; k < H; ++k) {
    float s[12], a[4], rp[3], rv[3], gs[12], ga[4];
    load_seq<LAYOUT, 12>(states, B, H, 12, bb, k, 0, s);
    load_seq<LAYOUT, 4>(actions, B, H, 4, bb, k, 0, a);
    load_seq<LAYOUT, 3>(ref, B, H, ref_cols, bb, k, 0, rp);
    load_seq<LAYOUT, 3>(ref, B, H, ref_cols, bb, k, vel_col, rv);
    float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float dp = s[i] - rp[i], dv = s[6 + i] - rv[i], d = a[1 + i] - 0.5f;
      lp += dp * dp, lv += dv * dv, lw += s[9 + i] * s[9 + i], lr += d * d;
      gs[i] = 2.f * w.pos * dp, gs[3 + i] = 0.f;
      gs[6 + i] = 2.f * w.vel * dv, gs[9 + i] = 2.f * w.av * s[9 + i];
      ga[1 + i] = 2.f * w.rates * d;
    }
    const float da0 = a[0] - 0.5f;
    ga[0] = 2.f * w.thrust * da0;
    loss += w.pos * lp + w.vel * lv + w.av * lw + w.rates * lr +
            w.thrust * da0 * da0;
    if (live && grad_states)
      store_seq<LAYOUT, 12>(grad_states, B, H, 12, b, k, 0, gs);
    if (live && grad_actions)
      store_seq<LAYOUT, 4>(grad_actions, B, H, 4, b, k, 0, ga);
  }
  write_wave_partial(partials, live ? loss : 0.f);
}

// ------------------------------------------------------ policy-input features
template <int LAYOUT>
__global__ __launch_bounds__(256) void quad_features_fwd_kernel(
    const float *__restrict__ state, int B, float *__restrict__ feat) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], f[15];
  load_state<LAYOUT, 12>(state, B, b, s);
  quad_features(s, make_trig(&s[3]), f);
  store_state<LAYOUT, 15>(feat, B, b, f);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void quad_features_bwd_kernel(
    const float *__restrict__ state, const float *__restrict__ gfeat, int B,
    float *__restrict__ gstate) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], gf[15], gs[12];
  load_state<LAYOUT, 12>(state, B, b, s);
  load_state<LAYOUT, 15>(gfeat, B, b, gf);
  quad_features_adjoint(s, make_trig(&s[3]), gf, gs);
  store_state<LAYOUT, 12>(gstate, B, b, gs);
}

// ----------------------------------------------------------------- host ----
inline int grid_for(int B, int block) { return (B + block - 1) / block; }

int check_common(const void *p0, const void *p1, const void *params, int B,
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

template <int LAYOUT, bool SO>
int launch_rollout(const RolloutArgs &A, hipStream_t st) {
  const dim3 grid(grid_for(A.B, APG_ROLLOUT_BLOCK)), block(APG_ROLLOUT_BLOCK);
  // buffer addressing needs every tensor below 2 GiB (32-bit byte offsets)
  const bool buf_ok = (long long)A.H * 12 * A.B * 4 < (1ll << 31);
  if constexpr (LAYOUT == APG_LAYOUT_AOS) {
    // slab path: every tensor below 2 GiB and a register-resident horizon
    if (buf_ok && (A.H == 5 || A.H == 10)) {
#define APG_AOS(HT, RC)                                                       \
  hipLaunchKernelGGL((quad_rollout_aos_kernel<HT, RC, SO>), grid, block, 0,  \
                     st, A)
      if (A.H == 10 && A.ref_cols == 9) APG_AOS(10, 9);
      else if (A.H == 10) APG_AOS(10, 6);
      else if (A.ref_cols == 9) APG_AOS(5, 9);
      else APG_AOS(5, 6);
#undef APG_AOS
      return check_launch("quad_rollout_fwd_bwd");
    }
  }
  switch (A.H) {
#define APG_CASE(HT)                                                          \
  case HT:                                                                    \
    if (LAYOUT == APG_LAYOUT_SOA && buf_ok)                                   \
      hipLaunchKernelGGL(                                                     \
          (quad_rollout_reg_kernel<LAYOUT, HT, SO, LAYOUT == APG_LAYOUT_SOA>), \
          grid, block, 0, st, A);                                             \
    else                                                                      \
      hipLaunchKernelGGL((quad_rollout_reg_kernel<LAYOUT, HT, SO, false>),    \
                         grid, block, 0, st, A);                              \
    break;
    APG_CASE(5) APG_CASE(10)  // register-resident horizons (reference configs)
#undef APG_CASE
    default: {
      const size_t lds = (size_t)A.H * 12 * APG_ROLLOUT_BLOCK * sizeof(float);
      if (lds > 64 * 1024 &&
          hipFuncSetAttribute((const void *)quad_rollout_lds_kernel<LAYOUT, SO>,
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return check_launch("hipFuncSetAttribute(quad_rollout_lds)");
      hipLaunchKernelGGL((quad_rollout_lds_kernel<LAYOUT, SO>), grid, block, lds,
                         st, A);
    }
  }
  return check_launch("quad_rollout_fwd_bwd");
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

int apg_quad_step_fwd(const float *state, const float *action, float dt,
                      const ApgQuadParams *params, int B, int layout,
                      float *next_state, apg_stream_t stream) {
  if (int e = check_common(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!next_state) { set_error("next_state is NULL"); return APG_ERR_ARG; }
  QuadConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_step_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, next_state);
  else
    hipLaunchKernelGGL(quad_step_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, next_state);
  return check_launch("quad_step_fwd");
}

int apg_quad_step_bwd(const float *state, const float *action, float dt,
                      const ApgQuadParams *params, int B, int layout,
                      const float *grad_next, float *grad_state,
                      float *grad_action, apg_stream_t stream) {
  if (int e = check_common(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!grad_next) { set_error("grad_next is NULL"); return APG_ERR_ARG; }
  QuadConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_step_bwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, grad_next,
                       grad_state, grad_action);
  else
    hipLaunchKernelGGL(quad_step_bwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, c, B, grad_next,
                       grad_state, grad_action);
  return check_launch("quad_step_bwd");
}

int apg_quad_rollout_fwd_bwd(const float *state0, const float *actions,
                             const float *ref, int ref_cols, float dt,
                             const ApgQuadParams *params,
                             const ApgQuadLossWeights *weights, int B, int H,
                             int layout, float *loss_partials, float *loss,
                             float *grad_actions, float *grad_state0,
                             float *states_out,
                             const ApgDeferredLoss *deferred,
                             apg_stream_t stream) {
  if (int e = check_common(state0, actions, params, B, layout)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  if (deferred && deferred->prev_partials) {
    if (!deferred->prev_loss || deferred->prev_count < 0 ||
        deferred->prev_partials == loss_partials) {
      set_error("deferred: prev_loss NULL, prev_count < 0 or prev_partials "
                "aliases loss_partials");
      return APG_ERR_ARG;
    }
  }
  if (H < 1 || H > APG_MAX_HORIZON) {
    set_error("H must be in [1, %d] (got %d)", APG_MAX_HORIZON, H);
    return APG_ERR_ARG;
  }
  if (ref_cols != 9 && ref_cols != 6) {
    set_error("ref_cols must be 9 ([pos, euler, vel]) or 6 ([pos, vel])");
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool has_prev = deferred && deferred->prev_partials;
  if (B == 0) {  // empty batch: loss = 0, nothing else to write
    if (loss) {
      if (hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
        return check_launch("memset(loss)");
    }
    if (has_prev)
      return launch_reduce_partials(deferred->prev_partials,
                                    deferred->prev_count, deferred->prev_loss, st);
    return APG_OK;
  }
  if (!ref || !loss_partials || !grad_actions) {
    set_error("ref / loss_partials / grad_actions must not be NULL");
    return APG_ERR_ARG;
  }
  RolloutArgs A;
  A.state0 = state0, A.actions = actions, A.ref = ref;
  A.loss_partials = loss_partials, A.grad_actions = grad_actions;
  A.grad_state0 = grad_state0, A.states_out = states_out;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.prev = has_prev ? *deferred : ApgDeferredLoss{nullptr, 0, nullptr};
  A.B = B, A.H = H, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  int e;
  if (layout == APG_LAYOUT_SOA)
    e = states_out ? launch_rollout<APG_LAYOUT_SOA, true>(A, st)
                   : launch_rollout<APG_LAYOUT_SOA, false>(A, st);
  else
    e = states_out ? launch_rollout<APG_LAYOUT_AOS, true>(A, st)
                   : launch_rollout<APG_LAYOUT_AOS, false>(A, st);
  if (e) return e;
  if (loss) return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
  return APG_OK;
}

int apg_quad_rollout_fwd(const float *state0, const float *actions, float dt,
                         const ApgQuadParams *params, int B, int H, int layout,
                         float *states_out, apg_stream_t stream) {
  if (int e = check_common(state0, actions, params, B, layout)) return e;
  if (H < 1) { set_error("H must be >= 1 (got %d)", H); return APG_ERR_ARG; }
  if (B == 0) return APG_OK;
  if (!states_out) { set_error("states_out is NULL"); return APG_ERR_ARG; }
  QuadConst c = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_rollout_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state0, actions, c, B, H, states_out);
  else
    hipLaunchKernelGGL(quad_rollout_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state0, actions, c, B, H, states_out);
  return check_launch("quad_rollout_fwd");
}

int apg_quad_loss_fwd_bwd(const float *states, const float *ref, int ref_cols,
                          const float *actions,
                          const ApgQuadLossWeights *weights, int B, int H,
                          int layout, float *loss_partials, float *loss,
                          float *grad_states, float *grad_actions,
                          apg_stream_t stream) {
  if (int e = check_common(states, actions, weights, B, layout)) return e;
  if (H < 1) { set_error("H must be >= 1 (got %d)", H); return APG_ERR_ARG; }
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
  if (!ref || !loss_partials) {
    set_error("ref / loss_partials must not be NULL");
    return APG_ERR_ARG;
  }
  const int vel_col = ref_cols == 9 ? 6 : 3;
  const dim3 grid(grid_for(B, APG_ROLLOUT_BLOCK)), block(APG_ROLLOUT_BLOCK);
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_loss_kernel<APG_LAYOUT_SOA>, grid, block, 0, st, states,
                       ref, ref_cols, vel_col, actions, *weights, B, H,
                       loss_partials, grad_states, grad_actions);
  else
    hipLaunchKernelGGL(quad_loss_kernel<APG_LAYOUT_AOS>, grid, block, 0, st, states,
                       ref, ref_cols, vel_col, actions, *weights, B, H,
                       loss_partials, grad_states, grad_actions);
  if (int e = check_launch("quad_loss_fwd_bwd")) return e;
  if (loss) return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
  return APG_OK;
}

int apg_quad_features_fwd(const float *state, int B, int layout,
                          float *features, apg_stream_t stream) {
  if (int e = check_common(state, features, "", B, layout)) return e;
  if (B == 0) return APG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_features_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, B, features);
  else
    hipLaunchKernelGGL(quad_features_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, B, features);
  return check_launch("quad_features_fwd");
}

int apg_quad_features_bwd(const float *state, const float *grad_features,
                          int B, int layout, float *grad_state,
                          apg_stream_t stream) {
  if (int e = check_common(state, grad_features, "", B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!grad_state) { set_error("grad_state is NULL"); return APG_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(quad_features_bwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, grad_features, B, grad_state);
  else
    hipLaunchKernelGGL(quad_features_bwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, grad_features, B, grad_state);
  return check_launch("quad_features_bwd");
}

}  // extern "C"
