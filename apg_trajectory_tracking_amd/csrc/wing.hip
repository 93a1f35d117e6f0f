// wing.hip - fixed-wing kernels: single step (+VJP), fused rollout +
// fixed_wing_mpc_loss + analytic adjoint.
//
// Arithmetic restated from (paths relative to the reference repo):
//   neural_control/dynamics/fixed_wing_dynamics.py:41-267
//   neural_control/dynamics/config_fixed_wing.json
//   neural_control/drone_loss.py:72-82
//   scripts/train_fixed_wing.py:90-110
// State = [pos NED (x,y,z), body velocity (u,v,w), euler (phi,theta,psi),
// body rates (p,q,r)]; explicit Euler: next = state + dt * state_dot.
// Quirks kept on purpose: all three moments scale with the chord c
// (:170-175), gravity is rotated with psi = 0 (:195-197), alpha and beta are
// clamped to +-10 deg so their gradient vanishes outside (:132-134).
#include <stddef.h>
#include <stdlib.h>

#include "apg_device.h"
#include "wing_math.h"

namespace apg {
namespace {


template <int LAYOUT>
__global__ __launch_bounds__(256) void wing_step_fwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    WingConst k, int B, float *__restrict__ next) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], a[4];
  load_state<LAYOUT, 12>(state, B, b, s);
  load_state<LAYOUT, 4>(action, B, b, a);
  wing_step(s, a, k);
  store_state<LAYOUT, 12>(next, B, b, s);
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void wing_step_bwd_kernel(
    const float *__restrict__ state, const float *__restrict__ action,
    WingConst k, int B, const float *__restrict__ grad_next,
    float *__restrict__ grad_state, float *__restrict__ grad_action) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12], a[4], lam[12], sd[12];
  load_state<LAYOUT, 12>(state, B, b, s);
  load_state<LAYOUT, 4>(action, B, b, a);
  load_state<LAYOUT, 12>(grad_next, B, b, lam);
  WingAux x;
  wing_rates(s, a, k, x, sd);
  float ga[4] = {0.f, 0.f, 0.f, 0.f};
  wing_step_adjoint(lam, ga, s, x, sd, k);
  if (grad_state) store_state<LAYOUT, 12>(grad_state, B, b, lam);
  if (grad_action) store_state<LAYOUT, 4>(grad_action, B, b, ga);
}

struct WingRolloutArgs {
  const float *state0, *actions, *ref;
  float *loss_partials, *grad_actions, *grad_state0, *states_out;
  WingConst k;
  ApgWingLossWeights w;
  ApgDeferredLoss prev;
  int B, H, stride;
};

// Run-time horizon, checkpointed.  The reverse sweep needs the pre-step state
// of every step; stashing all of them ([H][12][lane] in LDS, 61 KB per wave
// at H = 20) leaves room for only two waves per CU, i.e. half the SIMDs idle
// on VALU-bound work.  Instead only every `stride`-th pre-step state is kept
// ([ceil(H/stride)][12][lane]); the reverse sweep walks the horizon group by
// group, re-integrating the (stride-1) missing states of a group from its
// checkpoint into registers.  Costs (stride-1)/stride extra state_dot
// evaluations, buys 3-5x the resident waves.
constexpr int kWingMaxStride = 4;
constexpr float kWingDefaultPosWeight = 10.f, kWingDefaultActionWeight = 0.1f;
// Experiments (tools/exp): waves per SIMD the register allocator must leave
// room for, and whether the reverse sweep requests a group's rows one group
// ahead (28 more live registers).
#ifdef APG_WING_WAVES
#define APG_WING_OCCUPANCY __attribute__((amdgpu_waves_per_eu(APG_WING_WAVES, APG_WING_WAVES)))
#else
#define APG_WING_OCCUPANCY   /* 227 registers: two waves per SIMD */
#endif
#ifndef APG_WING_GROUP_PREFETCH
#define APG_WING_GROUP_PREFETCH 1
#endif

// LIT: the default parameter set as instruction literals (WingDefaultK) -
// selected by the host when make_const(params, dt) equals that table.
// BUF (plane layout, every tensor below 2 GiB): rows are addressed through
// buffer resources - per-lane offset in one VGPR, plane offset in an SGPR -
// instead of 64-bit per-lane address arithmetic on the VALU; dead lanes store
// through an out-of-range offset instead of an exec-mask branch.
template <int LAYOUT, bool LIT, bool BUF>
__global__ __launch_bounds__(APG_ROLLOUT_BLOCK) APG_WING_OCCUPANCY void wing_rollout_lds_kernel(
    WingRolloutArgs A) {
  static_assert(!BUF || LAYOUT == APG_LAYOUT_SOA, "buffer path is SoA only");
  extern __shared__ float stash[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;
  const SoaPlanes b_act(A.actions, A.H * 4, A.B, bb), b_ref(A.ref, A.H * 3, A.B, bb);
  SoaPlanes b_ga(A.grad_actions, A.H * 4, A.B, bb),
      b_so(A.states_out, A.H * 12, A.B, bb), b_gs(A.grad_state0, 12, A.B, bb);
  if (!live) b_ga.voff = b_so.voff = b_gs.voff = (int)0x80000000;
  auto ld_act = [&](int kq, float(&o)[4]) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = b_act.ld(kq * 4 + i);
    } else {
      load_seq<LAYOUT, 4>(A.actions, A.B, A.H, 4, bb, kq, 0, o);
    }
  };
  auto ld_ref = [&](int kq, float(&o)[3]) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 3; ++i) o[i] = b_ref.ld(kq * 3 + i);
    } else {
      load_seq<LAYOUT, 3>(A.ref, A.B, A.H, 3, bb, kq, 0, o);
    }
  };
  // The ~70 coefficients do not fit the SGPR file next to everything else:
  // kept live for the whole kernel they are spilled to VGPR lanes
  // (v_writelane / v_readlane, ~10 % of the issued instructions).  Reading
  // them from the kernel-argument segment where they are used (scalar loads,
  // scalar-cache hits) is cheaper; `launder` stops the compiler from hoisting
  // those loads back out of the step loops.
  typedef __attribute__((address_space(4))) const WingConst *const_ptr;
  typedef __attribute__((address_space(4))) const char *const_bytes;
  const_ptr kp = (const_ptr)((const_bytes)__builtin_amdgcn_kernarg_segment_ptr() +
                             offsetof(WingRolloutArgs, k));
#define APG_LAUNDER(p) asm volatile("" : "+s"(p))
  const WingDefaultK kl;
  // (the literal instance is selected for the default loss weights only:
  // fixed_wing_mpc_loss, neural_control/drone_loss.py:72-82)
  const float w_pos = LIT ? kWingDefaultPosWeight : A.w.pos,
              w_act = LIT ? kWingDefaultActionWeight : A.w.action;
  // one step / one evaluation on whichever constant table this instance uses
  auto step = [&](float(&s_)[12], const float(&a_)[4]) {
    if constexpr (LIT) {
      wing_step(s_, a_, kl);
    } else {
      APG_LAUNDER(kp);
      wing_step(s_, a_, *kp);
    }
  };
  const int H = A.H, S = A.stride;
  auto ST = [&](int slot, int i) -> float & {
    return stash[(slot * 12 + i) * APG_ROLLOUT_BLOCK + lane];
  };
  float s[12];
  load_state<LAYOUT, 12>(A.state0, A.B, bb, s);
  if (blockIdx.x == 0 && threadIdx.x < kWave && A.prev.prev_partials)
    reduce_prev_partials(A.prev);
  // The loads are software-pipelined against the arithmetic: with two waves
  // per SIMD a step's own loads would expose most of the ~1 us HBM latency
  // twice per step.  Forward sweep: the rows of step kk + 2 are requested
  // while step kk computes.
  float loss = 0.f;
  float a_q[2][4], r_q[2][3];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int kq = d < H ? d : H - 1;
    ld_act(kq, a_q[d]);
    ld_ref(kq, r_q[d]);
  }
  for (int kk = 0, slot = 0, phase = 0; kk < H; ++kk) {
    float a[4], rp[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a_q[0][i], a_q[0][i] = a_q[1][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) rp[i] = r_q[0][i], r_q[0][i] = r_q[1][i];
    {
      const int kq = kk + 2 < H ? kk + 2 : H - 1;
      ld_act(kq, a_q[1]);
      ld_ref(kq, r_q[1]);
    }
    if (phase == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) ST(slot, i) = s[i];
      ++slot;
    }
    if (++phase == S) phase = 0;
    step(s, a);
    if (A.states_out) {
      if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < 12; ++i) b_so.st(kk * 12 + i, s[i]);
      } else if (live) {
        store_seq<LAYOUT, 12>(A.states_out, A.B, H, 12, b, kk, 0, s);
      }
    }
    float lp = 0.f, la = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float dp = s[i] - rp[i], d = a[1 + i] - 0.5f;
      lp += dp * dp, la += d * d;
    }
    loss += w_pos * lp + w_act * la;
  }
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);

  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
  float nxt[3] = {s[0], s[1], s[2]};  // position after the group's last step
  const int G = (H + S - 1) / S;
  // reverse sweep: the action and reference rows of group g - 1 are requested
  // while group g is re-integrated and differentiated
  float act[kWingMaxStride][4], rpg[kWingMaxStride][3];
  float act_n[kWingMaxStride][4], rpg_n[kWingMaxStride][3];
  auto load_group = [&](int g, float(&ga_)[kWingMaxStride][4],
                        float(&gr_)[kWingMaxStride][3]) {
#pragma unroll
    for (int j = 0; j < kWingMaxStride; ++j) {
      int kq = g * S + j;           // rows past the group / horizon: any valid row
      kq = (j < S && kq < H) ? kq : H - 1;
      ld_act(kq, ga_[j]);
      ld_ref(kq, gr_[j]);
    }
  };
#if APG_WING_GROUP_PREFETCH
  load_group(G - 1, act, rpg);
#endif
  for (int g = G - 1; g >= 0; --g) {
    const int k0 = g * S;
    const int n = (H - k0) < S ? (H - k0) : S;
#if APG_WING_GROUP_PREFETCH
    load_group(g > 0 ? g - 1 : 0, act_n, rpg_n);
#else
    load_group(g, act, rpg);
#endif
    float pre[kWingMaxStride + 1][12];
#pragma unroll
    for (int i = 0; i < 12; ++i) pre[0][i] = ST(g, i);
    // re-integrate the states inside the group
#pragma unroll
    for (int j = 0; j < kWingMaxStride; ++j) {
      if (j < n) {
        if (j + 1 < kWingMaxStride && j + 1 < n) {
#pragma unroll
          for (int i = 0; i < 12; ++i) pre[j + 1][i] = pre[j][i];
          step(pre[j + 1], act[j]);
        }
      }
    }
    // adjoint of the group's steps, last first
#pragma unroll
    for (int j = kWingMaxStride - 1; j >= 0; --j) {
      if (j < n) {
        const int kk = k0 + j;
        float pn[3] = {nxt[0], nxt[1], nxt[2]}, sd[12];
        if (j + 1 < kWingMaxStride && j + 1 < n) {
#pragma unroll
          for (int i = 0; i < 3; ++i) pn[i] = pre[j + 1][i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) lam[i] += 2.f * w_pos * (pn[i] - rpg[j][i]);
        float ga[4] = {0.f, 2.f * w_act * (act[j][1] - 0.5f),
                       2.f * w_act * (act[j][2] - 0.5f),
                       2.f * w_act * (act[j][3] - 0.5f)};
        WingAux x;
        if constexpr (LIT) {
          wing_rates(pre[j], act[j], kl, x, sd);
          wing_step_adjoint(lam, ga, pre[j], x, sd, kl);
        } else {
          APG_LAUNDER(kp);
          wing_rates(pre[j], act[j], *kp, x, sd);
          wing_step_adjoint(lam, ga, pre[j], x, sd, *kp);
        }
        if constexpr (BUF) {
#pragma unroll
          for (int i = 0; i < 4; ++i) b_ga.st(kk * 4 + i, ga[i]);
        } else if (live) {
          store_seq<LAYOUT, 4>(A.grad_actions, A.B, H, 4, b, kk, 0, ga);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) nxt[i] = pre[0][i];
#if APG_WING_GROUP_PREFETCH
#pragma unroll
    for (int j = 0; j < kWingMaxStride; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) act[j][i] = act_n[j][i];
#pragma unroll
      for (int i = 0; i < 3; ++i) rpg[j][i] = rpg_n[j][i];
    }
#endif
  }
  if (A.grad_state0) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 12; ++i) b_gs.st(i, lam[i]);
    } else if (live) {
      store_state<LAYOUT, 12>(A.grad_state0, A.B, b, lam);
    }
  }
#undef APG_LAUNDER
}

// ------------------------------------------------ two trajectories per lane --
// The same rollout with T = fx2 (wing_math.h): lane l of wave w integrates
// trajectories 128 w + 2 l and + 1.  Plane layout ([C][B], B even): the two
// floats of a lane are adjacent, so every load / store is one 8-byte buffer
// access per lane (512 B per wave instruction, fully coalesced).  B = 131 072
// is then 1 024 waves = ONE wave per SIMD: packed fp32 issues at the single-
// wave rate, transcendentals cost 8 cycles instead of the ~23 they cost each
// of two co-resident waves (profiles/r03_issue_probe2.jsonl), and up to 512
// registers per lane hold what the two trajectories keep live.
// KMODE: where the ~70 coefficients come from.  0: the kernel-argument table,
// read with scalar loads where used (as the one-per-lane kernel does);
// 1: instruction literals of the default parameter set (a packed op cannot
// encode a literal, so each becomes an s_mov into an SGPR operand);
// 2: VGPR-resident for the whole kernel (no scalar traffic, but with two
// trajectories per lane the 256 architectural VGPRs are full: the table ends
// up in AGPRs and every use costs a v_accvgpr_read - measured, not shipped).
#if !defined(APG_EXPERIMENT_BUILD) && defined(APG_WING_CLOCK)
#error "experiment macro in a product build (variants: tools/build_wing_variant.sh)"
#endif
#ifdef APG_WING_CLOCK
// Shader clock under the kernel: every wave stamps s_memtime
// (shader cycles) and s_memrealtime (the constant 100 MHz reference clock) at
// its first and last instruction; tools/wing_clock.py reads them back.
__device__ unsigned long long apg_wing_clock[4 * 4096];
#endif
template <int KMODE>
__global__ __launch_bounds__(64) void wing_rollout_pk_kernel(WingRolloutArgs A) {
  extern __shared__ fx2 stash2[];
  typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
#ifdef APG_WING_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(),
                           clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x;
  const int pr = blockIdx.x * 64 + lane;     // pair of trajectories
  const int npairs = A.B >> 1;
  const bool live = pr < npairs;
  const int vld = (live ? pr : npairs - 1) * 8;
  const int vst = live ? pr * 8 : (int)0x80000000;   // dead lanes: out of range
  const int pitch = A.B * 4;
  const auto rs = [&](const void *p, int planes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0,
                                             planes * A.B * 4, 0x00020000);
  };
  const auto r_s0 = rs(A.state0, 12), r_act = rs(A.actions, A.H * 4),
             r_ref = rs(A.ref, A.H * 3), r_ga = rs(A.grad_actions, A.H * 4),
             r_so = rs(A.states_out, A.H * 12), r_gs = rs(A.grad_state0, 12);
  auto ld2 = [&](__amdgpu_buffer_rsrc_t r, int plane) {
    return __builtin_bit_cast(
        fx2, __builtin_amdgcn_raw_buffer_load_b64(r, vld, plane * pitch, 0));
  };
  auto st2 = [&](__amdgpu_buffer_rsrc_t r, int plane, fx2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v_, v), r, vst,
                                          plane * pitch, 2);
  };
  auto ld_act = [&](int kq, fx2(&o)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ld2(r_act, kq * 4 + i);
  };
  auto ld_ref = [&](int kq, fx2(&o)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = ld2(r_ref, kq * 3 + i);
  };
  typedef __attribute__((address_space(4))) const WingConst *const_ptr;
  typedef __attribute__((address_space(4))) const char *const_bytes;
  const_ptr kp = (const_ptr)((const_bytes)__builtin_amdgcn_kernarg_segment_ptr() +
                             offsetof(WingRolloutArgs, k));
#define APG_LAUNDER(p) asm volatile("" : "+s"(p))
  const WingDefaultK kl;
  WingConst kv = A.k;
  float w_pos = KMODE == 1 ? kWingDefaultPosWeight : A.w.pos,
        w_act = KMODE == 1 ? kWingDefaultActionWeight : A.w.action;
  if constexpr (KMODE == 2) {
    float *f = reinterpret_cast<float *>(&kv);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(WingConst) / sizeof(float)); ++i)
      asm volatile("" : "+v"(f[i]));
    asm volatile("" : "+v"(w_pos), "+v"(w_act));
  }
  auto step = [&](fx2(&s_)[12], const fx2(&a_)[4]) {
    if constexpr (KMODE == 1) {
      wing_step(s_, a_, kl);
    } else if constexpr (KMODE == 2) {
      wing_step(s_, a_, kv);
    } else {
      APG_LAUNDER(kp);
      wing_step(s_, a_, *kp);
    }
  };
  const int H = A.H, S = A.stride;
  auto ST = [&](int slot, int i) -> fx2 & { return stash2[(slot * 12 + i) * 64 + lane]; };
  fx2 s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = ld2(r_s0, i);
  if (blockIdx.x == 0 && A.prev.prev_partials) reduce_prev_partials(A.prev);
  fx2 loss = {0.f, 0.f};
  fx2 a_q[2][4], r_q[2][3];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int kq = d < H ? d : H - 1;
    ld_act(kq, a_q[d]);
    ld_ref(kq, r_q[d]);
  }
  for (int kk = 0, slot = 0, phase = 0; kk < H; ++kk) {
    fx2 a[4], rp[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = a_q[0][i], a_q[0][i] = a_q[1][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) rp[i] = r_q[0][i], r_q[0][i] = r_q[1][i];
    {
      const int kq = kk + 2 < H ? kk + 2 : H - 1;
      ld_act(kq, a_q[1]);
      ld_ref(kq, r_q[1]);
    }
    if (phase == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) ST(slot, i) = s[i];
      ++slot;
    }
    if (++phase == S) phase = 0;
    step(s, a);
    if (A.states_out) {
#pragma unroll
      for (int i = 0; i < 12; ++i) st2(r_so, kk * 12 + i, s[i]);
    }
    fx2 lp = {0.f, 0.f}, la = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const fx2 dp = s[i] - rp[i], d = a[1 + i] - 0.5f;
      lp += dp * dp, la += d * d;
    }
    loss += w_pos * lp + w_act * la;
  }
  {  // the host reduces ceil(B / 64) partials (one per 64 trajectories): this
     // wave's 128 trajectories fill two of them (even / odd trajectories), or
     // one when the second lies beyond the batch's partial count
    const int count = (A.B + kWave - 1) / kWave;
    const float s0 = wave_sum(live ? loss.x : 0.f), s1 = wave_sum(live ? loss.y : 0.f);
    if (lane == 0) {
      const int j = 2 * blockIdx.x;
      if (j + 1 < count) A.loss_partials[j] = s0, A.loss_partials[j + 1] = s1;
      else if (j < count) A.loss_partials[j] = s0 + s1;
    }
  }

  fx2 lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = (fx2){0.f, 0.f};
  fx2 nxt[3] = {s[0], s[1], s[2]};  // position after the group's last step
  const int G = (H + S - 1) / S;
  fx2 act[kWingMaxStride][4], rpg[kWingMaxStride][3];
  auto load_group = [&](int g, fx2(&ga_)[kWingMaxStride][4],
                        fx2(&gr_)[kWingMaxStride][3]) {
#pragma unroll
    for (int j = 0; j < kWingMaxStride; ++j) {
      int kq = g * S + j;           // rows past the group / horizon: any valid row
      kq = (j < S && kq < H) ? kq : H - 1;
      ld_act(kq, ga_[j]);
      ld_ref(kq, gr_[j]);
    }
  };
  for (int g = G - 1; g >= 0; --g) {
    const int k0 = g * S;
    const int n = (H - k0) < S ? (H - k0) : S;
    load_group(g, act, rpg);
    fx2 pre[kWingMaxStride + 1][12];
#pragma unroll
    for (int i = 0; i < 12; ++i) pre[0][i] = ST(g, i);
#pragma unroll
    for (int j = 0; j < kWingMaxStride; ++j) {
      if (j < n) {
        if (j + 1 < kWingMaxStride && j + 1 < n) {
#pragma unroll
          for (int i = 0; i < 12; ++i) pre[j + 1][i] = pre[j][i];
          step(pre[j + 1], act[j]);
        }
      }
    }
#pragma unroll
    for (int j = kWingMaxStride - 1; j >= 0; --j) {
      if (j < n) {
        const int kk = k0 + j;
        fx2 pn[3] = {nxt[0], nxt[1], nxt[2]}, sd[12];
        if (j + 1 < kWingMaxStride && j + 1 < n) {
#pragma unroll
          for (int i = 0; i < 3; ++i) pn[i] = pre[j + 1][i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) lam[i] += 2.f * w_pos * (pn[i] - rpg[j][i]);
        fx2 ga[4] = {(fx2){0.f, 0.f}, 2.f * w_act * (act[j][1] - 0.5f),
                     2.f * w_act * (act[j][2] - 0.5f),
                     2.f * w_act * (act[j][3] - 0.5f)};
        WingAuxT<fx2> x;
        if constexpr (KMODE == 1) {
          wing_rates(pre[j], act[j], kl, x, sd);
          wing_step_adjoint(lam, ga, pre[j], x, sd, kl);
        } else if constexpr (KMODE == 2) {
          wing_rates(pre[j], act[j], kv, x, sd);
          wing_step_adjoint(lam, ga, pre[j], x, sd, kv);
        } else {
          APG_LAUNDER(kp);
          wing_rates(pre[j], act[j], *kp, x, sd);
          wing_step_adjoint(lam, ga, pre[j], x, sd, *kp);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) st2(r_ga, kk * 4 + i, ga[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) nxt[i] = pre[0][i];
  }
  if (A.grad_state0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) st2(r_gs, i, lam[i]);
  }
#ifdef APG_WING_CLOCK
  if (lane == 0 && blockIdx.x < 4096) {
    unsigned long long *c = apg_wing_clock + 4 * blockIdx.x;
    c[0] = clk_t0, c[1] = __builtin_amdgcn_s_memtime();
    c[2] = clk_r0, c[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#undef APG_LAUNDER
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void wing_rollout_fwd_kernel(
    const float *__restrict__ state0, const float *__restrict__ actions,
    WingConst k, int B, int H, float *__restrict__ states_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[12];
  load_state<LAYOUT, 12>(state0, B, b, s);
  for (int kk = 0; kk < H; ++kk) {
    float a[4];
    load_seq<LAYOUT, 4>(actions, B, H, 4, b, kk, 0, a);
    wing_step(s, a, k);
    store_seq<LAYOUT, 12>(states_out, B, H, 12, b, kk, 0, s);
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

// two trajectories per lane: 0 never, 1 whenever possible, 2 by batch size
// (shipped; experiment builds may start from another value)
#ifndef APG_WING_PK
#define APG_WING_PK 2
#endif
static int g_wing_pk_mode = APG_WING_PK;

extern "C" {

#ifdef APG_WING_CLOCK
int apg_wing_clock_read(unsigned long long *host, int n) {   // experiment builds only
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(apg_wing_clock),
                             (size_t)n * sizeof(unsigned long long)) == hipSuccess
             ? APG_OK : APG_ERR_HIP;
}
#endif

int apg_wing_set_two_per_lane(int mode) {
  if (mode < 0 || mode > 2) {
    set_error("apg_wing_set_two_per_lane: mode must be 0, 1 or 2 (got %d)", mode);
    return APG_ERR_ARG;
  }
  g_wing_pk_mode = mode;
  return APG_OK;
}

int apg_wing_step_fwd(const float *state, const float *action, float dt,
                      const ApgWingParams *params, int B, int layout,
                      float *next_state, apg_stream_t stream) {
  if (int e = check_args(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!next_state) { set_error("next_state is NULL"); return APG_ERR_ARG; }
  WingConst k = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(wing_step_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, k, B, next_state);
  else
    hipLaunchKernelGGL(wing_step_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, k, B, next_state);
  return check_launch("wing_step_fwd");
}

int apg_wing_step_bwd(const float *state, const float *action, float dt,
                      const ApgWingParams *params, int B, int layout,
                      const float *grad_next, float *grad_state,
                      float *grad_action, apg_stream_t stream) {
  if (int e = check_args(state, action, params, B, layout)) return e;
  if (B == 0) return APG_OK;
  if (!grad_next) { set_error("grad_next is NULL"); return APG_ERR_ARG; }
  WingConst k = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(wing_step_bwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, k, B, grad_next,
                       grad_state, grad_action);
  else
    hipLaunchKernelGGL(wing_step_bwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state, action, k, B, grad_next,
                       grad_state, grad_action);
  return check_launch("wing_step_bwd");
}

int apg_wing_rollout_fwd_bwd(const float *state0, const float *actions,
                             const float *ref, float dt,
                             const ApgWingParams *params,
                             const ApgWingLossWeights *weights, int B, int H,
                             int layout, float *loss_partials, float *loss,
                             float *grad_actions, float *grad_state0,
                             float *states_out,
                             const ApgDeferredLoss *deferred,
                             apg_stream_t stream) {
  if (int e = check_args(state0, actions, params, B, layout)) return e;
  if (!weights) { set_error("weights is NULL"); return APG_ERR_ARG; }
  const bool has_prev = deferred && deferred->prev_partials;
  if (has_prev && (!deferred->prev_loss || deferred->prev_count < 0 ||
                   deferred->prev_partials == loss_partials)) {
    set_error("deferred: prev_loss NULL, prev_count < 0 or prev_partials "
              "aliases loss_partials");
    return APG_ERR_ARG;
  }
  if (H < 1 || H > APG_MAX_HORIZON) {
    set_error("H must be in [1, %d] (got %d)", APG_MAX_HORIZON, H);
    return APG_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    if (has_prev)
      return launch_reduce_partials(deferred->prev_partials,
                                    deferred->prev_count, deferred->prev_loss, st);
    return APG_OK;
  }
  if (!ref || !loss_partials || !grad_actions) {
    set_error("ref / loss_partials / grad_actions must not be NULL");
    return APG_ERR_ARG;
  }
  WingRolloutArgs A;
  A.state0 = state0, A.actions = actions, A.ref = ref;
  A.loss_partials = loss_partials, A.grad_actions = grad_actions;
  A.grad_state0 = grad_state0, A.states_out = states_out;
  A.k = make_const(*params, dt);
  A.w = *weights;
  A.prev = has_prev ? *deferred : ApgDeferredLoss{nullptr, 0, nullptr};
  A.B = B, A.H = H;
  // checkpoint stride: keep the stash of a wave <= 6 slots (18 KB) so that
  // >= 8 waves fit a CU's 160 KB of LDS.  (Stride sweeps: variant builds
  // only, tools/build_wing_variant.sh - the product reads no environment.)
  int stride = (H + 5) / 6;
#ifdef APG_EXPERIMENT_BUILD
  if (const char *e = getenv("APG_WING_STRIDE")) stride = atoi(e);
#endif
  if (stride < 1) stride = 1;
  if (stride > kWingMaxStride) stride = kWingMaxStride;
  A.stride = stride;
  const size_t lds = (size_t)((H + stride - 1) / stride) * 12 *
                     APG_ROLLOUT_BLOCK * sizeof(float);
  const dim3 grid(grid_for(B, APG_ROLLOUT_BLOCK)), block(APG_ROLLOUT_BLOCK);
  // default parameter set and dt: the instance with the coefficients as
  // instruction literals; anything else (modified_params, another dt): the
  // table in the kernel-argument segment
#ifndef APG_WING_LITERALS
#define APG_WING_LITERALS 1
#endif
  const bool lit = APG_WING_LITERALS && is_default(A.k) &&
                   A.w.pos == kWingDefaultPosWeight &&
                   A.w.action == kWingDefaultActionWeight;
  // buffer addressing needs every tensor below 2 GiB (32-bit byte offsets)
  const bool buf_ok = (long long)H * 12 * B * 4 < (1ll << 31);
  // Two trajectories per lane (packed fp32, one wave per SIMD) as soon as the
  // one-per-lane kernel would put more than one wave on a SIMD; needs the
  // plane layout, an even batch and 8-byte aligned tensors.
  const int simds = 4 * device_cu_count();
  const auto al8 = [](const void *q) { return ((size_t)q & 7) == 0; };
  const bool pk_ok = layout == APG_LAYOUT_SOA && buf_ok && (B & 1) == 0 &&
                     al8(state0) && al8(actions) && al8(ref) && al8(grad_actions) &&
                     al8(grad_state0) && al8(states_out);
  // nothing in the environment changes the shipped choice; tests pick a
  // kernel through apg_wing_set_two_per_lane (1 = whenever possible, 0 = never)
  const int pk_mode = g_wing_pk_mode;
  if (pk_ok && (pk_mode == 1 || (pk_mode == 2 && grid_for(B, 64) > simds))) {
    const size_t lds2 = 2 * lds;
    const dim3 grid2(grid_for(B / 2, 64)), block2(64);
#define APG_WING_PK_LAUNCH(KM)                                                \
  do {                                                                        \
    if (lds2 > 64 * 1024 &&                                                   \
        hipFuncSetAttribute((const void *)wing_rollout_pk_kernel<KM>,         \
                            hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)lds2) != hipSuccess)                         \
      return check_launch("hipFuncSetAttribute(wing_rollout_pk)");            \
    hipLaunchKernelGGL((wing_rollout_pk_kernel<KM>), grid2, block2, lds2, st, A); \
  } while (0)
#ifdef APG_WING_PK_KMODE     /* experiment builds: force one constant source */
    if (APG_WING_PK_KMODE == 1 && !lit) APG_WING_PK_LAUNCH(0);
    else APG_WING_PK_LAUNCH(APG_WING_PK_KMODE);
#else
    if (lit) APG_WING_PK_LAUNCH(1);
    else APG_WING_PK_LAUNCH(0);
#endif
#undef APG_WING_PK_LAUNCH
    if (int e = check_launch("wing_rollout_fwd_bwd(two per lane)")) return e;
    if (loss)
      return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
    return APG_OK;
  }
#define APG_WING_LAUNCH(L, LIT, BUF)                                          \
  do {                                                                        \
    if (lds > 64 * 1024 &&                                                    \
        hipFuncSetAttribute(                                                  \
            (const void *)wing_rollout_lds_kernel<L, LIT, BUF>,               \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
      return check_launch("hipFuncSetAttribute(wing_rollout)");               \
    hipLaunchKernelGGL((wing_rollout_lds_kernel<L, LIT, BUF>), grid, block,   \
                       lds, st, A);                                           \
  } while (0)
  if (layout == APG_LAYOUT_SOA && buf_ok) {
    if (lit) APG_WING_LAUNCH(APG_LAYOUT_SOA, true, true);
    else APG_WING_LAUNCH(APG_LAYOUT_SOA, false, true);
  } else if (layout == APG_LAYOUT_SOA) {
    if (lit) APG_WING_LAUNCH(APG_LAYOUT_SOA, true, false);
    else APG_WING_LAUNCH(APG_LAYOUT_SOA, false, false);
  } else {
    if (lit) APG_WING_LAUNCH(APG_LAYOUT_AOS, true, false);
    else APG_WING_LAUNCH(APG_LAYOUT_AOS, false, false);
  }
#undef APG_WING_LAUNCH
  if (int e = check_launch("wing_rollout_fwd_bwd")) return e;
  if (loss)
    return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
  return APG_OK;
}

int apg_wing_rollout_fwd(const float *state0, const float *actions, float dt,
                         const ApgWingParams *params, int B, int H, int layout,
                         float *states_out, apg_stream_t stream) {
  if (int e = check_args(state0, actions, params, B, layout)) return e;
  if (H < 1) { set_error("H must be >= 1 (got %d)", H); return APG_ERR_ARG; }
  if (B == 0) return APG_OK;
  if (!states_out) { set_error("states_out is NULL"); return APG_ERR_ARG; }
  WingConst k = make_const(*params, dt);
  hipStream_t st = (hipStream_t)stream;
  if (layout == APG_LAYOUT_SOA)
    hipLaunchKernelGGL(wing_rollout_fwd_kernel<APG_LAYOUT_SOA>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state0, actions, k, B, H, states_out);
  else
    hipLaunchKernelGGL(wing_rollout_fwd_kernel<APG_LAYOUT_AOS>, dim3(grid_for(B, 256)),
                       dim3(256), 0, st, state0, actions, k, B, H, states_out);
  return check_launch("wing_rollout_fwd");
}

}  // extern "C"
