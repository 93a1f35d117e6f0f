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

#ifndef APG_QX
#define APG_QX 0  // experiment bits (tools/exp builds): 1 loss accumulators,
                  // 4 attitude-first request order
#endif

namespace apg {
namespace {

#ifdef APG_STAMP
// timing experiments only (tools/ab_quad.cpp): per-wave s_memtime stamps.  They
// are parked in the lanes of ONE VGPR (v_writelane) and written out once at the
// end of the kernel, so that a stamp costs the s_memtime round trip and two
// VALU slots - no stores, no loads of the buffer pointer in between.
__device__ unsigned long long *g_stamps = nullptr;
#define APG_STAMP_DECL unsigned apg_stv = 0
#define APG_STAMP_PUT(i, t_)                                                   \
  asm volatile("v_writelane_b32 %0, %1, %3\nv_writelane_b32 %0, %2, %4"        \
               : "+v"(apg_stv)                                                 \
               : "s"((unsigned)(t_)), "s"((unsigned)((t_) >> 32)), "n"(2 * (i)), \
                 "n"(2 * (i) + 1))
#define APG_STAMP_AT(i)                                                        \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    unsigned long long t_;                                                     \
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   \
    APG_STAMP_PUT(i, t_);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
// same, but not before the scalar value `dep` (a kernel argument) has arrived
#define APG_STAMP_DEP(i, dep)                                                  \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    unsigned long long t_;                                                     \
    asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)"                          \
                 : "=s"(t_)                                                    \
                 : "s"(dep)                                                    \
                 : "memory");                                                  \
    APG_STAMP_PUT(i, t_);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
// s_memrealtime: the constant 100 MHz clock all XCDs share (s_memtime is per XCD)
#define APG_STAMP_REAL(i)                                                      \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    unsigned long long t_;                                                     \
    asm volatile("s_memrealtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    APG_STAMP_PUT(i, t_);                                                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define APG_STAMP_FLUSH()                                                      \
  do {                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
    APG_STAMP_AT(4);                                                           \
    APG_STAMP_REAL(5);                                                         \
    if (g_stamps && (threadIdx.x & 63) < 16)                                   \
      ((unsigned *)g_stamps)[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 16 + \
                             (threadIdx.x & 63)] = apg_stv;                    \
  } while (0)
#else
#define APG_STAMP_DECL
#define APG_STAMP_AT(i)
#define APG_STAMP_DEP(i, dep)
#define APG_STAMP_REAL(i)
#define APG_STAMP_FLUSH()
#endif

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
  APG_STAMP_DECL;
  APG_STAMP_AT(0);
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
#ifndef APG_REG_ACT_PRE
#define APG_REG_ACT_PRE 3
#endif
#ifndef APG_REG_REF_PER_STEP
#define APG_REG_REF_PER_STEP 1
#endif
  constexpr int kActPre = HT < (APG_REG_ACT_PRE) ? HT : (APG_REG_ACT_PRE);
  constexpr int kRefPerStep = APG_REG_REF_PER_STEP;
#ifndef APG_REG_REF_LOOK
#define APG_REG_REF_LOOK 3
#endif
  // kRefLook > 0 (measured faster, profiles/r02_ab_jit.json): the reference
  // rows are requested just in time - the last kRefLook forward steps ask for
  // rows H-1 .. H-kRefLook, reverse step k for row k - kRefLook - so that the
  // forward sweep is not paced by the reference stream (loads return in order)
  // and the reverse sweep reads while it computes
  constexpr int kRefLook = APG_REG_REF_LOOK > HT ? HT : APG_REG_REF_LOOK;

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

  APG_STAMP_REAL(7);
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
  if constexpr (BUF && (APG_QX & 4)) {
    // attitude and body rates first: the sin/cos of step 0 (the long pole of
    // a step) starts as soon as three planes have landed
#pragma unroll
    for (int i = 3; i < 6; ++i) s[i] = b_s0.ld(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 9; i < 12; ++i) s[i] = b_s0.ld(i);
    __builtin_amdgcn_sched_barrier(0);
    ld_act(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 6; i < 9; ++i) s[i] = b_s0.ld(i);
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = b_s0.ld(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 1; k < kActPre; ++k) {
      ld_act(k);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
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
  }

  Trig st_trig[HT];
  float st_w[HT + 1][3];
  float st_pv[HT][6];
#pragma unroll
  for (int k = 0; k < HT; ++k) {
    {  // requests of this step: action row k + kActPre, reference rows in
       // reverse step order
      if (k + kActPre < HT) ld_act(k + kActPre);
      if constexpr (kRefLook == 0) {
#pragma unroll
        for (int j = 0; j < kRefPerStep; ++j) {
          const int kr = HT - 1 - (k * kRefPerStep + j);
          if (kr >= 0) ld_ref(kr, rp[kr], rv[kr]);
        }
      } else if (k >= HT - kRefLook) {
        const int kr = HT - 1 - (k - (HT - kRefLook));
        ld_ref(kr, rp[kr], rv[kr]);
      }
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
    if (k == 0) APG_STAMP_AT(1);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[HT][i] = s[9 + i];
  APG_STAMP_AT(2);

  // loss terms (drone_loss.py:22-34): the five sums run over the whole
  // horizon and meet their weights once at the end
  float sum_p = 0.f, sum_v = 0.f, sum_w = 0.f, sum_r = 0.f, sum_t = 0.f;
  const float wp2 = 2.f * A.w.pos, wv2 = 2.f * A.w.vel, ww2 = 2.f * A.w.av,
              wr2 = 2.f * A.w.rates, wt2 = 2.f * A.w.thrust;
  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int k = HT - 1; k >= 0; --k) {
    if constexpr (kRefLook > 0) {
      if (k - kRefLook >= 0) ld_ref(k - kRefLook, rp[k - kRefLook], rv[k - kRefLook]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rp[k][i];
      const float dv = st_pv[k][3 + i] - rv[k][i];
      const float wn = st_w[k + 1][i];
      sum_p = fmaf(dp, dp, sum_p), sum_v = fmaf(dv, dv, sum_v);
      sum_w = fmaf(wn, wn, sum_w);
      lam[i] = fmaf(wp2, dp, lam[i]);
      lam[6 + i] = fmaf(wv2, dv, lam[6 + i]);
      lam[9 + i] = fmaf(ww2, wn, lam[9 + i]);
    }
    const float a0 = act[k][0], da0 = a0 - 0.5f;
    float ga[4];
    ga[0] = wt2 * da0;
    sum_t = fmaf(da0, da0, sum_t);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = act[k][i] - 0.5f;
      sum_r = fmaf(d, d, sum_r);
      ga[i] = wr2 * d;
    }
    quad_step_adjoint(lam, ga, a0, st_w[k], c, st_trig[k]);
    if (live) {
      if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) b_ga.st(k * 4 + i, ga[i]);
      } else {
        store_seq<LAYOUT, 4>(A.grad_actions, A.B, HT, 4, b, k, 0, ga);
      }
    }
    if (k == HT / 2) APG_STAMP_AT(6);
  }
  if (A.grad_state0 && live) {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < 12; ++i) b_gs.st(i, lam[i]);
    } else {
      store_state<LAYOUT, 12>(A.grad_state0, A.B, b, lam);
    }
  }
  APG_STAMP_AT(3);
  const float loss = A.w.pos * sum_p + A.w.vel * sum_v + A.w.av * sum_w +
                     A.w.rates * sum_r + A.w.thrust * sum_t;
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);
  if (reducer) reduce_prev_tail(A.prev, pp);
  APG_STAMP_FLUSH();
}

// ---------------------------------------------- fused rollout, packed rows --
// APG_LAYOUT_PACKED.  Same sweeps as quad_rollout_reg_kernel, but every tensor
// is a stack of ROWS - the consecutive floats one trajectory needs at one
// step - with the batch as the next-faster dimension:
//   state0 [3][B][4]   actions [H][B][4]   ref [H][B][6] = [pos, vel]
//   grad_actions [H][B][4]   grad_state0 [3][B][4]   states_out [H][3][B][4]
// so a lane moves its action row with ONE 16-byte access and a wave
// instruction covers 1 KiB of contiguous memory.  Why it matters (measured,
// tools/issue_probe.hip, one wave per SIMD, all four waves of a CU active):
// a dword-per-lane buffer load occupies the issuing wave for ~38 cycles, a
// dword store for ~33 - the 112 loads + 40 stores of the plane layout cost a
// wave ~5 600 cycles (2.4 us) of pure issue time on top of ~2 000 VALU slots
// of 4.3 cycles; the row layout needs 33 loads + 10 stores.
// The first 16 dwords of the argument list are scalars so that they can be
// PRELOADED into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16): the wave
// issues its first loads without waiting ~0.35 us for an s_load round trip.
struct RowArgs {
  float wd[3];  // dt * rot_drag / inertia
  ApgQuadLossWeights w;
  float *loss_partials, *grad_actions, *grad_state0, *states_out;
  ApgDeferredLoss prev;
};

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// The request / store schedule of quad_rollout_rows_kernel, FROZEN: every
// value below won its sweep on the 20-buffer-set protocol (DESIGN.md §3.1,
// profiles/r02_ab_jit.json, r02_ab_actpre.json, r03_ab_quad.json).  The product
// build has no knobs; a variant build (tools/build_variant.py,
// -DAPG_EXPERIMENT_BUILD) may override single values through APG_ROWS_*.
namespace rows_tune {
#ifdef APG_EXPERIMENT_BUILD
#define APG_ROWS_TUNE(frozen, knob) (knob)
#else
#define APG_ROWS_TUNE(frozen, knob) (frozen)
#endif
#ifndef APG_ROWS_BLOCK
#define APG_ROWS_BLOCK 64
#endif
#ifndef APG_ROWS_ACT_PRE
#define APG_ROWS_ACT_PRE 3
#endif
#ifndef APG_ROWS_REF_PER_STEP
#define APG_ROWS_REF_PER_STEP 1
#endif
#ifndef APG_ROWS_REF_LOOK
#define APG_ROWS_REF_LOOK 3
#endif
#ifndef APG_ROWS_ST_AUX
#define APG_ROWS_ST_AUX 2
#endif
#ifndef APG_ROWS_LD_AUX
#define APG_ROWS_LD_AUX 0
#endif
#ifndef APG_ROWS_STORE_AT_END
#define APG_ROWS_STORE_AT_END 1
#endif
#ifndef APG_ROWS_STORE_FLUSH_AT
#define APG_ROWS_STORE_FLUSH_AT (-1)
#endif
#ifndef APG_ROWS_REF_TOP
#define APG_ROWS_REF_TOP 0
#endif
constexpr int kBlock = APG_ROWS_TUNE(64, APG_ROWS_BLOCK);  // threads per workgroup: one wave
constexpr int kActPreMax = APG_ROWS_TUNE(3, APG_ROWS_ACT_PRE);  // action rows requested up front
constexpr int kRefPerStep = APG_ROWS_TUNE(1, APG_ROWS_REF_PER_STEP);  // (kRefLookMax == 0 only)
// just-in-time reference rows: the last kRefLook forward steps request rows
// H-1 .. H-kRefLook, reverse step k requests row k - kRefLook
constexpr int kRefLookMax = APG_ROWS_TUNE(3, APG_ROWS_REF_LOOK);
constexpr int kStAux = APG_ROWS_TUNE(2, APG_ROWS_ST_AUX);  // dL/dactions stores: nt
constexpr int kLdAux = APG_ROWS_TUNE(0, APG_ROWS_LD_AUX);  // input rows: default cache policy
// dL/dactions rows wait in registers (the action rows' own) and are written
// AFTER the reverse sweep: once the inputs come from HBM, writes in flight
// slow the reads (8.63 -> 8.39 us at 20 buffer sets); kStoreFlushAt >= 0: rows
// above it are written when that step is done, later rows at once
constexpr bool kStoreAtEnd = APG_ROWS_TUNE(1, APG_ROWS_STORE_AT_END) != 0;
constexpr int kStoreFlushAt = APG_ROWS_TUNE(-1, APG_ROWS_STORE_FLUSH_AT);
// true: rows below H - kRefLook all requested at the top of the reverse sweep
constexpr bool kRefTop = APG_ROWS_TUNE(0, APG_ROWS_REF_TOP) != 0;
#undef APG_ROWS_TUNE
}  // namespace rows_tune

template <int HT, bool STATES_OUT>
__global__ __launch_bounds__(rows_tune::kBlock) void quad_rollout_rows_kernel(
    const float *state0, const float *actions, const float *ref, int B, float dt,
    float half_dt, float half_dt2, float g0, float g1, float g2, float k0,
    float k1, float k2, RowArgs R) {
  APG_STAMP_DECL;
  APG_STAMP_AT(0);
  const int b = blockIdx.x * rows_tune::kBlock + threadIdx.x;
  const bool live = b < B;
  const int bb = live ? b : B - 1;  // keep the wave convergent for the reduce
  // Request schedule.  Loads return in order, so the order of the requests IS
  // the order of arrival; the memory system delivers ~11 B / cycle / CU, i.e.
  // one 1 KiB row per wave every ~370 cycles - more than a forward step
  // computes in.  The forward sweep needs state0 and the action rows only:
  // they are requested first, all of them; the reference rows follow in
  // REVERSE step order, kRefPerStep per forward step, so that the reverse
  // sweep consumes them while they are still streaming in.
  using namespace rows_tune;
  constexpr int kActPre = HT < kActPreMax ? HT : kActPreMax;
  constexpr int kRefLook = kRefLookMax > HT ? HT : kRefLookMax;
  const auto r_s0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(state0), 0, 3 * B * 16, 0x00020000);
  const auto r_act = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(actions), 0, HT * B * 16, 0x00020000);
  const auto r_ref = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(ref), 0, HT * B * 24, 0x00020000);
  const int v16 = bb * 16, v24 = bb * 24, p16 = B * 16, p24 = B * 24;
  APG_STAMP_REAL(7);

  u4v s4[3], a4[HT], rA[HT];
  u2v rB[HT];
  auto ld_ref = [&](int kr) {
    rA[kr] = __builtin_amdgcn_raw_buffer_load_b128(r_ref, v24, kr * p24, kLdAux);
    rB[kr] = __builtin_amdgcn_raw_buffer_load_b64(r_ref, v24 + 16, kr * p24, kLdAux);
  };
#pragma unroll
  for (int g = 0; g < 3; ++g)
    s4[g] = __builtin_amdgcn_raw_buffer_load_b128(r_s0, v16, g * p16, kLdAux);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < kActPre; ++k)
    a4[k] = __builtin_amdgcn_raw_buffer_load_b128(r_act, v16, k * p16, kLdAux);
  __builtin_amdgcn_sched_barrier(0);

  // everything below the first requests can wait for the rest of the arguments
  QuadConst c;
  c.dt = dt, c.half_dt = half_dt, c.half_dt2 = half_dt2;
  c.g[0] = g0, c.g[1] = g1, c.g[2] = g2;
  c.kdt[0] = k0, c.kdt[1] = k1, c.kdt[2] = k2;
  c.wd[0] = R.wd[0], c.wd[1] = R.wd[1], c.wd[2] = R.wd[2];
  const auto r_ga = __builtin_amdgcn_make_buffer_rsrc(R.grad_actions, 0,
                                                      HT * B * 16, 0x00020000);
  // dead lanes store out of range (dropped by the buffer range check): no
  // exec-mask branches around the stores
  const int st16 = live ? b * 16 : (int)0x80000000;

  const bool reducer = blockIdx.x == 0 && threadIdx.x < kWave &&
                       R.prev.prev_partials != nullptr;
  PrevPartials pp;
  if (reducer) reduce_prev_head(R.prev, pp);
  __builtin_amdgcn_sched_barrier(0);

  float s[12];
  Trig st_trig[HT];
  float st_w[HT + 1][3];
  float st_pv[HT][6];
#pragma unroll
  for (int k = 0; k < HT; ++k) {
    {  // requests of this step
      if (k + kActPre < HT)
        a4[k + kActPre] = __builtin_amdgcn_raw_buffer_load_b128(
            r_act, v16, (k + kActPre) * p16, kLdAux);
      if constexpr (kRefLook == 0) {
#pragma unroll
        for (int j = 0; j < kRefPerStep; ++j) {
          const int kr = HT - 1 - (k * kRefPerStep + j);
          if (kr >= 0) ld_ref(kr);
        }
      } else if (k >= HT - kRefLook) {
        ld_ref(HT - 1 - (k - (HT - kRefLook)));  // rows H-1 .. H-kRefLook
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (k == 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        s[4 * g] = u2f(s4[g].x), s[4 * g + 1] = u2f(s4[g].y);
        s[4 * g + 2] = u2f(s4[g].z), s[4 * g + 3] = u2f(s4[g].w);
      }
    }
    const float act[4] = {u2f(a4[k].x), u2f(a4[k].y), u2f(a4[k].z), u2f(a4[k].w)};
#pragma unroll
    for (int i = 0; i < 3; ++i) st_w[k][i] = s[9 + i];
    st_trig[k] = make_trig(&s[3]);
    quad_step(s, act, c, st_trig[k]);
#pragma unroll
    for (int i = 0; i < 3; ++i) st_pv[k][i] = s[i], st_pv[k][3 + i] = s[6 + i];
    if constexpr (STATES_OUT) {
      const auto r_so = __builtin_amdgcn_make_buffer_rsrc(
          R.states_out, 0, HT * 3 * B * 16, 0x00020000);
#pragma unroll
      for (int g = 0; g < 3; ++g)
        __builtin_amdgcn_raw_buffer_store_b128(
            (u4v){f2u(s[4 * g]), f2u(s[4 * g + 1]), f2u(s[4 * g + 2]),
                  f2u(s[4 * g + 3])},
            r_so, st16, (k * 3 + g) * p16, 2);
    }
    if (k == 0) APG_STAMP_AT(1);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) st_w[HT][i] = s[9 + i];
  APG_STAMP_AT(2);

  float sum_p = 0.f, sum_v = 0.f, sum_w = 0.f, sum_r = 0.f, sum_t = 0.f;
  const float wp2 = 2.f * R.w.pos, wv2 = 2.f * R.w.vel, ww2 = 2.f * R.w.av,
              wr2 = 2.f * R.w.rates, wt2 = 2.f * R.w.thrust;
  float lam[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f;
#pragma unroll
  for (int k = HT - 1; k >= 0; --k) {
    if constexpr (kRefLook > 0 && kRefTop) {
      if (k == HT - 1) {
#pragma unroll
        for (int kr = HT - 1 - kRefLook; kr >= 0; --kr) ld_ref(kr);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (kRefLook > 0) {
      if (k - kRefLook >= 0) ld_ref(k - kRefLook);
      __builtin_amdgcn_sched_barrier(0);
    }
    const float rp[3] = {u2f(rA[k].x), u2f(rA[k].y), u2f(rA[k].z)};
    const float rv[3] = {u2f(rA[k].w), u2f(rB[k].x), u2f(rB[k].y)};
    const float act[4] = {u2f(a4[k].x), u2f(a4[k].y), u2f(a4[k].z), u2f(a4[k].w)};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = st_pv[k][i] - rp[i];
      const float dv = st_pv[k][3 + i] - rv[i];
      const float wn = st_w[k + 1][i];
      sum_p = fmaf(dp, dp, sum_p), sum_v = fmaf(dv, dv, sum_v);
      sum_w = fmaf(wn, wn, sum_w);
      lam[i] = fmaf(wp2, dp, lam[i]);
      lam[6 + i] = fmaf(wv2, dv, lam[6 + i]);
      lam[9 + i] = fmaf(ww2, wn, lam[9 + i]);
    }
    const float a0 = act[0], da0 = a0 - 0.5f;
    float ga[4];
    ga[0] = wt2 * da0;
    sum_t = fmaf(da0, da0, sum_t);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = act[i] - 0.5f;
      sum_r = fmaf(d, d, sum_r);
      ga[i] = wr2 * d;
    }
    quad_step_adjoint(lam, ga, a0, st_w[k], c, st_trig[k]);
    if constexpr (kStoreAtEnd) {
      // rows k > kFlush are kept in registers and written when step kFlush is
      // done (kFlush = -1, shipped: all of them after the sweep), later rows at
      // once (profiles/r03_ab_quad.json `store_timing`)
      constexpr int kFlush = kStoreFlushAt;
      a4[k] = (u4v){f2u(ga[0]), f2u(ga[1]), f2u(ga[2]), f2u(ga[3])};
      if (k <= kFlush) {
        if (k == kFlush) {
#pragma unroll
          for (int j = HT - 1; j > kFlush; --j)
            __builtin_amdgcn_raw_buffer_store_b128(a4[j], r_ga, st16, j * p16, kStAux);
        }
        __builtin_amdgcn_raw_buffer_store_b128(a4[k], r_ga, st16, k * p16, kStAux);
      }
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(
          (u4v){f2u(ga[0]), f2u(ga[1]), f2u(ga[2]), f2u(ga[3])}, r_ga, st16, k * p16,
          kStAux);
    }
    if (k == HT / 2) APG_STAMP_AT(6);
  }
  if constexpr (kStoreAtEnd && kStoreFlushAt < 0) {
#pragma unroll
    for (int k = HT - 1; k >= 0; --k)
      __builtin_amdgcn_raw_buffer_store_b128(a4[k], r_ga, st16, k * p16, kStAux);
  }
  if (R.grad_state0) {
    const auto r_gs = __builtin_amdgcn_make_buffer_rsrc(R.grad_state0, 0,
                                                        3 * B * 16, 0x00020000);
#pragma unroll
    for (int g = 0; g < 3; ++g)
      __builtin_amdgcn_raw_buffer_store_b128(
          (u4v){f2u(lam[4 * g]), f2u(lam[4 * g + 1]), f2u(lam[4 * g + 2]),
                f2u(lam[4 * g + 3])},
          r_gs, st16, g * p16, 2);
  }
  APG_STAMP_AT(3);
  const float loss = R.w.pos * sum_p + R.w.vel * sum_v + R.w.av * sum_w +
                     R.w.rates * sum_r + R.w.thrust * sum_t;
  write_wave_partial(R.loss_partials, live ? loss : 0.f, (B + kWave - 1) / kWave);
  if (reducer) reduce_prev_tail(R.prev, pp);
  APG_STAMP_FLUSH();
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
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);
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
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);

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

// ------------------------------------------- rollout through LearntDynamics --
// Controller phase of TrainBase.run_dynamics (scripts/train_base.py:334-375):
// the simulator is neural_control/dynamics/quad_dynamics_trained.py:61-69,
//   a' = linear_at a;  s' = simulate_quadrotor(a', s) + W2 relu(W1 [s; a'] + b1) + b2
// (4x4 action transform, 16 -> 64 -> 12 residual network), unrolled H steps
// with quad_mpc_loss and differentiated back to the policy's actions and
// state0 - the simulator's own parameters are frozen in this phase, so no
// parameter gradient leaves the kernel.  One lane = one trajectory; the
// weights are wave-uniform and arrive as scalar operands (s_load from the
// module's own tensors); the pre-step states of the reverse sweep are stashed
// in LDS as [k][12][lane]; the hidden layer is recomputed in the reverse sweep.
// ~4 600 fma per env-step fwd + bwd against ~175 for the analytic simulator:
// VALU-issue bound (run-time horizon, loops over k are not unrolled).
struct LearntArgs {
  const float *state0, *actions, *ref;
  float *loss_partials, *grad_actions, *grad_state0, *states_out;
  ApgLearntResidual m;
  QuadConst c;
  ApgQuadLossWeights w;
  int B, H, ref_cols, vel_col;
};

// the module's tensors seen through the constant address space: wave-uniform
// indices then become scalar loads (s_load_dwordxN) and the weights scalar
// operands of the v_fmac - no VGPR, no vector memory instruction per weight
typedef __attribute__((address_space(4))) const float *cfloat_ptr;
struct LearntWeights {
  cfloat_ptr linear_at, w1, b1, w2, b2;
  __device__ explicit LearntWeights(const ApgLearntResidual &m)
      : linear_at((cfloat_ptr)m.linear_at), w1((cfloat_ptr)m.w1),
        b1((cfloat_ptr)m.b1), w2((cfloat_ptr)m.w2), b2((cfloat_ptr)m.b2) {}
};

// (APG_OPAQUE: without it the compiler hoists hundreds of weight loads, runs
// out of SGPRs and spills them through v_writelane / v_readlane - 16 000 extra
// instructions per step body; with it a row of weights is loaded where it is
// used)
// The asm is tied to a VALUE of the running computation too: an asm that only
// touches the pointer is free to float to the top of the block with its loads.
#define APG_OPAQUE(p, v) asm volatile("" : "+s"(p), "+v"(v))
// ... and to all sixteen accumulators of a row, where a row has sixteen
#define APG_OPAQUE16(p, v, o)                                                   \
  asm volatile(""                                                              \
               : "+s"(p), "+v"(v[o + 0]), "+v"(v[o + 1]), "+v"(v[o + 2]),      \
                 "+v"(v[o + 3]), "+v"(v[o + 4]), "+v"(v[o + 5]), "+v"(v[o + 6]), \
                 "+v"(v[o + 7]), "+v"(v[o + 8]), "+v"(v[o + 9]), "+v"(v[o + 10]), \
                 "+v"(v[o + 11]), "+v"(v[o + 12]), "+v"(v[o + 13]),            \
                 "+v"(v[o + 14]), "+v"(v[o + 15]))
__device__ __forceinline__ void learnt_hidden(const LearntWeights &m,
                                              const float (&x)[16], float (&z)[64]) {
  cfloat_ptr w1 = m.w1, b1 = m.b1;
  float x0 = x[0];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    // row j's loads may start once row j-2 is accumulated: one row of scalar
    // loads in flight behind the row being multiplied
    if (j < 2) APG_OPAQUE(w1, x0);
    else APG_OPAQUE(w1, z[j - 2]);
    if (j % 16 == 0) APG_OPAQUE(b1, x0);   // (biases too: loop-invariant loads
    float acc = b1[j];                     //  get hoisted and spilled)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc = fmaf(w1[j * 16 + i], i == 0 ? x0 : x[i], acc);
    z[j] = acc;
  }
}

__device__ __forceinline__ void learnt_transform(const LearntWeights &m,
                                                 const float (&a)[4], float (&ap)[4]) {
  cfloat_ptr L = m.linear_at;
  float a0 = a[0];
  asm volatile("" : "+s"(L), "+v"(a0));
#pragma unroll
  for (int r = 0; r < 4; ++r)
    ap[r] = L[r * 4] * a0 + L[r * 4 + 1] * a[1] + L[r * 4 + 2] * a[2] +
            L[r * 4 + 3] * a[3];
}

// s <- LearntDynamics.forward(s, a)
__device__ __forceinline__ void learnt_step(float (&s)[12], const float (&a)[4],
                                            const LearntWeights &m,
                                            const QuadConst &c) {
  float ap[4], x[16], z[64], r[12];
  learnt_transform(m, a, ap);
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = s[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[12 + i] = ap[i];
  learnt_hidden(m, x, z);
#pragma unroll
  for (int j = 0; j < 64; ++j) z[j] = fmaxf(z[j], 0.f);
  cfloat_ptr w2 = m.w2, b2 = m.b2;
  APG_OPAQUE(b2, z[0]);
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    float acc[2] = {b2[q], 0.f};   // alternating chunks: one chunk of loads ahead
#pragma unroll
    for (int jb = 0; jb < 64; jb += 16) {
      float &ac = acc[(jb >> 4) & 1];
      APG_OPAQUE(w2, ac);
#pragma unroll
      for (int j = jb; j < jb + 16; ++j) ac = fmaf(w2[q * 64 + j], z[j], ac);
    }
    r[q] = acc[0] + acc[1];
  }
  const Trig t = make_trig(&s[3]);
  quad_step(s, ap, c, t);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] += r[i];
}

// lam: dL/d(next state) -> dL/d(state); ga += dL/d(action); `pre` = the state
// the step started from
__device__ __forceinline__ void learnt_step_adjoint(float (&lam)[12], float (&ga)[4],
                                                    const float (&pre)[12],
                                                    const float (&a)[4],
                                                    const LearntWeights &m,
                                                    const QuadConst &c) {
  float ap[4], x[16], z[64], gx[16];
  learnt_transform(m, a, ap);
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = pre[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) x[12 + i] = ap[i];
  learnt_hidden(m, x, z);
#pragma unroll
  for (int i = 0; i < 16; ++i) gx[i] = 0.f;
  // through W2 (row by row: contiguous scalar loads), the relu and W1
  cfloat_ptr w1 = m.w1, w2 = m.w2;
  float gh[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) gh[j] = 0.f;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
#pragma unroll
    for (int jb = 0; jb < 64; jb += 16) {
      APG_OPAQUE16(w2, gh, (jb + 32) % 64);   // after the chunk before the previous one
#pragma unroll
      for (int j = jb; j < jb + 16; ++j) gh[j] = fmaf(w2[q * 64 + j], lam[q], gh[j]);
    }
  }
  float gx2[32];   // even / odd rows accumulate apart: one row of loads ahead
#pragma unroll
  for (int i = 0; i < 32; ++i) gx2[i] = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const int o = (j & 1) * 16;
    if (j & 1) APG_OPAQUE16(w1, gx2, 16);   // after ALL fmas of row j - 2
    else APG_OPAQUE16(w1, gx2, 0);
    const float g = z[j] > 0.f ? gh[j] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) gx2[o + i] = fmaf(w1[j * 16 + i], g, gx2[o + i]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) gx[i] = gx2[i] + gx2[16 + i];
  const Trig t = make_trig(&pre[3]);
  float gap[4] = {0.f, 0.f, 0.f, 0.f};
  const float w[3] = {pre[9], pre[10], pre[11]};
  quad_step_adjoint(lam, gap, ap[0], w, c, t);   // analytic part, lam in place
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] += gx[i];
#pragma unroll
  for (int r = 0; r < 4; ++r) gap[r] += gx[12 + r];
  cfloat_ptr L = m.linear_at;
  asm volatile("" : "+s"(L), "+v"(gap[0]));
#pragma unroll
  for (int i = 0; i < 4; ++i)     // linear_at^T
    ga[i] += L[i] * gap[0] + L[4 + i] * gap[1] + L[8 + i] * gap[2] +
             L[12 + i] * gap[3];
}

template <int LAYOUT>
__global__ __launch_bounds__(64) void quad_learnt_rollout_kernel(LearntArgs A) {
  extern __shared__ float stash[];
  const int lane = threadIdx.x;
  const int b = blockIdx.x * 64 + lane;
  const bool live = b < A.B;
  const int bb = live ? b : A.B - 1;
  const QuadConst c = A.c;
  const int H = A.H;
  const LearntWeights W(A.m);
  auto ST = [&](int k, int i) -> float & { return stash[(k * 12 + i) * 64 + lane]; };
  float s[12];
  load_state<LAYOUT, 12>(A.state0, A.B, bb, s);
  float sum_p = 0.f, sum_v = 0.f, sum_w = 0.f, sum_r = 0.f, sum_t = 0.f;
  for (int k = 0; k < H; ++k) {
    float a[4], rp[3], rv[3];
    load_seq<LAYOUT, 4>(A.actions, A.B, H, 4, bb, k, 0, a);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, 0, rp);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, A.vel_col, rv);
#pragma unroll
    for (int i = 0; i < 12; ++i) ST(k, i) = s[i];
    learnt_step(s, a, W, c);
    if (A.states_out && live)
      store_seq<LAYOUT, 12>(A.states_out, A.B, H, 12, b, k, 0, s);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float dp = s[i] - rp[i], dv = s[6 + i] - rv[i], d = a[1 + i] - 0.5f;
      sum_p = fmaf(dp, dp, sum_p), sum_v = fmaf(dv, dv, sum_v);
      sum_w = fmaf(s[9 + i], s[9 + i], sum_w), sum_r = fmaf(d, d, sum_r);
    }
    sum_t = fmaf(a[0] - 0.5f, a[0] - 0.5f, sum_t);
  }
  const float loss = A.w.pos * sum_p + A.w.vel * sum_v + A.w.av * sum_w +
                     A.w.rates * sum_r + A.w.thrust * sum_t;
  write_wave_partial(A.loss_partials, live ? loss : 0.f, (A.B + kWave - 1) / kWave);

  float lam[12], nxt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lam[i] = 0.f, nxt[i] = s[i];
  for (int k = H - 1; k >= 0; --k) {
    float a[4], rp[3], rv[3], pre[12];
    load_seq<LAYOUT, 4>(A.actions, A.B, H, 4, bb, k, 0, a);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, 0, rp);
    load_seq<LAYOUT, 3>(A.ref, A.B, H, A.ref_cols, bb, k, A.vel_col, rv);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      lam[i] += 2.f * A.w.pos * (nxt[i] - rp[i]);
      lam[6 + i] += 2.f * A.w.vel * (nxt[6 + i] - rv[i]);
      lam[9 + i] += 2.f * A.w.av * nxt[9 + i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) pre[i] = ST(k, i);
    float ga[4] = {2.f * A.w.thrust * (a[0] - 0.5f), 2.f * A.w.rates * (a[1] - 0.5f),
                   2.f * A.w.rates * (a[2] - 0.5f), 2.f * A.w.rates * (a[3] - 0.5f)};
    learnt_step_adjoint(lam, ga, pre, a, W, c);
    if (live) store_seq<LAYOUT, 4>(A.grad_actions, A.B, H, 4, b, k, 0, ga);
#pragma unroll
    for (int i = 0; i < 12; ++i) nxt[i] = pre[i];
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
  write_wave_partial(partials, live ? loss : 0.f, (B + kWave - 1) / kWave);
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
                 int layout, bool packed_ok = false) {
  if (B < 0) { set_error("B must be >= 0 (got %d)", B); return APG_ERR_ARG; }
  if (layout != APG_LAYOUT_SOA && layout != APG_LAYOUT_AOS &&
      !(packed_ok && layout == APG_LAYOUT_PACKED)) {
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

// APG_LAYOUT_PACKED: register-resident horizons only (the reference configs)
template <bool SO>
int launch_rollout_rows(const RolloutArgs &A, hipStream_t st) {
  if (A.H != 5 && A.H != 10) {
    set_error("APG_LAYOUT_PACKED: H must be 5 or 10 (got %d)", A.H);
    return APG_ERR_ARG;
  }
  if (A.ref_cols != 6) {
    set_error("APG_LAYOUT_PACKED: ref rows are [pos, vel] (ref_cols = 6)");
    return APG_ERR_ARG;
  }
  if ((long long)A.H * 3 * A.B * 16 >= (1ll << 31)) {
    set_error("APG_LAYOUT_PACKED: H * B too large for 32-bit buffer offsets");
    return APG_ERR_ARG;
  }
  RowArgs R;
  for (int i = 0; i < 3; ++i) R.wd[i] = A.c.wd[i];
  R.w = A.w;
  R.loss_partials = A.loss_partials, R.grad_actions = A.grad_actions;
  R.grad_state0 = A.grad_state0, R.states_out = A.states_out;
  R.prev = A.prev;
  const QuadConst &c = A.c;
  const dim3 grid(grid_for(A.B, rows_tune::kBlock)), block(rows_tune::kBlock);
#define APG_ROWS(HT)                                                        \
  hipLaunchKernelGGL((quad_rollout_rows_kernel<HT, SO>), grid, block, 0, st,  \
                     A.state0, A.actions, A.ref, A.B, c.dt, c.half_dt,        \
                     c.half_dt2, c.g[0], c.g[1], c.g[2], c.kdt[0], c.kdt[1],  \
                     c.kdt[2], R)
  if (A.H == 10) APG_ROWS(10);
  else APG_ROWS(5);
#undef APG_ROWS
  return check_launch("quad_rollout_fwd_bwd(packed)");
}

}  // namespace
}  // namespace apg

using namespace apg;

extern "C" {

#ifdef APG_STAMP
int apg_debug_set_stamps(unsigned long long *dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dev, sizeof(dev));
}
#endif

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
  if (int e = check_common(state0, actions, params, B, layout, true)) return e;
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
  if (layout == APG_LAYOUT_PACKED)
    e = states_out ? launch_rollout_rows<true>(A, st)
                   : launch_rollout_rows<false>(A, st);
  else if (layout == APG_LAYOUT_SOA)
    e = states_out ? launch_rollout<APG_LAYOUT_SOA, true>(A, st)
                   : launch_rollout<APG_LAYOUT_SOA, false>(A, st);
  else
    e = states_out ? launch_rollout<APG_LAYOUT_AOS, true>(A, st)
                   : launch_rollout<APG_LAYOUT_AOS, false>(A, st);
  if (e) return e;
  if (loss) return launch_reduce_partials(loss_partials, apg_loss_partials_count(B), loss, st);
  return APG_OK;
}

int apg_quad_learnt_rollout_fwd_bwd(const float *state0, const float *actions,
                                    const float *ref, int ref_cols, float dt,
                                    const ApgQuadParams *params,
                                    const ApgLearntResidual *model,
                                    const ApgQuadLossWeights *weights, int B, int H,
                                    int layout, float *loss_partials, float *loss,
                                    float *grad_actions, float *grad_state0,
                                    float *states_out, apg_stream_t stream) {
  if (int e = check_common(state0, actions, params, B, layout)) return e;
  if (!weights || !model || !model->linear_at || !model->w1 || !model->b1 ||
      !model->w2 || !model->b2) {
    set_error("weights / model tensors must not be NULL");
    return APG_ERR_ARG;
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
  if (B == 0) {
    if (loss && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess)
      return check_launch("memset(loss)");
    return APG_OK;
  }
  if (!ref || !loss_partials || !grad_actions) {
    set_error("ref / loss_partials / grad_actions must not be NULL");
    return APG_ERR_ARG;
  }
  LearntArgs A;
  A.state0 = state0, A.actions = actions, A.ref = ref;
  A.loss_partials = loss_partials, A.grad_actions = grad_actions;
  A.grad_state0 = grad_state0, A.states_out = states_out;
  A.m = *model;
  A.c = make_const(*params, dt);
  A.w = *weights;
  A.B = B, A.H = H, A.ref_cols = ref_cols, A.vel_col = ref_cols == 9 ? 6 : 3;
  const size_t lds = (size_t)H * 12 * 64 * sizeof(float);
  const dim3 grid(grid_for(B, 64)), block(64);
#define APG_LEARNT(L)                                                          \
  do {                                                                         \
    if (lds > 64 * 1024 &&                                                     \
        hipFuncSetAttribute((const void *)quad_learnt_rollout_kernel<L>,       \
                            hipFuncAttributeMaxDynamicSharedMemorySize,        \
                            (int)lds) != hipSuccess)                           \
      return check_launch("hipFuncSetAttribute(quad_learnt_rollout)");         \
    hipLaunchKernelGGL((quad_learnt_rollout_kernel<L>), grid, block, lds, st, A); \
  } while (0)
  if (layout == APG_LAYOUT_SOA) APG_LEARNT(APG_LAYOUT_SOA);
  else APG_LEARNT(APG_LAYOUT_AOS);
#undef APG_LEARNT
  if (int e = check_launch("quad_learnt_rollout_fwd_bwd")) return e;
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
