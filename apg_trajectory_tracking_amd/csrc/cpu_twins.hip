// libapg_cpu.so (include/apg_cpu.h): the host twins of the dynamics entry
// points - the SAME per-trajectory headers the GPU kernels call per lane
// (quad_math.h, wing_math.h, cartpole_math.h), compiled for the host
// (hipcc --cuda-host-only) and looped over the batch.  Separate library,
// separate symbol names, never loaded by the Python package: not a fallback.
// The compositions (forward sweep, loss terms and their seeds, reverse sweep)
// follow the rollout kernels of quad.hip / wing.hip / cartpole.hip lane for
// lane; tests/test_cpu_twins.py pins them to the golden vectors and, on a GPU,
// to the device entry points.
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "apg_cpu.h"
#include "cartpole_math.h"
#include "quad_math.h"
#include "wing_math.h"

using namespace apg;

namespace {

thread_local char g_err[512] = "";

int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return APG_ERR_ARG;
}

// element addresses of the three layouts of apg.h
struct Idx {
  int layout;
  size_t B;
  // component i of an S-component per-trajectory vector (state, action, cotangent)
  size_t vec(int b, int i, int S) const {
    if (layout == APG_LAYOUT_AOS) return (size_t)b * S + i;
    if (layout == APG_LAYOUT_SOA) return (size_t)i * B + b;
    return ((size_t)(i >> 2) * B + b) * 4 + (i & 3);          // [S/4][B][4]
  }
  // column c of row k of an [H][C] per-trajectory sequence
  size_t seq(int b, int k, int c, int H, int C) const {
    if (layout == APG_LAYOUT_AOS) return ((size_t)b * H + k) * C + c;
    if (layout == APG_LAYOUT_SOA) return ((size_t)k * C + c) * B + b;
    return ((size_t)k * B + b) * C + c;                        // [H][B][C]
  }
  // component i of the state after step k (states_out)
  size_t states(int b, int k, int i, int H, int S) const {
    if (layout == APG_LAYOUT_PACKED)                           // [H][S/4][B][4]
      return (((size_t)k * (S / 4) + (i >> 2)) * B + b) * 4 + (i & 3);
    return seq(b, k, i, H, S);
  }
};

int check_common(const void *params, int B, int layout, bool packed_ok) {
  if (B < 0) return fail("B must be >= 0 (got %d)", B);
  if (layout != APG_LAYOUT_AOS && layout != APG_LAYOUT_SOA &&
      !(packed_ok && layout == APG_LAYOUT_PACKED))
    return fail("unknown layout %d", layout);
  if (!params) return fail("params is NULL");
  return APG_OK;
}

int check_h(int H) {
  if (H < 1 || H > APG_MAX_HORIZON)
    return fail("H must be in [1, %d] (got %d)", APG_MAX_HORIZON, H);
  return APG_OK;
}

// one partial per 64 trajectories, then their fixed-order sum
struct LossOut {
  float *partials, *loss;
  float wave = 0.f;
  void add(int b, int B, float v) {
    wave += v;
    if ((b & 63) == 63 || b == B - 1) partials[b >> 6] = wave, wave = 0.f;
  }
  void finish(int B) {
    if (!loss) return;
    float s = 0.f;
    for (int i = 0; i < (B + 63) / 64; ++i) s += partials[i];
    *loss = s;
  }
};

int run_deferred(const ApgDeferredLoss *d) {
  if (!d) return APG_OK;
  if (!d->prev_loss || d->prev_count < 0 || (d->prev_count > 0 && !d->prev_partials))
    return fail("deferred: prev_loss NULL, prev_count < 0 or prev_partials NULL");
  float s = 0.f;
  for (int i = 0; i < d->prev_count; ++i) s += d->prev_partials[i];
  *d->prev_loss = s;
  return APG_OK;
}

}  // namespace

extern "C" {

int apg_cpu_version(void) { return APG_VERSION_MAJOR * 1000 + APG_VERSION_MINOR; }
const char *apg_cpu_last_error_string(void) { return g_err; }

// ------------------------------------------------------------------ quad
int apg_quad_step_fwd_cpu(const float *state, const float *action, float dt,
                          const ApgQuadParams *params, int B, int layout,
                          float *next_state) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!next_state) return fail("next_state is NULL");
  const QuadConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12], a[4];
    for (int i = 0; i < 12; ++i) s[i] = state[ix.vec(b, i, 12)];
    for (int i = 0; i < 4; ++i) a[i] = action[ix.vec(b, i, 4)];
    const Trig t = make_trig(&s[3]);
    quad_step(s, a, c, t);
    for (int i = 0; i < 12; ++i) next_state[ix.vec(b, i, 12)] = s[i];
  }
  return APG_OK;
}

int apg_quad_step_bwd_cpu(const float *state, const float *action, float dt,
                          const ApgQuadParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!grad_next) return fail("grad_next is NULL");
  const QuadConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12], lam[12], ga[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) s[i] = state[ix.vec(b, i, 12)];
    for (int i = 0; i < 12; ++i) lam[i] = grad_next[ix.vec(b, i, 12)];
    const Trig t = make_trig(&s[3]);
    quad_step_adjoint(lam, ga, action[ix.vec(b, 0, 4)], &s[9], c, t);
    if (grad_state)
      for (int i = 0; i < 12; ++i) grad_state[ix.vec(b, i, 12)] = lam[i];
    if (grad_action)
      for (int i = 0; i < 4; ++i) grad_action[ix.vec(b, i, 4)] = ga[i];
  }
  return APG_OK;
}

int apg_quad_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                 const float *ref, int ref_cols, float dt,
                                 const ApgQuadParams *params,
                                 const ApgQuadLossWeights *weights, int B, int H,
                                 int layout, float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out,
                                 const ApgDeferredLoss *deferred) {
  if (int e = check_common(params, B, layout, true)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (!weights) return fail("weights is NULL");
  if (int e = run_deferred(deferred)) return e;
  if (int e = check_h(H)) return e;
  if (ref_cols != 9 && ref_cols != 6)
    return fail("ref_cols must be 9 ([pos, euler, vel]) or 6 ([pos, vel])");
  if (layout == APG_LAYOUT_PACKED && H != 5 && H != 10)
    return fail("APG_LAYOUT_PACKED: H must be 5 or 10 (got %d)", H);
  if (layout == APG_LAYOUT_PACKED && ref_cols != 6)
    return fail("APG_LAYOUT_PACKED: ref rows are [pos, vel] (ref_cols = 6)");
  if (!ref || !loss_partials || !grad_actions)
    return fail("ref / loss_partials / grad_actions must not be NULL");
  const QuadConst c = make_const(*params, dt);
  const ApgQuadLossWeights &w = *weights;
  const Idx ix{layout, (size_t)B};
  const int vc = ref_cols == 9 ? 6 : 3;
  LossOut out{loss_partials, loss};
  std::vector<Trig> trig(H);
  std::vector<float> st((size_t)H * 12), wold((size_t)H * 3);
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[ix.vec(b, i, 12)];
    for (int k = 0; k < H; ++k) {
      for (int i = 0; i < 3; ++i) wold[k * 3 + i] = s[9 + i];
      trig[k] = make_trig(&s[3]);
      float a[4];
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, k, j, H, 4)];
      quad_step(s, a, c, trig[k]);
      for (int i = 0; i < 12; ++i) st[k * 12 + i] = s[i];
      if (states_out)
        for (int i = 0; i < 12; ++i) states_out[ix.states(b, k, i, H, 12)] = s[i];
    }
    // quad_mpc_loss (neural_control/drone_loss.py:12-39) and its seeds, then the
    // adjoint of step k - as quad_rollout_kernel's reverse sweep
    float lam[12] = {0.f}, l = 0.f;
    for (int k = H - 1; k >= 0; --k) {
      const float *x = &st[k * 12];
      float a[4], ga[4], lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, k, j, H, 4)];
      for (int i = 0; i < 3; ++i) {
        const float dp = x[i] - ref[ix.seq(b, k, i, H, ref_cols)];
        const float dv = x[6 + i] - ref[ix.seq(b, k, vc + i, H, ref_cols)];
        const float wn = x[9 + i];
        lp += dp * dp, lv += dv * dv, lw += wn * wn;
        lam[i] += 2.f * w.pos * dp;
        lam[6 + i] += 2.f * w.vel * dv;
        lam[9 + i] += 2.f * w.av * wn;
      }
      const float da0 = a[0] - 0.5f;
      ga[0] = 2.f * w.thrust * da0;
      for (int j = 1; j < 4; ++j) {
        const float d = a[j] - 0.5f;
        lr += d * d;
        ga[j] = 2.f * w.rates * d;
      }
      l += w.pos * lp + w.vel * lv + w.av * lw + w.rates * lr + w.thrust * da0 * da0;
      const float wo[3] = {wold[k * 3], wold[k * 3 + 1], wold[k * 3 + 2]};
      quad_step_adjoint(lam, ga, a[0], wo, c, trig[k]);
      for (int j = 0; j < 4; ++j) grad_actions[ix.seq(b, k, j, H, 4)] = ga[j];
    }
    if (grad_state0)
      for (int i = 0; i < 12; ++i) grad_state0[ix.vec(b, i, 12)] = lam[i];
    out.add(b, B, l);
  }
  out.finish(B);
  return APG_OK;
}

int apg_quad_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                             const ApgQuadParams *params, int B, int H, int layout,
                             float *states_out) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (H < 1) return fail("H must be >= 1 (got %d)", H);
  if (!states_out) return fail("states_out is NULL");
  const QuadConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[ix.vec(b, i, 12)];
    for (int k = 0; k < H; ++k) {
      float a[4];
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, k, j, H, 4)];
      quad_step(s, a, c, make_trig(&s[3]));
      for (int i = 0; i < 12; ++i) states_out[ix.seq(b, k, i, H, 12)] = s[i];
    }
  }
  return APG_OK;
}

// ------------------------------------------------------------------ wing
int apg_wing_step_fwd_cpu(const float *state, const float *action, float dt,
                          const ApgWingParams *params, int B, int layout,
                          float *next_state) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!next_state) return fail("next_state is NULL");
  const WingConst k = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12], a[4];
    for (int i = 0; i < 12; ++i) s[i] = state[ix.vec(b, i, 12)];
    for (int i = 0; i < 4; ++i) a[i] = action[ix.vec(b, i, 4)];
    wing_step(s, a, k);
    for (int i = 0; i < 12; ++i) next_state[ix.vec(b, i, 12)] = s[i];
  }
  return APG_OK;
}

int apg_wing_step_bwd_cpu(const float *state, const float *action, float dt,
                          const ApgWingParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!grad_next) return fail("grad_next is NULL");
  const WingConst k = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12], a[4], sd[12], lam[12], ga[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) s[i] = state[ix.vec(b, i, 12)];
    for (int i = 0; i < 4; ++i) a[i] = action[ix.vec(b, i, 4)];
    for (int i = 0; i < 12; ++i) lam[i] = grad_next[ix.vec(b, i, 12)];
    WingAux x;
    wing_rates(s, a, k, x, sd);
    wing_step_adjoint(lam, ga, s, x, sd, k);
    if (grad_state)
      for (int i = 0; i < 12; ++i) grad_state[ix.vec(b, i, 12)] = lam[i];
    if (grad_action)
      for (int i = 0; i < 4; ++i) grad_action[ix.vec(b, i, 4)] = ga[i];
  }
  return APG_OK;
}

int apg_wing_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                 const float *ref, float dt,
                                 const ApgWingParams *params,
                                 const ApgWingLossWeights *weights, int B, int H,
                                 int layout, float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out,
                                 const ApgDeferredLoss *deferred) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (!weights) return fail("weights is NULL");
  if (int e = run_deferred(deferred)) return e;
  if (int e = check_h(H)) return e;
  if (!ref || !loss_partials || !grad_actions)
    return fail("ref / loss_partials / grad_actions must not be NULL");
  const WingConst k = make_const(*params, dt);
  const ApgWingLossWeights &w = *weights;
  const Idx ix{layout, (size_t)B};
  LossOut out{loss_partials, loss};
  std::vector<float> pre((size_t)H * 12), st((size_t)H * 12);
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[ix.vec(b, i, 12)];
    for (int n = 0; n < H; ++n) {
      for (int i = 0; i < 12; ++i) pre[n * 12 + i] = s[i];
      float a[4];
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, n, j, H, 4)];
      wing_step(s, a, k);
      for (int i = 0; i < 12; ++i) st[n * 12 + i] = s[i];
      if (states_out)
        for (int i = 0; i < 12; ++i) states_out[ix.seq(b, n, i, H, 12)] = s[i];
    }
    // fixed_wing_mpc_loss (neural_control/drone_loss.py:72-82) with its seeds
    float lam[12] = {0.f}, l = 0.f;
    for (int n = H - 1; n >= 0; --n) {
      float a[4], ga[4] = {0.f, 0.f, 0.f, 0.f}, lp = 0.f, la = 0.f, sp[12], sd[12];
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, n, j, H, 4)];
      for (int i = 0; i < 12; ++i) sp[i] = pre[n * 12 + i];
      for (int i = 0; i < 3; ++i) {
        const float dp = st[n * 12 + i] - ref[ix.seq(b, n, i, H, 3)];
        lp += dp * dp;
        lam[i] += 2.f * w.pos * dp;
      }
      for (int j = 1; j < 4; ++j) {
        const float d = a[j] - 0.5f;
        la += d * d;
        ga[j] = 2.f * w.action * d;
      }
      l += w.pos * lp + w.action * la;
      WingAux x;
      wing_rates(sp, a, k, x, sd);
      wing_step_adjoint(lam, ga, sp, x, sd, k);
      for (int j = 0; j < 4; ++j) grad_actions[ix.seq(b, n, j, H, 4)] = ga[j];
    }
    if (grad_state0)
      for (int i = 0; i < 12; ++i) grad_state0[ix.vec(b, i, 12)] = lam[i];
    out.add(b, B, l);
  }
  out.finish(B);
  return APG_OK;
}

int apg_wing_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                             const ApgWingParams *params, int B, int H, int layout,
                             float *states_out) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (H < 1) return fail("H must be >= 1 (got %d)", H);
  if (!states_out) return fail("states_out is NULL");
  const WingConst k = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[ix.vec(b, i, 12)];
    for (int n = 0; n < H; ++n) {
      float a[4];
      for (int j = 0; j < 4; ++j) a[j] = actions[ix.seq(b, n, j, H, 4)];
      wing_step(s, a, k);
      for (int i = 0; i < 12; ++i) states_out[ix.seq(b, n, i, H, 12)] = s[i];
    }
  }
  return APG_OK;
}

// -------------------------------------------------------------- cartpole
int apg_cartpole_step_fwd_cpu(const float *state, const float *action, float dt,
                              const ApgCartpoleParams *params, int B, int layout,
                              float *next_state) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!next_state) return fail("next_state is NULL");
  const CartConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[4];
    for (int i = 0; i < 4; ++i) s[i] = state[ix.vec(b, i, 4)];
    cart_step(s, action[b], c);
    for (int i = 0; i < 4; ++i) next_state[ix.vec(b, i, 4)] = s[i];
  }
  return APG_OK;
}

int apg_cartpole_step_bwd_cpu(const float *state, const float *action, float dt,
                              const ApgCartpoleParams *params, int B, int layout,
                              const float *grad_next, float *grad_state,
                              float *grad_action) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state || !action)) return fail("NULL input pointer");
  if (!grad_next) return fail("grad_next is NULL");
  const CartConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[4], lam[4];
    for (int i = 0; i < 4; ++i) s[i] = state[ix.vec(b, i, 4)];
    for (int i = 0; i < 4; ++i) lam[i] = grad_next[ix.vec(b, i, 4)];
    const float xd = s[1], thd = s[3];
    const CartAux x = cart_step(s, action[b], c);
    const float ga = cart_step_adjoint(lam, xd, thd, x, c);
    if (grad_state)
      for (int i = 0; i < 4; ++i) grad_state[ix.vec(b, i, 4)] = lam[i];
    if (grad_action) grad_action[b] = ga;
  }
  return APG_OK;
}

int apg_cartpole_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                     float dt, const ApgCartpoleParams *params,
                                     int B, int H, int layout,
                                     float *loss_partials, float *loss,
                                     float *grad_actions, float *grad_state0,
                                     float *states_out) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (int e = check_h(H)) return e;
  if (!loss_partials || !grad_actions)
    return fail("loss_partials / grad_actions must not be NULL");
  const CartConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  // cartpole_loss_mpc (neural_control/drone_loss.py:136-145) against
  // make_reference (scripts/train_cartpole.py:103-110): the initial state
  // fading linearly to zero over the horizon; the gradient flows through it
  const float wq[4] = {0.f, 3.f, 10.f, 1.f};
  const double inv = H > 1 ? 1.0 / (double)(H - 1) : 0.0;
  LossOut out{loss_partials, loss};
  std::vector<float> pre((size_t)H * 4), st((size_t)H * 4);
  for (int b = 0; b < B; ++b) {
    float s0[4], s[4], l = 0.f;
    for (int i = 0; i < 4; ++i) s0[i] = s[i] = state0[ix.vec(b, i, 4)];
    for (int k = 0; k < H; ++k) {
      const float a = actions[ix.seq(b, k, 0, H, 1)];
      for (int i = 0; i < 4; ++i) pre[k * 4 + i] = s[i];
      cart_step(s, a, c);
      const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
      for (int i = 0; i < 4; ++i) {
        st[k * 4 + i] = s[i];
        if (states_out) states_out[ix.seq(b, k, i, H, 4)] = s[i];
        const float d = s[i] - s0[i] * f;
        l += (d * d) * wq[i];
      }
      l += 0.01f * a * a;
    }
    float lam[4] = {0.f, 0.f, 0.f, 0.f}, g0[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = H - 1; k >= 0; --k) {
      const float a = actions[ix.seq(b, k, 0, H, 1)];
      const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
      for (int i = 0; i < 4; ++i) {
        const float seed = 2.f * wq[i] * (st[k * 4 + i] - s0[i] * f);
        lam[i] += seed;
        g0[i] -= seed * f;
      }
      float tmp[4] = {pre[k * 4], pre[k * 4 + 1], pre[k * 4 + 2], pre[k * 4 + 3]};
      const CartAux x = cart_step(tmp, a, c);
      grad_actions[ix.seq(b, k, 0, H, 1)] =
          cart_step_adjoint(lam, pre[k * 4 + 1], pre[k * 4 + 3], x, c) + 0.02f * a;
    }
    if (grad_state0)
      for (int i = 0; i < 4; ++i) grad_state0[ix.vec(b, i, 4)] = lam[i] + g0[i];
    out.add(b, B, l);
  }
  out.finish(B);
  return APG_OK;
}

int apg_cartpole_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                                 const ApgCartpoleParams *params, int B, int H,
                                 int layout, float *states_out) {
  if (int e = check_common(params, B, layout, false)) return e;
  if (B > 0 && (!state0 || !actions)) return fail("NULL input pointer");
  if (H < 1) return fail("H must be >= 1 (got %d)", H);
  if (!states_out) return fail("states_out is NULL");
  const CartConst c = make_const(*params, dt);
  const Idx ix{layout, (size_t)B};
  for (int b = 0; b < B; ++b) {
    float s[4];
    for (int i = 0; i < 4; ++i) s[i] = state0[ix.vec(b, i, 4)];
    for (int k = 0; k < H; ++k) {
      cart_step(s, actions[ix.seq(b, k, 0, H, 1)], c);
      for (int i = 0; i < 4; ++i) states_out[ix.seq(b, k, i, H, 4)] = s[i];
    }
  }
  return APG_OK;
}

}  // extern "C"
