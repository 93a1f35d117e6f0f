// Test infrastructure: the SHIPPED per-trajectory fixed-wing arithmetic
// (csrc/wing_math.h: wing_rates, wing_step, wing_step_adjoint - what every
// fixed-wing kernel calls per lane) compiled for the HOST, so that tests/ can
// pin it to the golden vectors without a GPU.  Row-major [B, ...] tensors.
// (Host differences: sqrtf / correctly rounded quotient instead of the
// hardware v_sqrt_f32 / v_rcp_f32 seeds - both within an ulp.)
#include "wing_math.h"

using namespace apg;

extern "C" void hm_wing_step(const float *state, const float *action, float dt,
                             const ApgWingParams *p, int B, int steps,
                             const float *cot, float *next, float *gstate,
                             float *gaction) {
  const WingConst k = make_const(*p, dt);
  for (int b = 0; b < B; ++b) {
    float s[12], a[4], s0[12];
    for (int i = 0; i < 12; ++i) s0[i] = s[i] = state[b * 12 + i];
    for (int i = 0; i < 4; ++i) a[i] = action[b * 4 + i];
    for (int n = 0; n < steps; ++n) wing_step(s, a, k);
    for (int i = 0; i < 12; ++i) next[b * 12 + i] = s[i];
    if (!cot) continue;
    WingAux x;
    float sd[12], lam[12], ga[4] = {0.f, 0.f, 0.f, 0.f};
    wing_rates(s0, a, k, x, sd);
    for (int i = 0; i < 12; ++i) lam[i] = cot[b * 12 + i];
    wing_step_adjoint(lam, ga, s0, x, sd, k);
    for (int i = 0; i < 12; ++i) gstate[b * 12 + i] = lam[i];
    for (int i = 0; i < 4; ++i) gaction[b * 4 + i] = ga[i];
  }
}

// the composition of the rollout kernel per lane: unroll, fixed_wing_mpc_loss
// (drone_loss.py:72-82) with its seeds, reverse sweep
extern "C" double hm_wing_rollout(const float *state0, const float *actions,
                                  const float *ref, float dt, const ApgWingParams *p,
                                  const ApgWingLossWeights *w, int B, int H,
                                  float *states, float *gactions, float *gstate0) {
  const WingConst k = make_const(*p, dt);
  double total = 0.0;
  float(*pre)[12] = new float[H][12];
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[b * 12 + i];
    const float *act = actions + (size_t)b * H * 4, *rf = ref + (size_t)b * H * 3;
    float *st = states + (size_t)b * H * 12;
    for (int n = 0; n < H; ++n) {
      for (int i = 0; i < 12; ++i) pre[n][i] = s[i];
      const float a[4] = {act[n * 4], act[n * 4 + 1], act[n * 4 + 2], act[n * 4 + 3]};
      wing_step(s, a, k);
      for (int i = 0; i < 12; ++i) st[n * 12 + i] = s[i];
    }
    float lam[12] = {0.f}, loss = 0.f;
    for (int n = H - 1; n >= 0; --n) {
      const float a[4] = {act[n * 4], act[n * 4 + 1], act[n * 4 + 2], act[n * 4 + 3]};
      float lp = 0.f, la = 0.f, ga[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < 3; ++i) {
        const float dp = st[n * 12 + i] - rf[n * 3 + i];
        lp += dp * dp;
        lam[i] += 2.f * w->pos * dp;
      }
      for (int i = 1; i < 4; ++i) {
        const float d = a[i] - 0.5f;
        la += d * d;
        ga[i] = 2.f * w->action * d;
      }
      loss += w->pos * lp + w->action * la;
      WingAux x;
      float sd[12];
      wing_rates(pre[n], a, k, x, sd);
      wing_step_adjoint(lam, ga, pre[n], x, sd, k);
      for (int i = 0; i < 4; ++i) gactions[((size_t)b * H + n) * 4 + i] = ga[i];
    }
    for (int i = 0; i < 12; ++i) gstate0[b * 12 + i] = lam[i];
    total += loss;
  }
  delete[] pre;
  return total;
}
