// Test infrastructure: the SHIPPED per-trajectory quadrotor arithmetic
// (csrc/quad_math.h, csrc/apg_device.h: sincos_fast, quad_step,
// quad_step_adjoint, quad_features and its adjoint - the functions every quad
// kernel calls per lane) compiled for the HOST, so that tests/ can pin it to
// the golden vectors without a GPU.  Row-major [B, ...] tensors.
#include "quad_math.h"

using namespace apg;

extern "C" void hm_sincos(const float *x, int n, float *s, float *c) {
  for (int i = 0; i < n; ++i) sincos_fast(x[i], &s[i], &c[i]);
}

extern "C" void hm_quad_step(const float *state, const float *action, float dt,
                             const ApgQuadParams *p, int B, const float *cot,
                             float *next, float *gstate, float *gaction) {
  const QuadConst c = make_const(*p, dt);
  for (int b = 0; b < B; ++b) {
    float s[12], a[4];
    for (int i = 0; i < 12; ++i) s[i] = state[b * 12 + i];
    for (int i = 0; i < 4; ++i) a[i] = action[b * 4 + i];
    const Trig t = make_trig(&s[3]);
    const float w[3] = {s[9], s[10], s[11]};
    quad_step(s, a, c, t);
    for (int i = 0; i < 12; ++i) next[b * 12 + i] = s[i];
    if (!cot) continue;
    float lam[12], ga[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) lam[i] = cot[b * 12 + i];
    quad_step_adjoint(lam, ga, a[0], w, c, t);
    for (int i = 0; i < 12; ++i) gstate[b * 12 + i] = lam[i];
    for (int i = 0; i < 4; ++i) gaction[b * 4 + i] = ga[i];
  }
}

extern "C" void hm_quad_features(const float *state, int B, const float *gf,
                                 float *feat, float *gstate) {
  for (int b = 0; b < B; ++b) {
    float s[12], f[15];
    for (int i = 0; i < 12; ++i) s[i] = state[b * 12 + i];
    const Trig t = make_trig(&s[3]);
    quad_features(s, t, f);
    for (int i = 0; i < 15; ++i) feat[b * 15 + i] = f[i];
    if (!gf) continue;
    float g[15], gs[12];
    for (int i = 0; i < 15; ++i) g[i] = gf[b * 15 + i];
    quad_features_adjoint(s, t, g, gs);
    for (int i = 0; i < 12; ++i) gstate[b * 12 + i] = gs[i];
  }
}

// the composition the rollout kernels perform per lane (forward sweep, loss
// terms and seeds of drone_loss.py:22-34, reverse sweep), written with the
// shipped step / adjoint
extern "C" double hm_quad_rollout(const float *state0, const float *actions,
                                  const float *ref, int ref_cols, float dt,
                                  const ApgQuadParams *p,
                                  const ApgQuadLossWeights *w, int B, int H,
                                  float *states, float *gactions, float *gstate0) {
  const QuadConst c = make_const(*p, dt);
  const int vc = ref_cols == 9 ? 6 : 3;
  double total = 0.0;
  Trig *trig = new Trig[H];
  float(*wold)[3] = new float[H][3];
  for (int b = 0; b < B; ++b) {
    float s[12];
    for (int i = 0; i < 12; ++i) s[i] = state0[b * 12 + i];
    const float *act = actions + (size_t)b * H * 4, *rf = ref + (size_t)b * H * ref_cols;
    float *st = states + (size_t)b * H * 12;
    for (int k = 0; k < H; ++k) {
      for (int i = 0; i < 3; ++i) wold[k][i] = s[9 + i];
      trig[k] = make_trig(&s[3]);
      const float a[4] = {act[k * 4], act[k * 4 + 1], act[k * 4 + 2], act[k * 4 + 3]};
      quad_step(s, a, c, trig[k]);
      for (int i = 0; i < 12; ++i) st[k * 12 + i] = s[i];
    }
    float lam[12] = {0.f}, loss = 0.f;
    for (int k = H - 1; k >= 0; --k) {
      const float *x = st + k * 12, *r = rf + k * ref_cols, *a = act + k * 4;
      float lp = 0.f, lv = 0.f, lw = 0.f, lr = 0.f;
      for (int i = 0; i < 3; ++i) {
        const float dp = x[i] - r[i], dv = x[6 + i] - r[vc + i], wn = x[9 + i];
        lp += dp * dp, lv += dv * dv, lw += wn * wn;
        lam[i] += 2.f * w->pos * dp;
        lam[6 + i] += 2.f * w->vel * dv;
        lam[9 + i] += 2.f * w->av * wn;
      }
      const float da0 = a[0] - 0.5f;
      float ga[4];
      ga[0] = 2.f * w->thrust * da0;
      for (int i = 1; i < 4; ++i) {
        const float d = a[i] - 0.5f;
        lr += d * d;
        ga[i] = 2.f * w->rates * d;
      }
      loss += w->pos * lp + w->vel * lv + w->av * lw + w->rates * lr +
              w->thrust * da0 * da0;
      quad_step_adjoint(lam, ga, a[0], wold[k], c, trig[k]);
      for (int i = 0; i < 4; ++i) gactions[((size_t)b * H + k) * 4 + i] = ga[i];
    }
    for (int i = 0; i < 12; ++i) gstate0[b * 12 + i] = lam[i];
    total += loss;
  }
  delete[] trig;
  delete[] wold;
  return total;
}
