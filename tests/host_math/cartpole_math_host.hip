// Test infrastructure: the SHIPPED per-trajectory cart-pole arithmetic
// (csrc/cartpole_math.h) compiled for the HOST (see quad_math_host.hip).
#include "cartpole_math.h"

using namespace apg;

extern "C" void hm_cart_step(const float *state, const float *action, float dt,
                             const ApgCartpoleParams *p, int B, const float *cot,
                             float *next, float *gstate, float *gaction) {
  const CartConst c = make_const(*p, dt);
  for (int b = 0; b < B; ++b) {
    float s[4] = {state[b * 4], state[b * 4 + 1], state[b * 4 + 2], state[b * 4 + 3]};
    const float xd = s[1], thd = s[3];
    const CartAux x = cart_step(s, action[b], c);
    for (int i = 0; i < 4; ++i) next[b * 4 + i] = s[i];
    if (!cot) continue;
    float lam[4] = {cot[b * 4], cot[b * 4 + 1], cot[b * 4 + 2], cot[b * 4 + 3]};
    gaction[b] = cart_step_adjoint(lam, xd, thd, x, c);
    for (int i = 0; i < 4; ++i) gstate[b * 4 + i] = lam[i];
  }
}

// the composition of cart_rollout_kernel per lane: unroll, make_reference
// (scripts/train_cartpole.py:103-110), cartpole_loss_mpc, reverse sweep incl.
// the gradient through the reference
extern "C" double hm_cart_rollout(const float *state0, const float *actions, float dt,
                                  const ApgCartpoleParams *p, int B, int H,
                                  float *states, float *gactions, float *gstate0) {
  const CartConst c = make_const(*p, dt);
  const float wq[4] = {0.f, 3.f, 10.f, 1.f};
  const double inv = H > 1 ? 1.0 / (double)(H - 1) : 0.0;
  double total = 0.0;
  float(*pre)[4] = new float[H][4];
  for (int b = 0; b < B; ++b) {
    float s0[4], s[4], loss = 0.f;
    for (int i = 0; i < 4; ++i) s0[i] = s[i] = state0[b * 4 + i];
    const float *act = actions + (size_t)b * H;
    float *st = states + (size_t)b * H * 4;
    for (int k = 0; k < H; ++k) {
      for (int i = 0; i < 4; ++i) pre[k][i] = s[i];
      cart_step(s, act[k], c);
      const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
      for (int i = 0; i < 4; ++i) {
        st[k * 4 + i] = s[i];
        const float d = s[i] - s0[i] * f;
        loss += (d * d) * wq[i];
      }
      loss += 0.01f * act[k] * act[k];
    }
    float lam[4] = {0.f, 0.f, 0.f, 0.f}, g0[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = H - 1; k >= 0; --k) {
      const float f = k < H - 1 ? (float)(1.0 - inv * (double)k) : 0.f;
      for (int i = 0; i < 4; ++i) {
        const float seed = 2.f * wq[i] * (st[k * 4 + i] - s0[i] * f);
        lam[i] += seed;
        g0[i] -= seed * f;
      }
      float tmp[4] = {pre[k][0], pre[k][1], pre[k][2], pre[k][3]};
      const CartAux x = cart_step(tmp, act[k], c);
      gactions[(size_t)b * H + k] =
          cart_step_adjoint(lam, pre[k][1], pre[k][3], x, c) + 0.02f * act[k];
    }
    for (int i = 0; i < 4; ++i) gstate0[b * 4 + i] = lam[i] + g0[i];
    total += loss;
  }
  delete[] pre;
  return total;
}
