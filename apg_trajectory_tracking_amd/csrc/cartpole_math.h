// cartpole_math.h - per-trajectory cart-pole arithmetic of cartpole.hip (see
// cartpole.hip for the reference lines it restates): constants, one step with
// what the adjoint re-uses, the adjoint.  Host-callable as well, so that
// tests/host_math can run this very arithmetic on the CPU against the golden
// vectors.
#pragma once
#include <math.h>

#include "apg_device.h"

namespace apg {
namespace {

struct CartConst {
  float dt, force_scale, mu;
  float pml, mp_g3, tm4, mp3, tm_g6, l_tm4, pml3;
};

CartConst make_const(const ApgCartpoleParams &p, float dt) {
  CartConst c;
  const float tm = p.masspole + p.masscart;
  const float pml = p.masspole * p.length;
  c.dt = dt;
  c.force_scale = p.max_force_mag * 0.5f;
  c.mu = p.friction;
  c.pml = pml;
  c.mp_g3 = 3.f * p.masspole * p.gravity;
  c.tm4 = 4.f * tm;
  c.mp3 = 3.f * p.masspole;
  c.tm_g6 = 6.f * tm * p.gravity;
  c.l_tm4 = 4.f * p.length * tm;
  c.pml3 = 3.f * pml;
  return c;
}

struct CartAux {  // what the adjoint needs from the forward evaluation
  float s, co, den_x, den_t, xacc, thacc, force;
};

// state = [x, x_dot, theta, theta_dot], in place.
__host__ __device__ __forceinline__ CartAux cart_step(float (&st)[4], float a,
                                             const CartConst &c) {
  CartAux x;
  x.force = a * c.force_scale;
  const float xd = st[1], thd = st[3];
  sincosf(st[2], &x.s, &x.co);
  x.den_x = c.tm4 - c.mp3 * x.co * x.co;
  x.den_t = c.l_tm4 - c.pml3 * x.co * x.co;
  x.xacc = (-2.f * c.pml * (thd * thd) * x.s + c.mp_g3 * x.s * x.co +
            4.f * x.force - 4.f * c.mu * xd) / x.den_x;
  x.thacc = (-c.pml3 * (thd * thd) * x.s * x.co + c.tm_g6 * x.s +
             6.f * (x.force - c.mu * xd) * x.co) / x.den_t;
  float sd, cd;
  sincosf(thd * c.dt, &sd, &cd);
  const float ns = x.s * cd + x.co * sd, nc = x.co * cd - x.s * sd;
  st[0] = st[0] + xd * c.dt;
  st[1] = xd + x.xacc * c.dt;
  st[2] = atan2f(ns, nc);
  st[3] = thd + x.thacc * c.dt;
  return x;
}

// lam: dL/dnext on entry, dL/dstate on exit; returns dL/daction.
// (xd, thd) are the PRE-step velocities.
__host__ __device__ __forceinline__ float cart_step_adjoint(float (&lam)[4], float xd,
                                                   float thd, const CartAux &x,
                                                   const CartConst &c) {
  const float lxa = c.dt * lam[1], lta = c.dt * lam[3];
  const float ix = 1.f / x.den_x, it = 1.f / x.den_t;
  const float c2s2 = x.co * x.co - x.s * x.s;
  const float dnx_th = -2.f * c.pml * thd * thd * x.co + c.mp_g3 * c2s2;
  const float ddx_th = 2.f * c.mp3 * x.co * x.s;
  const float dnt_th = -c.pml3 * thd * thd * c2s2 + c.tm_g6 * x.co -
                       6.f * (x.force - c.mu * xd) * x.s;
  const float ddt_th = 2.f * c.pml3 * x.co * x.s;
  const float l_x = lam[0];
  const float l_xd = lam[1] + c.dt * lam[0] + lxa * (-4.f * c.mu * ix) +
                     lta * (-6.f * c.mu * x.co * it);
  const float l_th = lam[2] + lxa * (dnx_th - x.xacc * ddx_th) * ix +
                     lta * (dnt_th - x.thacc * ddt_th) * it;
  const float l_thd = lam[3] + c.dt * lam[2] +
                      lxa * (-4.f * c.pml * thd * x.s * ix) +
                      lta * (-2.f * c.pml3 * thd * x.s * x.co * it);
  const float l_f = lxa * 4.f * ix + lta * 6.f * x.co * it;
  lam[0] = l_x, lam[1] = l_xd, lam[2] = l_th, lam[3] = l_thd;
  return l_f * c.force_scale;
}

}  // namespace
}  // namespace apg
