// quad_lane_pk.h - the quadrotor rollout of ONE trajectory (forward sweep,
// quad_mpc_loss, reverse sweep) written on 2-wide float vectors so that the
// compiler emits packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32: two lanes of one register pair per issue slot, operand
// halves chosen by op_sel, signs by neg_lo / neg_hi - no shuffles).
//
// Same arithmetic as quad_math.h (see quad.hip for the closed form and the
// reference lines); what differs is the grouping:
//   * 3-vectors are an (x, y) pair + z;  attitude is phi + a (theta, psi) pair;
//   * every angle keeps its (sin, cos) pair: R (roll), P (pitch), Y (yaw);
//     pitch and yaw share one packed polynomial evaluation;
//   * swapped / negated operands are written as `(f2){x, -x} * v.yx` - the
//     form the compiler folds into op_sel / neg_hi instead of moves;
//   * (dz/datt)^T lz is contracted through m = cy lz0 + sy lz1 and
//     n = sy lz0 - cy lz1 instead of forming the nine partials.
// Host + device: tests/ compiles this very file for the CPU and checks the
// lane algorithm against the golden vectors; the kernel supplies the plane
// accessors (quad.hip, quad_rollout_pk_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include "apg.h"

#define APG_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define APG_PIN_ORDER() __builtin_amdgcn_sched_barrier(0)
#else
#define APG_PIN_ORDER() ((void)0)
#endif

namespace apg {
namespace pk {

typedef float f2 __attribute__((ext_vector_type(2)));

APG_HD f2 bc(float x) { return (f2){x, x}; }
APG_HD f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
APG_HD float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
APG_HD float hsum(f2 v) { return v.x + v.y; }

struct Const {  // per-launch constants (QuadConst regrouped)
  float dt, hd, hd2;
  f2 g01, kdt01, wd01;
  float g2, kdt2, wd2;
};

// the grouping of quad_math.h's make_const
inline Const make_const(const ApgQuadParams &p, float dt) {
  Const c;
  c.dt = dt, c.hd = 0.5f * dt, c.hd2 = 0.5f * dt * dt;
  float g[3], k[3], wd[3];
  for (int i = 0; i < 3; ++i) {
    g[i] = p.gravity[i] + p.trans_drag[i];
    k[i] = dt * p.kinv[i];
    wd[i] = dt * (p.rot_drag[i] / p.inertia[i]);
  }
  c.g01 = (f2){g[0], g[1]}, c.g2 = g[2];
  c.kdt01 = (f2){k[0], k[1]}, c.kdt2 = k[2];
  c.wd01 = (f2){wd[0], wd[1]}, c.wd2 = wd[2];
  return c;
}

struct Trig {
  f2 R;  // (sin phi, cos phi)
  f2 P;  // (sin theta, cos theta)
  f2 Y;  // (sin psi, cos psi)
};

struct State {
  f2 p01, tp, v01, w01;  // tp = (theta, psi)
  float p2, phi, v2, w2;
};

// quadrant fix-up shared by the scalar and the packed evaluation
APG_HD void sincos_select(float s, float c, int k, float *sn, float *cs) {
  const bool swap = (k & 1) != 0;
  const float so = swap ? c : s, co = swap ? s : c;
  *sn = __builtin_bit_cast(
      float, __builtin_bit_cast(unsigned, so) ^ ((unsigned)(k & 2) << 30));
  *cs = __builtin_bit_cast(
      float, __builtin_bit_cast(unsigned, co) ^ ((unsigned)((k + 1) & 2) << 30));
}

// apg_device.h sincos_fast, one angle
APG_HD f2 sincos1(float x) {
  const float kf = __builtin_rintf(x * 0.6366197466850281f);
  float r = fma1(-kf, 1.5707963705062866f, x);
  r = fma1(-kf, -4.371138828673793e-08f, r);
  r = fma1(-kf, -1.7151245100058819e-15f, r);
  const float t = r * r;
  float ps = fma1(t, 2.6658919978217455e-06f, -0.0001983463589567691f);
  ps = fma1(t, ps, 0.008333319798111916f);
  ps = fma1(t, ps, -0.1666666716337204f);
  const float s = fma1(r * t, ps, r);
  float pc = fma1(t, -4.336599204179947e-07f, 2.494495674909558e-05f);
  pc = fma1(t, pc, -0.0013889188412576914f);
  pc = fma1(t, pc, 0.0416666679084301f);
  const float c = fma1(t * t, pc, fma1(t, -0.5f, 1.0f));
  f2 out;
  float sn, cs;
  sincos_select(s, c, (int)kf, &sn, &cs);
  out.x = sn, out.y = cs;
  return out;
}

// the same polynomial on two angles at once; returns their (sin, cos) pairs
APG_HD void sincos2(f2 x, f2 *a, f2 *b) {
  const f2 kf = __builtin_elementwise_rint(x * 0.6366197466850281f);
  const f2 nk = -kf;
  f2 r = fma2(nk, bc(1.5707963705062866f), x);
  r = fma2(nk, bc(-4.371138828673793e-08f), r);
  r = fma2(nk, bc(-1.7151245100058819e-15f), r);
  const f2 t = r * r;
  f2 ps = fma2(t, bc(2.6658919978217455e-06f), bc(-0.0001983463589567691f));
  ps = fma2(t, ps, bc(0.008333319798111916f));
  ps = fma2(t, ps, bc(-0.1666666716337204f));
  const f2 s = fma2(r * t, ps, r);
  f2 pc = fma2(t, bc(-4.336599204179947e-07f), bc(2.494495674909558e-05f));
  pc = fma2(t, pc, bc(-0.0013889188412576914f));
  pc = fma2(t, pc, bc(0.0416666679084301f));
  const f2 c = fma2(t * t, pc, fma2(t, bc(-0.5f), bc(1.0f)));
  float s0, c0, s1, c1;
  sincos_select(s.x, c.x, (int)kf.x, &s0, &c0);
  sincos_select(s.y, c.y, (int)kf.y, &s1, &c1);
  *a = (f2){s0, c0};
  *b = (f2){s1, c1};
}

APG_HD Trig make_trig(float phi, f2 tp) {
  Trig t;
  t.R = sincos1(phi);
  sincos2(tp, &t.P, &t.Y);
  return t;
}

APG_HD float thrust_of(float a0) { return a0 * 15.0f - 7.5f + 9.81f; }

// thrust direction: (z0, z1) = sp cr (cy, sy) + sr (sy, -cy), z2 = cr cp
struct Dir {
  f2 z01;
  float z2, spcr;
};
APG_HD Dir thrust_dir(const Trig &t) {
  Dir d;
  d.spcr = t.P.x * t.R.y;
  d.z01 = fma2(bc(d.spcr), t.Y.yx, (f2){t.R.x, -t.R.x} * t.Y);
  d.z2 = t.R.y * t.P.y;
  return d;
}

// body-rate part of the attitude derivative for the (theta, psi) pair:
// (cr w1 + sr cp w2, -sr w1 + cr cp w2)
APG_HD f2 rate_tp(const Trig &t, float w1, float cpw2) {
  return fma2(bc(cpw2), t.R, (f2){w1, -w1} * t.R.yx);
}

APG_HD void step(State &s, float a0, f2 a12, float a3, const Const &c,
                 const Trig &t) {
  const Dir d = thrust_dir(t);
  const float T = thrust_of(a0);
  const f2 acc01 = fma2(bc(T), d.z01, c.g01);
  const float acc2 = fma1(T, d.z2, c.g2);
  const float w0 = s.w01.x, w1 = s.w01.y, w2 = s.w2;
  s.p01 = fma2(bc(c.hd), s.v01, fma2(bc(c.hd2), acc01, s.p01));
  s.p2 = fma1(c.hd, s.v2, fma1(c.hd2, acc2, s.p2));
  s.v01 = fma2(bc(c.dt), acc01, s.v01);
  s.v2 = fma1(c.dt, acc2, s.v2);
  s.w01 = fma2(c.kdt01, (a12 - 0.5f) - s.w01, s.w01) + c.wd01;
  s.w2 = fma1(c.kdt2, (a3 - 0.5f) - w2, w2) + c.wd2;
  // attitude with the OLD body rates
  s.phi = fma1(c.dt, fma1(-t.P.x, w2, w0), s.phi);
  s.tp = fma2(bc(c.dt), rate_tp(t, w1, t.P.y * w2), s.tp);
}

struct Adj {  // dL/d(state), grouped like State
  f2 p01, tp, v01, w01;
  float p2, phi, v2, w2;
};

// Adjoint of step(): l = dL/d(next state) on entry, dL/d(state) on exit;
// ga += dL/d(action) through the dynamics.  w01 / w2: the OLD body rates.
APG_HD void step_adjoint(Adj &l, float &ga0, f2 &ga12, float &ga3, float a0,
                         f2 w01, float w2, const Const &c, const Trig &t) {
  const Dir d = thrust_dir(t);
  const float T = thrust_of(a0);
  const float sr = t.R.x, cr = t.R.y, sp = t.P.x, cp = t.P.y, sy = t.Y.x,
              cy = t.Y.y;
  const f2 lacc01 = fma2(bc(c.hd2), l.p01, bc(c.dt) * l.v01);
  const float lacc2 = fma1(c.hd2, l.p2, c.dt * l.v2);
  l.v01 = fma2(bc(c.hd), l.p01, l.v01);
  l.v2 = fma1(c.hd, l.p2, l.v2);
  ga0 = fma1(15.0f, fma1(lacc2, d.z2, hsum(lacc01 * d.z01)), ga0);
  const f2 lz01 = bc(T) * lacc01;
  const float lz2 = T * lacc2;
  const float la0 = l.phi;
  const f2 la12 = l.tp;
  ga12 = fma2(c.kdt01, l.w01, ga12);
  ga3 = fma1(c.kdt2, l.w2, ga3);
  // dL/dw = (1 - dt K) lam_w' + dt E^T lam_att'
  const f2 ru = t.R.yx * la12;
  const float e1 = ru.x - ru.y;        // cr la1 - sr la2
  const float e2s = hsum(t.R * la12);  // sr la1 + cr la2
  const f2 lw01 = l.w01;
  const float lw2 = l.w2;
  l.w01 = fma2(bc(c.dt), (f2){la0, e1}, fma2(-c.kdt01, lw01, lw01));
  l.w2 = fma1(c.dt, fma1(-sp, la0, cp * e2s), fma1(-c.kdt2, lw2, lw2));
  // dL/datt = lam_att' + dt (d(E w)/datt)^T lam_att' + (dz/datt)^T (T lacc)
  const float cpw2 = cp * w2, spw2 = sp * w2;
  const f2 X = rate_tp(t, w01.y, cpw2);
  const f2 xd = X.yx * la12;
  const float dphi = xd.x - xd.y;
  const float dth = fma1(-cpw2, la0, -spw2 * e2s);
  const float m = fma1(sy, lz01.y, cy * lz01.x);
  const float n = fma1(-cy, lz01.y, sy * lz01.x);
  const float gphi = fma1(-(sr * cp), lz2, fma1(cr, n, -(sp * sr) * m));
  const float gth = fma1(-d.spcr, lz2, d.z2 * m);
  const float gpsi = fma1(sr, m, -d.spcr * n);
  l.phi = fma1(c.dt, dphi, la0) + gphi;
  l.tp = la12 + (f2){fma1(c.dt, dth, gth), gpsi};
}

// ---------------------------------------------------------------- the lane
// IO supplies the trajectory's data:
//   float s0(int i)                 state0 component i (0..11: p, att, v, w)
//   float act(int k, int i)         action i of step k
//   float ref_p(int k, int i), ref_v(int k, int i)   reference position / velocity
//   void  ga(int k, int i, float)   dL/daction
//   void  gs(int i, float)          dL/dstate0 (only called if want_gs)
//   void  state(int k, int i, float) rollout state (only if STATES_OUT)
// Request schedule as in quad_rollout_reg_kernel: state0 and the first
// kActPre action rows up front, then one action row and one reference row
// (last first) per forward step.
template <int HT, bool STATES_OUT, class IO>
APG_HD float rollout_lane(IO &io, const Const &c, const ApgQuadLossWeights &w,
                          bool want_gs) {
  constexpr int kActPre = HT < 3 ? HT : 3;
  State s;
  {
    float r[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) r[i] = io.s0(i);
    s.p01 = (f2){r[0], r[1]}, s.p2 = r[2];
    s.phi = r[3], s.tp = (f2){r[4], r[5]};
    s.v01 = (f2){r[6], r[7]}, s.v2 = r[8];
    s.w01 = (f2){r[9], r[10]}, s.w2 = r[11];
  }
  APG_PIN_ORDER();
  float a0[HT], a3[HT];
  f2 a12[HT];
  f2 rp01[HT], rv01[HT];
  float rp2[HT], rv2[HT];
  auto ld_act = [&](int k) {
    a0[k] = io.act(k, 0);
    a12[k] = (f2){io.act(k, 1), io.act(k, 2)};
    a3[k] = io.act(k, 3);
  };
#pragma unroll
  for (int k = 0; k < kActPre; ++k) {
    ld_act(k);
    APG_PIN_ORDER();
  }
  Trig st_trig[HT];
  f2 st_w01[HT + 1], st_p01[HT], st_v01[HT];
  float st_w2[HT + 1], st_p2[HT], st_v2[HT];
#pragma unroll
  for (int k = 0; k < HT; ++k) {
    {
      const int kr = HT - 1 - k;
      if (k + kActPre < HT) ld_act(k + kActPre);
      rp01[kr] = (f2){io.ref_p(kr, 0), io.ref_p(kr, 1)};
      rp2[kr] = io.ref_p(kr, 2);
      rv01[kr] = (f2){io.ref_v(kr, 0), io.ref_v(kr, 1)};
      rv2[kr] = io.ref_v(kr, 2);
      APG_PIN_ORDER();
    }
    st_w01[k] = s.w01, st_w2[k] = s.w2;
    st_trig[k] = make_trig(s.phi, s.tp);
    step(s, a0[k], a12[k], a3[k], c, st_trig[k]);
    st_p01[k] = s.p01, st_p2[k] = s.p2, st_v01[k] = s.v01, st_v2[k] = s.v2;
    if constexpr (STATES_OUT) {
      io.state(k, 0, s.p01.x), io.state(k, 1, s.p01.y), io.state(k, 2, s.p2);
      io.state(k, 3, s.phi), io.state(k, 4, s.tp.x), io.state(k, 5, s.tp.y);
      io.state(k, 6, s.v01.x), io.state(k, 7, s.v01.y), io.state(k, 8, s.v2);
      io.state(k, 9, s.w01.x), io.state(k, 10, s.w01.y), io.state(k, 11, s.w2);
    }
  }
  st_w01[HT] = s.w01, st_w2[HT] = s.w2;

  // loss terms accumulate pairwise over the steps; weights applied once
  f2 sp01 = bc(0.f), sv01 = bc(0.f), sw01 = bc(0.f), sr12 = bc(0.f);
  float sp2 = 0.f, sv2 = 0.f, sw2 = 0.f, sr3 = 0.f, st0 = 0.f;
  Adj l;
  l.p01 = l.tp = l.v01 = l.w01 = bc(0.f);
  l.p2 = l.phi = l.v2 = l.w2 = 0.f;
  const float kp = 2.f * w.pos, kv = 2.f * w.vel, kw = 2.f * w.av,
              kr_ = 2.f * w.rates, kt = 2.f * w.thrust;
#pragma unroll
  for (int k = HT - 1; k >= 0; --k) {
    // loss terms of step k (drone_loss.py:22-34) and their seeds
    const f2 dp01 = st_p01[k] - rp01[k], dv01 = st_v01[k] - rv01[k];
    const float dp2 = st_p2[k] - rp2[k], dv2 = st_v2[k] - rv2[k];
    const f2 wn01 = st_w01[k + 1];
    const float wn2 = st_w2[k + 1];
    sp01 = fma2(dp01, dp01, sp01), sp2 = fma1(dp2, dp2, sp2);
    sv01 = fma2(dv01, dv01, sv01), sv2 = fma1(dv2, dv2, sv2);
    sw01 = fma2(wn01, wn01, sw01), sw2 = fma1(wn2, wn2, sw2);
    l.p01 = fma2(bc(kp), dp01, l.p01), l.p2 = fma1(kp, dp2, l.p2);
    l.v01 = fma2(bc(kv), dv01, l.v01), l.v2 = fma1(kv, dv2, l.v2);
    l.w01 = fma2(bc(kw), wn01, l.w01), l.w2 = fma1(kw, wn2, l.w2);
    const float da0 = a0[k] - 0.5f, da3 = a3[k] - 0.5f;
    const f2 da12 = a12[k] - 0.5f;
    st0 = fma1(da0, da0, st0);
    sr12 = fma2(da12, da12, sr12), sr3 = fma1(da3, da3, sr3);
    float g0 = kt * da0, g3 = kr_ * da3;
    f2 g12 = bc(kr_) * da12;
    step_adjoint(l, g0, g12, g3, a0[k], st_w01[k], st_w2[k], c, st_trig[k]);
    io.ga(k, 0, g0), io.ga(k, 1, g12.x), io.ga(k, 2, g12.y), io.ga(k, 3, g3);
  }
  if (want_gs) {
    io.gs(0, l.p01.x), io.gs(1, l.p01.y), io.gs(2, l.p2);
    io.gs(3, l.phi), io.gs(4, l.tp.x), io.gs(5, l.tp.y);
    io.gs(6, l.v01.x), io.gs(7, l.v01.y), io.gs(8, l.v2);
    io.gs(9, l.w01.x), io.gs(10, l.w01.y), io.gs(11, l.w2);
  }
  return w.pos * (hsum(sp01) + sp2) + w.vel * (hsum(sv01) + sv2) +
         w.av * (hsum(sw01) + sw2) + w.rates * (hsum(sr12) + sr3) +
         w.thrust * st0;
}

}  // namespace pk
}  // namespace apg
