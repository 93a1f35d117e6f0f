// quad_math.h - per-trajectory quadrotor arithmetic shared by the kernels of
// quad.hip, lstm.hip and mlp.hip (see quad.hip for the closed form and the
// reference lines it restates).  The functions are host-callable too, so that
// tests/host_math can run this very arithmetic on the CPU against the golden
// vectors.
#pragma once
#include "apg_device.h"

namespace apg {
namespace {

struct QuadConst {  // per-launch constants, derived on the host
  float dt, half_dt, half_dt2;
  float g[3];    // gravity + translational drag
  float kdt[3];  // dt * kinv
  float wd[3];   // dt * rot_drag / inertia
};

struct Trig {
  float sr, cr, sp, cp, sy, cy;  // roll(phi) pitch(theta) yaw(psi)
};

__host__ __device__ __forceinline__ Trig make_trig(const float att[3]) {
  Trig t;
  // device code: the hardware pair (7 issue slots per angle instead of ~30);
  // host builds (tests/host_math) and -DAPG_SW_TRIG: the branch-free software
  // pair, which stays the fixed-wing kernels' sin/cos as well
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APG_SW_TRIG)
  sincos_hw(att[0], &t.sr, &t.cr);
  sincos_hw(att[1], &t.sp, &t.cp);
  sincos_hw(att[2], &t.sy, &t.cy);
#else
  sincos_fast(att[0], &t.sr, &t.cr);
  sincos_fast(att[1], &t.sp, &t.cp);
  sincos_fast(att[2], &t.sy, &t.cy);
#endif
  return t;
}

__host__ __device__ __forceinline__ float thrust_of(float a0) {
  return a0 * 15.0f - 7.5f + 9.81f;  // quad_dynamics_flightmare.py:139
}

// thrust direction = third row of world_to_body (quad_dynamics_base.py:87-91)
__host__ __device__ __forceinline__ void thrust_dir(const Trig &t, float z[3]) {
  z[0] = t.cy * t.sp * t.cr + t.sr * t.sy;
  z[1] = t.cr * t.sy * t.sp - t.cy * t.sr;
  z[2] = t.cr * t.cp;
}

// s = [p(0:3), att(3:6), v(6:9), w(9:12)] updated in place.
__host__ __device__ __forceinline__ void quad_step(float (&s)[12], const float (&a)[4],
                                          const QuadConst &c, const Trig &t) {
  float z[3];
  thrust_dir(t, z);
  const float T = thrust_of(a[0]);
  const float w0 = s[9], w1 = s[10], w2 = s[11];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float acc = T * z[i] + c.g[i];
    s[i] = s[i] + c.half_dt2 * acc + c.half_dt * s[6 + i];
    s[6 + i] = s[6 + i] + c.dt * acc;
    s[9 + i] = s[9 + i] + c.kdt[i] * ((a[1 + i] - 0.5f) - s[9 + i]) + c.wd[i];
  }
  // euler_rate, quad_dynamics_base.py:96-127, with the OLD angular velocity
  s[3] += c.dt * (w0 - t.sp * w2);
  s[4] += c.dt * (t.cr * w1 + t.cp * t.sr * w2);
  s[5] += c.dt * (-t.sr * w1 + t.cp * t.cr * w2);
}

// Adjoint of quad_step.  lam = dL/d(next state) on entry, dL/d(state) on
// exit; ga += dL/d(action) through the dynamics.
//   lacc   = 1/2 dt^2 lam_p' + dt lam_v'          (cotangent of acc)
//   lam_v  = lam_v' + 1/2 dt lam_p'
//   ga0   += 15 (lacc . z),  ga_{1:3} += dt K lam_w'
//   lam_w  = (1 - dt K) lam_w' + dt E^T lam_att'
//   lam_att= lam_att' + dt (d(E w)/d att)^T lam_att' + (dz/d att)^T (T lacc)
// The last product is contracted through two scalars instead of forming the
// nine partials of z: with lz = T lacc, m = cy lz0 + sy lz1, n = sy lz0 - cy lz1
//   (dz/dphi)^T lz   = cr n - sr (sp m + cp lz2)
//   (dz/dtheta)^T lz = cr (cp m - sp lz2)
//   (dz/dpsi)^T lz   = sr m - sp cr n
// and d(E w)/d att shares r = sr la1 + cr la2 with E^T lam_att'.
__host__ __device__ __forceinline__ void quad_step_adjoint(float (&lam)[12],
                                                  float (&ga)[4], float a0,
                                                  const float w[3],
                                                  const QuadConst &c,
                                                  const Trig &t) {
  const float spcr = t.sp * t.cr;
  const float z0 = fmaf(t.cy, spcr, t.sr * t.sy);
  const float z1 = fmaf(t.sy, spcr, -(t.cy * t.sr));
  const float z2 = t.cr * t.cp;
  const float T = thrust_of(a0);
  float lacc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    lacc[i] = c.half_dt2 * lam[i] + c.dt * lam[6 + i];
    lam[6 + i] += c.half_dt * lam[i];  // dL/dv
  }
  ga[0] += 15.0f * (lacc[0] * z0 + lacc[1] * z1 + lacc[2] * z2);
  const float lz0 = T * lacc[0], lz1 = T * lacc[1], lz2 = T * lacc[2];
  const float la0 = lam[3], la1 = lam[4], la2 = lam[5];
  const float lw0 = lam[9], lw1 = lam[10], lw2 = lam[11];
  ga[1] += c.kdt[0] * lw0;
  ga[2] += c.kdt[1] * lw1;
  ga[3] += c.kdt[2] * lw2;
  const float r = t.sr * la1 + t.cr * la2;
  lam[9] = lw0 - c.kdt[0] * lw0 + c.dt * la0;
  lam[10] = lw1 - c.kdt[1] * lw1 + c.dt * (t.cr * la1 - t.sr * la2);
  lam[11] = lw2 - c.kdt[2] * lw2 + c.dt * (t.cp * r - t.sp * la0);
  // attitude rates e1, e2 of the step: d(E w)/dphi = (0, e2, -e1)
  const float cpw2 = t.cp * w[2];
  const float e1 = fmaf(t.cr, w[1], t.sr * cpw2);
  const float e2 = fmaf(t.cr, cpw2, -(t.sr * w[1]));
  const float gphi_e = la1 * e2 - la2 * e1;
  const float gth_e = -w[2] * (t.cp * la0 + t.sp * r);
  const float m = t.cy * lz0 + t.sy * lz1;
  const float n = t.sy * lz0 - t.cy * lz1;
  const float gphi_z = t.cr * n - t.sr * (t.sp * m + t.cp * lz2);
  const float gth_z = t.cr * (t.cp * m - t.sp * lz2);
  const float gpsi_z = t.sr * m - spcr * n;
  lam[3] = la0 + c.dt * gphi_e + gphi_z;
  lam[4] = la1 + c.dt * gth_e + gth_z;
  lam[5] = la2 + gpsi_z;
}

QuadConst make_const(const ApgQuadParams &p, float dt) {
  QuadConst c;
  c.dt = dt;
  c.half_dt = 0.5f * dt;
  c.half_dt2 = 0.5f * dt * dt;
  for (int i = 0; i < 3; ++i) {
    c.g[i] = p.gravity[i] + p.trans_drag[i];
    c.kdt[i] = dt * p.kinv[i];
    c.wd[i] = dt * (p.rot_drag[i] / p.inertia[i]);
  }
  return c;
}

// ------------------------------------------------------ policy-input features
// features = [v_world(3), R_wb[:, :, :2] row-major (6), v_body(3), w(3)]
struct Rot {
  float m[3][3];
};
__host__ __device__ __forceinline__ Rot world_to_body(const Trig &t) {
  Rot r;  // quad_dynamics_base.py:79-92
  r.m[0][0] = t.cy * t.cp, r.m[0][1] = t.sy * t.cp, r.m[0][2] = -t.sp;
  r.m[1][0] = t.cy * t.sp * t.sr - t.cr * t.sy;
  r.m[1][1] = t.cr * t.cy + t.sr * t.sy * t.sp;
  r.m[1][2] = t.cp * t.sr;
  r.m[2][0] = t.cy * t.sp * t.cr + t.sr * t.sy;
  r.m[2][1] = t.cr * t.sy * t.sp - t.cy * t.sr;
  r.m[2][2] = t.cr * t.cp;
  return r;
}

// state_preprocessing (neural_control/dataset.py:207-220) for one trajectory
__host__ __device__ __forceinline__ void quad_features(const float (&s)[12], const Trig &t,
                                              float (&f)[15]) {
  Rot r = world_to_body(t);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    f[i] = s[6 + i];
    f[3 + 2 * i] = r.m[i][0], f[3 + 2 * i + 1] = r.m[i][1];
    f[9 + i] = r.m[i][0] * s[6] + r.m[i][1] * s[7] + r.m[i][2] * s[8];
    f[12 + i] = s[9 + i];
  }
}

// its VJP: gs = (d features / d state)^T gf  (position gets zero)
__host__ __device__ __forceinline__ void quad_features_adjoint(const float (&s)[12],
                                                      const Trig &t,
                                                      const float (&gf)[15],
                                                      float (&gs)[12]) {
  Rot r = world_to_body(t);
  // cotangent of every rotation-matrix entry
  float gm[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gm[i][0] = gf[3 + 2 * i] + gf[9 + i] * s[6];
    gm[i][1] = gf[3 + 2 * i + 1] + gf[9 + i] * s[7];
    gm[i][2] = gf[9 + i] * s[8];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    gs[j] = 0.f;
    gs[6 + j] = gf[j] + r.m[0][j] * gf[9] + r.m[1][j] * gf[10] +
                r.m[2][j] * gf[11];
    gs[9 + j] = gf[12 + j];
  }
  // dR/droll, dR/dpitch, dR/dyaw contracted with gm
  gs[3] = gm[1][0] * (t.cy * t.sp * t.cr + t.sr * t.sy) +
          gm[1][1] * (-t.sr * t.cy + t.cr * t.sy * t.sp) +
          gm[1][2] * (t.cp * t.cr) +
          gm[2][0] * (-t.cy * t.sp * t.sr + t.cr * t.sy) +
          gm[2][1] * (-t.sr * t.sy * t.sp - t.cy * t.cr) +
          gm[2][2] * (-t.sr * t.cp);
  gs[4] = gm[0][0] * (-t.cy * t.sp) + gm[0][1] * (-t.sy * t.sp) +
          gm[0][2] * (-t.cp) + gm[1][0] * (t.cy * t.cp * t.sr) +
          gm[1][1] * (t.sr * t.sy * t.cp) + gm[1][2] * (-t.sp * t.sr) +
          gm[2][0] * (t.cy * t.cp * t.cr) + gm[2][1] * (t.cr * t.sy * t.cp) +
          gm[2][2] * (-t.cr * t.sp);
  gs[5] = gm[0][0] * (-t.sy * t.cp) + gm[0][1] * (t.cy * t.cp) +
          gm[1][0] * (-t.sy * t.sp * t.sr - t.cr * t.cy) +
          gm[1][1] * (-t.cr * t.sy + t.sr * t.cy * t.sp) +
          gm[2][0] * (-t.sy * t.sp * t.cr + t.sr * t.cy) +
          gm[2][1] * (t.cr * t.cy * t.sp + t.sy * t.sr);
}

}  // namespace
}  // namespace apg
