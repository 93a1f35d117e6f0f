// wing_math.h - per-trajectory fixed-wing arithmetic of wing.hip (see wing.hip
// for the reference lines it restates): constants, one step's state
// derivative with everything the adjoint re-uses, the explicit-Euler step and
// its adjoint.  Host-callable as well, so that tests/host_math can run this
// very arithmetic on the CPU against the golden vectors.
#pragma once
#include <math.h>
#include <string.h>

#include <type_traits>

#include "apg_device.h"

namespace apg {
namespace {

struct WingConst {
  float dt;
  float half_rho, S, c, inv_mass, g_m;
  float cos_eps, sin_eps, alpha_bound;
  // coefficient tables; the rate derivatives are pre-multiplied by c or b as
  // the reference does in double before touching a tensor (:139-164)
  float CL0, CL_a, CL_qc, CL_de;
  float CD0, CD_a, CD_qc, CD_de;
  float CY0, CY_b, CY_pb, CY_rb, CY_da, CY_dr;
  float Cl0, Cl_b, Cl_pb, Cl_rb, Cl_da, Cl_dr;
  float Cm0, Cm_a, Cm_qc, Cm_de;
  float Cn0, Cn_b, Cn_pb, Cn_rb, Cn_da, Cn_dr;
  float Ixx, Iyy, Izz, a13;       // inertia matrix entries (a13 = -I_xz)
  float i00, i02, i11, i22;       // its inverse
};

WingConst make_const(const ApgWingParams &p, float dt) {
  WingConst k;
  k.dt = dt;
  k.half_rho = (float)(0.5 * (double)p.rho);
  k.S = p.S, k.c = p.c;
  k.inv_mass = (float)(1.0 / (double)p.mass);
  k.g_m = (float)((double)p.g * (double)p.mass);
  k.cos_eps = cosf(p.epsilon), k.sin_eps = sinf(p.epsilon);
  k.alpha_bound = (float)(10.0 / 180.0 * 3.14159265358979323846);
  const double c = p.c, b = p.b;
  k.CL0 = p.CL0, k.CL_a = p.CL_alpha, k.CL_qc = (float)(p.CL_q * c), k.CL_de = p.CL_del_e;
  k.CD0 = p.CD0, k.CD_a = p.CD_alpha, k.CD_qc = (float)(p.CD_q * c), k.CD_de = p.CD_del_e;
  k.CY0 = p.CY0, k.CY_b = p.CY_beta, k.CY_pb = (float)(p.CY_p * b);
  k.CY_rb = (float)(p.CY_r * b), k.CY_da = p.CY_del_a, k.CY_dr = p.CY_del_r;
  k.Cl0 = p.Cl0, k.Cl_b = p.Cl_beta, k.Cl_pb = (float)(p.Cl_p * b);
  k.Cl_rb = (float)(p.Cl_r * b), k.Cl_da = p.Cl_del_a, k.Cl_dr = p.Cl_del_r;
  k.Cm0 = p.Cm0, k.Cm_a = p.Cm_alpha, k.Cm_qc = (float)(p.Cm_q * c), k.Cm_de = p.Cm_del_e;
  k.Cn0 = p.Cn0, k.Cn_b = p.Cn_beta, k.Cn_pb = (float)(p.Cn_p * b);
  k.Cn_rb = (float)(p.Cn_r * b), k.Cn_da = p.Cn_del_a, k.Cn_dr = p.Cn_del_r;
  k.Ixx = p.I_xx, k.Iyy = p.I_yy, k.Izz = p.I_zz, k.a13 = -p.I_xz;
  const double det = (double)p.I_xx * p.I_zz - (double)p.I_xz * p.I_xz;
  k.i00 = (float)(p.I_zz / det), k.i22 = (float)(p.I_xx / det);
  k.i02 = (float)((double)p.I_xz / det);  // -a13 / det
  k.i11 = (float)(1.0 / (double)p.I_yy);
  return k;
}

// The default parameter set (neural_control/dynamics/config_fixed_wing.json:
// 1-42, dt = 0.05 = delta_t_train of configs/wing_config.json) as COMPILE-TIME
// constants, folded exactly like make_const does.  A kernel instantiated on
// this type gets every coefficient as an instruction literal instead of an
// SGPR operand: on gfx950 two waves of a SIMD issue VALU ops in each other's
// gaps only when all operands are VGPRs or literals
// (profiles/r02_issue_probe_coissue.jsonl).  The host selects it only when
// make_const(params, dt) reproduces this table bit for bit (is_default).
struct WingDefaultK {
  static constexpr float dt = 0.05f;
  static constexpr float half_rho = (float)(0.5 * (double)1.225f);
  static constexpr float S = 0.276f, c = 0.185f;
  static constexpr float inv_mass = (float)(1.0 / (double)1.01f);
  static constexpr float g_m = (float)((double)9.81f * (double)1.01f);
  // cosf / sinf(0.16534698176788384f)
  static constexpr float cos_eps = 0x1.f9045ap-1f, sin_eps = 0x1.5116f8p-3f;
  static constexpr float alpha_bound = (float)(10.0 / 180.0 * 3.14159265358979323846);
  static constexpr double cd = 0.185f, bd = 1.54f;
  static constexpr float CL0 = 0.39f, CL_a = 4.5321f, CL_qc = (float)(0.318f * cd),
                         CL_de = 0.527f;
  static constexpr float CD0 = 0.0765f, CD_a = 0.3346f, CD_qc = (float)(0.354f * cd),
                         CD_de = 0.004f;
  static constexpr float CY0 = 0.0f, CY_b = -0.033f, CY_pb = (float)(-0.1f * bd),
                         CY_rb = (float)(0.039f * bd), CY_da = 0.0f, CY_dr = 0.225f;
  static constexpr float Cl0 = 0.0f, Cl_b = -0.081f, Cl_pb = (float)(-0.529f * bd),
                         Cl_rb = (float)(0.159f * bd), Cl_da = -0.453f, Cl_dr = 0.005f;
  static constexpr float Cm0 = 0.02f, Cm_a = -1.4037f, Cm_qc = (float)(-0.1324f * cd),
                         Cm_de = -0.4236f;
  static constexpr float Cn0 = 0.0f, Cn_b = 0.189f, Cn_pb = (float)(-0.083f * bd),
                         Cn_rb = (float)(-0.948f * bd), Cn_da = -0.041f, Cn_dr = -0.077f;
  static constexpr float Ixx = 0.04766f, Iyy = 0.05005f, Izz = 0.09558f,
                         a13 = -(-0.00105f);
  static constexpr double det = (double)0.04766f * 0.09558f - (double)-0.00105f * -0.00105f;
  static constexpr float i00 = (float)(0.09558f / det), i22 = (float)(0.04766f / det);
  static constexpr float i02 = (float)((double)-0.00105f / det);
  static constexpr float i11 = (float)(1.0 / (double)0.05005f);
};

// the same numbers as a WingConst (host side of the dispatch)
inline WingConst default_table() {
  typedef WingDefaultK D;
  WingConst k;
  k.dt = D::dt, k.half_rho = D::half_rho, k.S = D::S, k.c = D::c;
  k.inv_mass = D::inv_mass, k.g_m = D::g_m, k.cos_eps = D::cos_eps;
  k.sin_eps = D::sin_eps, k.alpha_bound = D::alpha_bound;
  k.CL0 = D::CL0, k.CL_a = D::CL_a, k.CL_qc = D::CL_qc, k.CL_de = D::CL_de;
  k.CD0 = D::CD0, k.CD_a = D::CD_a, k.CD_qc = D::CD_qc, k.CD_de = D::CD_de;
  k.CY0 = D::CY0, k.CY_b = D::CY_b, k.CY_pb = D::CY_pb, k.CY_rb = D::CY_rb;
  k.CY_da = D::CY_da, k.CY_dr = D::CY_dr;
  k.Cl0 = D::Cl0, k.Cl_b = D::Cl_b, k.Cl_pb = D::Cl_pb, k.Cl_rb = D::Cl_rb;
  k.Cl_da = D::Cl_da, k.Cl_dr = D::Cl_dr;
  k.Cm0 = D::Cm0, k.Cm_a = D::Cm_a, k.Cm_qc = D::Cm_qc, k.Cm_de = D::Cm_de;
  k.Cn0 = D::Cn0, k.Cn_b = D::Cn_b, k.Cn_pb = D::Cn_pb, k.Cn_rb = D::Cn_rb;
  k.Cn_da = D::Cn_da, k.Cn_dr = D::Cn_dr;
  k.Ixx = D::Ixx, k.Iyy = D::Iyy, k.Izz = D::Izz, k.a13 = D::a13;
  k.i00 = D::i00, k.i02 = D::i02, k.i11 = D::i11, k.i22 = D::i22;
  return k;
}

// bit-for-bit: the literal kernel may stand in for the table kernel
inline bool is_default(const WingConst &k) {
  const WingConst d = default_table();
  return memcmp(&k, &d, sizeof(WingConst)) == 0;
}

// LearntFixedWingDynamics (fixed_wing_dynamics.py:270-326) keeps the WHOLE 3x3
// inertia matrix as one trainable parameter: after an optimizer step it is
// neither symmetric nor sparse.  The table of its step therefore carries the
// matrix and its inverse in full (and b, which the parameter cotangents need).
struct WingGeneralConst : WingConst {
  float b;
  float I[3][3], Iinv[3][3];
};

inline WingGeneralConst make_general_const(const ApgWingParams &p, float dt,
                                           const float *I9) {
  WingGeneralConst k;
  static_cast<WingConst &>(k) = make_const(p, dt);
  k.b = p.b;
  double m[3][3], inv[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = I9[i * 3 + j];
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
                     m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                     m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int a = (j + 1) % 3, b2 = (j + 2) % 3, c = (i + 1) % 3, d = (i + 2) % 3;
      inv[i][j] = (m[a][c] * m[b2][d] - m[a][d] * m[b2][c]) / det;  // adjugate
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) k.I[i][j] = (float)m[i][j], k.Iinv[i][j] = (float)inv[i][j];
  return k;
}

template <typename K>
struct wing_general_inertia
    : std::is_same<typename std::remove_cv<K>::type, WingGeneralConst> {};

// Cotangents of the physical parameters, in the order of ApgWingParams (41
// floats; the I_* slots stay 0) followed by dL/dI row-major (9):
// apg_wing_learnt_step_bwd's grad_params.
constexpr int kWingParamGrads = 50;
struct WingParamGrads {
  static constexpr bool enabled = true;
  float v[kWingParamGrads];
};
struct NoWingParamGrads {
  static constexpr bool enabled = false;
};

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTanBound = 0.17632698070846498f;  // tan(10 deg)

// ---- one trajectory per lane (float) or TWO (fx2) ---------------------------
// The per-trajectory arithmetic below is written once on a value type T:
// float, or fx2 = two trajectories per lane.  With fx2 every multiply / add /
// fma is ONE v_pk_*_f32 for both trajectories (5.1 cycles of issue for a lone
// wave against 2 x 4.1-5.1 for two scalar ops, profiles/r03_issue_probe2.jsonl)
// and the quarter-rate ops (sin, cos, rcp, sqrt) are issued per component by
// a wave that has its SIMD to itself (8 cycles each) - in the two-waves-per-
// SIMD regime of the scalar kernel a transcendental costs each wave ~23
// cycles (same probe), which is what held that kernel at 1.2 x one wave.
__host__ __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
__host__ __device__ __forceinline__ float min_(float a, float b) { return fminf(a, b); }
__host__ __device__ __forceinline__ float max_(float a, float b) { return fmaxf(a, b); }
__host__ __device__ __forceinline__ float abs_(float a) { return fabsf(a); }
__host__ __device__ __forceinline__ fx2 fma_(fx2 a, fx2 b, fx2 c) { return __builtin_elementwise_fma(a, b, c); }
__host__ __device__ __forceinline__ fx2 min_(fx2 a, fx2 b) { return (fx2){fminf(a.x, b.x), fminf(a.y, b.y)}; }
__host__ __device__ __forceinline__ fx2 max_(fx2 a, fx2 b) { return (fx2){fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
__host__ __device__ __forceinline__ fx2 abs_(fx2 a) { return (fx2){fabsf(a.x), fabsf(a.y)}; }
__host__ __device__ __forceinline__ fx2 splat(fx2, float v) { return (fx2){v, v}; }
__host__ __device__ __forceinline__ float splat(float, float v) { return v; }
#define APG_T(v) splat(T{}, (v))   // a constant of the value type

// alpha = clamp(atan(t), +-10 deg) needs the arc tangent on |t| <= tan(10 deg)
// only: outside, the clamp returns the bound and kills the gradient.  With t
// clamped FIRST, atan is its Taylor polynomial to t^9 (next term 4.6e-10) -
// 6 instructions instead of libm's ~40 with their selects - and
// atan(tan(bound)) reproduces the bound to 3e-9.  `free` = 1 where the clamp
// was inactive, computed arithmetically (a compare + v_cndmask pair costs ~5
// issue slots on gfx950, tools/issue_probe.hip).
template <typename T>
__host__ __device__ __forceinline__ T atan_clamped(T t, T *tc_out, T *free_out) {
  const T tc = min_(max_(t, APG_T(-kTanBound)), APG_T(kTanBound));
  *tc_out = tc;
  // (|t| - bound, not t - tc: the compiler contracts `w * (1/u) - tc` into an
  // fma whose exact product differs from the rounded t by its rounding error)
  *free_out = 1.f - min_(APG_T(1.f),
                         max_(abs_(t) - kTanBound, APG_T(0.f)) * 1e30f);
  const T z = tc * tc;
  T p = fma_(z, APG_T(1.f / 9.f), APG_T(-1.f / 7.f));
  p = fma_(z, p, APG_T(1.f / 5.f));
  p = fma_(z, p, APG_T(-1.f / 3.f));
  return fma_(tc * z, p, tc);
}

// sin / cos on |x| <= 10 deg (the clamped alpha, beta): Taylor to x^7 / x^6,
// truncation < 3e-11
template <typename T>
__host__ __device__ __forceinline__ void sincos_small(T x, T *sn, T *cs) {
  const T z = x * x;
  T ps = fma_(z, APG_T(-1.f / 5040.f), APG_T(1.f / 120.f));
  ps = fma_(z, ps, APG_T(-1.f / 6.f));
  *sn = fma_(x * z, ps, x);
  T pc = fma_(z, APG_T(-1.f / 720.f), APG_T(1.f / 24.f));
  pc = fma_(z, pc, APG_T(-0.5f));
  *cs = fma_(z, pc, APG_T(1.f));
}

// attitude angles: the hardware pair on the device (apg_device.h), the
// branch-free software pair on the host (tests/host_math)
__host__ __device__ __forceinline__ void sincos_att(float x, float *sn, float *cs) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(APG_SW_TRIG)
  sincos_hw(x, sn, cs);
#else
  sincos_fast(x, sn, cs);
#endif
}
__host__ __device__ __forceinline__ void sincos_att(fx2 x, fx2 *sn, fx2 *cs) {
  float s0, c0, s1, c1;
  sincos_att(x.x, &s0, &c0);
  sincos_att(x.y, &s1, &c1);
  *sn = (fx2){s0, s1}, *cs = (fx2){c0, c1};
}

// Everything the adjoint re-uses from the forward evaluation of one step.
template <typename T>
struct WingAuxT {
  T Tt, de, da, dr;
  T V, V2, iV, r2V, tw, tb;      // iV = 1/V; tw = w/u, tb = v/V, both CLAMPED
                                 // to +-tan(10 deg)
  T alpha, beta;
  T alpha_free, beta_free;       // 1: clamp passes the gradient, 0: it does not
  T sa, ca, sb, cb;
  T CL, CD, CY, Cl, Cm, Cn, Q;
  T L, D, Y;
  T sph, cph, sth, cth, sps, cps, icth;
  T R[3][3];                     // rows as assembled at :80-91
  T h0, h1, h2;                  // I * omega
};
typedef WingAuxT<float> WingAux;

// Evaluates state_dot (12) and fills aux.
// (KT = const WingConst, possibly qualified with the constant address space,
// or WingDefaultK)
template <typename T, typename KT>
__host__ __device__ __forceinline__ void wing_rates(const T (&s)[12], const T (&a)[4],
                                           KT &k, WingAuxT<T> &x, T (&sd)[12]) {
  const T u = s[3], v = s[4], w = s[5];
  const T p = s[9], q = s[10], r = s[11];
  // normalize_action :41-46 (pi (40 a - 20) / 180 with the constants folded:
  // no IEEE division sequence per control surface)
  x.Tt = a[0] * 7.f;
  x.de = fma_(a[1], APG_T(kPi * 40.f / 180.f), APG_T(-kPi * 20.f / 180.f));
  x.da = fma_(a[2], APG_T(kPi * 5.f / 180.f), APG_T(-kPi * 2.5f / 180.f));
  x.dr = fma_(a[3], APG_T(kPi * 40.f / 180.f), APG_T(-kPi * 20.f / 180.f));
  // :130-134
  x.V2 = u * u + v * v + w * w;
  x.V = sqrt_fast(x.V2);
  x.iV = rcp_nr(x.V);
  // (1 / u through rcp_nr_finite: u = 0 gives t = +-inf -> alpha = +-bound
  // with a dead gradient like the reference's clamp(atan(w / u)), not NaN)
  x.alpha = atan_clamped(w * rcp_nr_finite(u), &x.tw, &x.alpha_free);
  x.beta = atan_clamped(v * x.iV, &x.tb, &x.beta_free);
  x.r2V = 0.5f * x.iV;
  // :139-164
  x.CL = k.CL0 + k.CL_a * x.alpha + k.CL_qc * x.r2V * q + k.CL_de * x.de;
  x.CD = k.CD0 + k.CD_a * x.alpha + k.CD_qc * x.r2V * q + k.CD_de * x.de;
  x.CY = k.CY0 + k.CY_b * x.beta + k.CY_pb * x.r2V * p + k.CY_rb * x.r2V * r +
         k.CY_da * x.da + k.CY_dr * x.dr;
  x.Cl = k.Cl0 + k.Cl_b * x.beta + k.Cl_pb * x.r2V * p + k.Cl_rb * x.r2V * r +
         k.Cl_da * x.da + k.Cl_dr * x.dr;
  x.Cm = k.Cm0 + k.Cm_a * x.alpha + k.Cm_qc * x.r2V * q + k.Cm_de * x.de;
  x.Cn = k.Cn0 + k.Cn_b * x.beta + k.Cn_pb * x.r2V * p + k.Cn_rb * x.r2V * r +
         k.Cn_da * x.da + k.Cn_dr * x.dr;
  // :167-175
  x.Q = k.half_rho * x.V2 * k.S;
  x.L = x.Q * x.CL, x.D = x.Q * x.CD, x.Y = x.Q * x.CY;
  const T l = x.Q * k.c * x.Cl, m = x.Q * k.c * x.Cm, n = x.Q * k.c * x.Cn;
  // :185-204 body forces
  sincos_small(x.alpha, &x.sa, &x.ca);
  sincos_small(x.beta, &x.sb, &x.cb);
  sincos_att(s[6], &x.sph, &x.cph);
  sincos_att(s[7], &x.sth, &x.cth);
  sincos_att(s[8], &x.sps, &x.cps);
  x.icth = rcp_nr(x.cth);
  const T f0 = -x.ca * x.cb * x.D - x.ca * x.sb * x.Y + x.sa * x.L -
               k.g_m * x.sth + x.Tt * k.cos_eps;
  const T f1 = -x.sb * x.D + x.cb * x.Y + k.g_m * x.sph * x.cth;
  const T f2_ = -x.sa * x.cb * x.D - x.sa * x.sb * x.Y - x.ca * x.L +
                k.g_m * x.cph * x.cth + x.Tt * k.sin_eps;
  // :213-216 position rate: R^T vel with R rows as at :80-91
  x.R[0][0] = x.cth * x.cps, x.R[0][1] = x.cth * x.sps, x.R[0][2] = -x.sth;
  x.R[1][0] = -x.cph * x.sps + x.sph * x.sth * x.cps;
  x.R[1][1] = x.cph * x.cps + x.sph * x.sth * x.sps;
  x.R[1][2] = x.sph * x.cth;
  x.R[2][0] = x.sph * x.sps + x.cph * x.sth * x.cps;
  x.R[2][1] = -x.sph * x.cps + x.cph * x.sth * x.sps;
  x.R[2][2] = x.cph * x.cth;
#pragma unroll
  for (int j = 0; j < 3; ++j)
    sd[j] = x.R[0][j] * u + x.R[1][j] * v + x.R[2][j] * w;
  // :220-221
  sd[3] = k.inv_mass * f0 - (q * w - r * v);
  sd[4] = k.inv_mass * f1 - (r * u - p * w);
  sd[5] = k.inv_mass * f2_ - (p * v - q * u);
  // :225-245
  const T tth = x.sth * x.icth;
  sd[6] = p + x.sph * tth * q + x.cph * tth * r;
  sd[7] = x.cph * q - x.sph * r;
  sd[8] = (x.sph * q + x.cph * r) * x.icth;
  // :250-255
  if constexpr (wing_general_inertia<KT>::value) {
    x.h0 = k.I[0][0] * p + k.I[0][1] * q + k.I[0][2] * r;
    x.h1 = k.I[1][0] * p + k.I[1][1] * q + k.I[1][2] * r;
    x.h2 = k.I[2][0] * p + k.I[2][1] * q + k.I[2][2] * r;
  } else {
    x.h0 = k.Ixx * p + k.a13 * r, x.h1 = k.Iyy * q, x.h2 = k.a13 * p + k.Izz * r;
  }
  const T r0 = l - (q * x.h2 - r * x.h1);
  const T r1 = m - (r * x.h0 - p * x.h2);
  const T r2 = n - (p * x.h1 - q * x.h0);
  if constexpr (wing_general_inertia<KT>::value) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      sd[9 + i] = k.Iinv[i][0] * r0 + k.Iinv[i][1] * r1 + k.Iinv[i][2] * r2;
  } else {
    sd[9] = k.i00 * r0 + k.i02 * r2;
    sd[10] = k.i11 * r1;
    sd[11] = k.i02 * r0 + k.i22 * r2;
  }
}

template <typename T, typename KT>
__host__ __device__ __forceinline__ void wing_step(T (&s)[12], const T (&a)[4], KT &k) {
  WingAuxT<T> x;
  T sd[12];
  wing_rates(s, a, k, x, sd);
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = s[i] + k.dt * sd[i];
}

// lam: dL/dnext on entry -> dL/dstate on exit; ga += dL/daction.
// `s`, `a` are the PRE-step state and the action; x the matching aux.
// `sd`: the state_dot wing_rates returned for (s, a) (its position part is
// re-used by the yaw cotangent).
// `pg` (WingParamGrads, float only): += the cotangents of the physical
// parameters for this trajectory; NoWingParamGrads compiles to nothing.
template <typename T, typename KT, typename PG>
__host__ __device__ __forceinline__ void wing_step_adjoint(T (&lam)[12], T (&ga)[4],
                                                  const T (&s)[12],
                                                  const WingAuxT<T> &x,
                                                  const T (&sd)[12], KT &k,
                                                  PG &pg) {
  const T u = s[3], v = s[4], w = s[5];
  const T p = s[9], q = s[10], r = s[11];
  const T zero = APG_T(0.f);
  T g[12];  // cotangent of state_dot
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = k.dt * lam[i];
  T du = zero, dv = zero, dw = zero, dph = zero, dth = zero, dps = zero;
  T dp = zero, dq = zero, dr = zero;

  // omega_dot = I^-1 (M - omega x I omega)
  T gr0, gr1, gr2;  // I^-T g_omega
  if constexpr (wing_general_inertia<KT>::value) {
    gr0 = k.Iinv[0][0] * g[9] + k.Iinv[1][0] * g[10] + k.Iinv[2][0] * g[11];
    gr1 = k.Iinv[0][1] * g[9] + k.Iinv[1][1] * g[10] + k.Iinv[2][1] * g[11];
    gr2 = k.Iinv[0][2] * g[9] + k.Iinv[1][2] * g[10] + k.Iinv[2][2] * g[11];
  } else {
    gr0 = k.i00 * g[9] + k.i02 * g[11];
    gr1 = k.i11 * g[10];
    gr2 = k.i02 * g[9] + k.i22 * g[11];
  }
  {
    const T c0 = -gr0, c1 = -gr1, c2 = -gr2;  // cotangent of the cross
    dp += -c1 * x.h2 + c2 * x.h1;
    dq += c0 * x.h2 - c2 * x.h0;
    dr += -c0 * x.h1 + c1 * x.h0;
    const T gh0 = c1 * r - c2 * q, gh1 = -c0 * r + c2 * p, gh2 = c0 * q - c1 * p;
    if constexpr (wing_general_inertia<KT>::value) {
      dp += k.I[0][0] * gh0 + k.I[1][0] * gh1 + k.I[2][0] * gh2;
      dq += k.I[0][1] * gh0 + k.I[1][1] * gh1 + k.I[2][1] * gh2;
      dr += k.I[0][2] * gh0 + k.I[1][2] * gh1 + k.I[2][2] * gh2;
    } else {
      dp += k.Ixx * gh0 + k.a13 * gh2;
      dq += k.Iyy * gh1;
      dr += k.a13 * gh0 + k.Izz * gh2;
    }
  }
  // euler rates
  {
    const T icth = x.icth, tth = x.sth * icth;
    const T sq_cr = x.sph * q + x.cph * r;   // sin(phi) q + cos(phi) r
    const T cq_sr = x.cph * q - x.sph * r;
    dp += g[6];
    dq += g[6] * x.sph * tth + g[7] * x.cph + g[8] * x.sph * icth;
    dr += g[6] * x.cph * tth - g[7] * x.sph + g[8] * x.cph * icth;
    dph += g[6] * tth * cq_sr - g[7] * sq_cr + g[8] * cq_sr * icth;
    dth += g[6] * sq_cr * icth * icth + g[8] * sq_cr * x.sth * icth * icth;
  }
  // uvw_dot = f/m - omega x vel
  const T gf0 = k.inv_mass * g[3], gf1 = k.inv_mass * g[4], gf2 = k.inv_mass * g[5];
  {
    const T x0 = -g[3], x1 = -g[4], x2 = -g[5];
    dq += x0 * w, dw += x0 * q, dr -= x0 * v, dv -= x0 * r;
    dr += x1 * u, du += x1 * r, dp -= x1 * w, dw -= x1 * p;
    dp += x2 * v, dv += x2 * p, dq -= x2 * u, du -= x2 * q;
  }
  // pos_dot_j = sum_i R[i][j] vel_i.  With Rg_i = R[i] . g (needed for the
  // velocity cotangent anyway) the attitude cotangents contract to a few
  // products, because every derivative of a row of R is another row:
  //   dR1/dphi = R2, dR2/dphi = -R1          => dphi   = v Rg_2 - w Rg_1
  //   dR1/dth = sph R0, dR2/dth = cph R0     => dtheta = u (g . dR0/dth)
  //                                                      + (v sph + w cph) Rg_0
  //   dcol0/dpsi = -col1, dcol1/dpsi = col0  => dpsi   = g1 pd0 - g0 pd1
  {
    T Rg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      Rg[i] = x.R[i][0] * g[0] + x.R[i][1] * g[1] + x.R[i][2] * g[2];
    du += Rg[0], dv += Rg[1], dw += Rg[2];
    dph += v * Rg[2] - w * Rg[1];
    const T g_dr0 = -x.sth * (x.cps * g[0] + x.sps * g[1]) - x.cth * g[2];
    dth += u * g_dr0 + (v * x.sph + w * x.cph) * Rg[0];
    dps += g[1] * sd[0] - g[0] * sd[1];
  }
  // f = R_bw [-D, Y, -L] + gravity(phi, theta) + thrust
  const T gT = gf0 * k.cos_eps + gf2 * k.sin_eps;
  dth += k.g_m * (-x.cth * gf0 - x.sph * x.sth * gf1 - x.cph * x.sth * gf2);
  dph += k.g_m * (x.cph * x.cth * gf1 - x.sph * x.cth * gf2);
  // f_aero = R_bw(alpha, beta) [-D, Y, -L], contracted through
  //   m = ca gf0 + sa gf2,  t = cb D + sb Y,  q = sb D - cb Y:
  //   d f_aero0 / dalpha = -f_aero2,  d f_aero2 / dalpha = f_aero0
  const T m_ = x.ca * gf0 + x.sa * gf2;
  const T t_ = x.cb * x.D + x.sb * x.Y;
  const T q_ = x.sb * x.D - x.cb * x.Y;
  const T gD = -x.cb * m_ - x.sb * gf1;
  const T gY = -x.sb * m_ + x.cb * gf1;
  const T gL = x.sa * gf0 - x.ca * gf2;
  const T fa0 = x.sa * x.L - x.ca * t_, fa2 = -x.sa * t_ - x.ca * x.L;
  T g_al = gf2 * fa0 - gf0 * fa2;
  T g_be = q_ * m_ - gf1 * t_;
  // forces and moments
  const T Qc = x.Q * k.c;
  const T gCL = x.Q * gL, gCD = x.Q * gD, gCY = x.Q * gY;
  const T gCl = Qc * gr0, gCm = Qc * gr1, gCn = Qc * gr2;
  const T gQ = gL * x.CL + gD * x.CD + gY * x.CY +
               k.c * (gr0 * x.Cl + gr1 * x.Cm + gr2 * x.Cn);
  T gV2 = k.half_rho * k.S * gQ;
  // coefficients
  g_al += k.CL_a * gCL + k.CD_a * gCD + k.Cm_a * gCm;
  g_be += k.CY_b * gCY + k.Cl_b * gCl + k.Cn_b * gCn;
  const T g_qt = k.CL_qc * gCL + k.CD_qc * gCD + k.Cm_qc * gCm;
  const T g_pt = k.CY_pb * gCY + k.Cl_pb * gCl + k.Cn_pb * gCn;
  const T g_rt = k.CY_rb * gCY + k.Cl_rb * gCl + k.Cn_rb * gCn;
  const T g_de = k.CL_de * gCL + k.CD_de * gCD + k.Cm_de * gCm;
  const T g_da = k.CY_da * gCY + k.Cl_da * gCl + k.Cn_da * gCn;
  const T g_dr = k.CY_dr * gCY + k.Cl_dr * gCl + k.Cn_dr * gCn;
  dq += g_qt * x.r2V, dp += g_pt * x.r2V, dr += g_rt * x.r2V;
  const T g_r2V = g_qt * q + g_pt * p + g_rt * r;
  const T iV = x.iV;
  T gV = -g_r2V * x.r2V * iV;
  // alpha = clamp(atan(w/u)), beta = clamp(atan(v/V)): where the clamp is
  // active the mask is 0 and tw / tb hold the (finite) bound
  {
    const T gt = x.alpha_free * g_al * rcp_nr(1.f + x.tw * x.tw);
    const T iu = rcp_nr_finite(u);   // gt = 0 where the clamp is active:
                                     // 0 * FLT_MAX = 0, never 0 * inf
    dw += gt * iu;
    du -= gt * x.tw * iu;
  }
  {
    const T gt = x.beta_free * g_be * rcp_nr(1.f + x.tb * x.tb);
    dv += gt * iV;
    gV -= gt * x.tb * iV;
  }
  gV2 += gV * 0.5f * iV;
  du += 2.f * u * gV2, dv += 2.f * v * gV2, dw += 2.f * w * gV2;
  // actions
  ga[0] += 7.f * gT;
  ga[1] += g_de * (kPi * 40.f / 180.f);   // (constant-folded)
  ga[2] += g_da * (kPi * 5.f / 180.f);
  ga[3] += g_dr * (kPi * 40.f / 180.f);
  lam[3] += du, lam[4] += dv, lam[5] += dw;
  lam[6] += dph, lam[7] += dth, lam[8] += dps;
  lam[9] += dp, lam[10] += dq, lam[11] += dr;
  if constexpr (PG::enabled) {
    // the coefficient sums are linear in their coefficients; the rate terms
    // carry c or b (k.CL_qc = CL_q c, ...), Q = rho/2 V^2 S, the moments Q c C
    float *d = pg.v;
    const T rq = x.r2V * q, rp = x.r2V * p, rr = x.r2V * r;
    // mass enters through 1/mass only: g m of the weight is a detached copy
    // in the reference (torch.tensor(g_m), :197) - so does g get no gradient
    d[0] -= k.inv_mass * (g[3] * (sd[3] + (q * w - r * v)) +
                          g[4] * (sd[4] + (r * u - p * w)) +
                          g[5] * (sd[5] + (p * v - q * u)));
    d[5] += gQ * 0.5f * x.V2 * k.S;                       // rho
    d[6] += gQ * k.half_rho * x.V2;                       // S
    d[7] += g_qt * rq / k.c + x.Q * (gr0 * x.Cl + gr1 * x.Cm + gr2 * x.Cn);  // c
    d[10] += gCL, d[11] += gCL * x.alpha, d[12] += gCL * rq * k.c, d[13] += gCL * x.de;
    d[14] += gCD, d[15] += gCD * x.alpha, d[16] += gCD * rq * k.c, d[17] += gCD * x.de;
    d[30] += gCm, d[31] += gCm * x.alpha, d[32] += gCm * rq * k.c, d[33] += gCm * x.de;
    d[40] += x.Tt * (gf2 * k.cos_eps - gf0 * k.sin_eps);  // epsilon
    if constexpr (wing_general_inertia<KT>::value) {
      d[8] += (g_pt * p + g_rt * r) * x.r2V / k.b;         // b
      const T cg[3] = {gCY, gCl, gCn};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float *e = d + (t == 0 ? 18 : t == 1 ? 24 : 34);
        e[0] += cg[t], e[1] += cg[t] * x.beta, e[2] += cg[t] * rp * k.b;
        e[3] += cg[t] * rr * k.b, e[4] += cg[t] * x.da, e[5] += cg[t] * x.dr;
      }
      // omega_dot = I^-1 (M - omega x I omega):
      //   dL/dI_ij = -gr_i omega_dot_j - (gr x omega)_i omega_j
      const T grv[3] = {gr0, gr1, gr2}, om[3] = {p, q, r};
      const T gxo[3] = {gr1 * r - gr2 * q, gr2 * p - gr0 * r, gr0 * q - gr1 * p};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          d[41 + i * 3 + j] -= grv[i] * sd[9 + j] + gxo[i] * om[j];
    }
  }
}

template <typename T, typename KT>
__host__ __device__ __forceinline__ void wing_step_adjoint(T (&lam)[12], T (&ga)[4],
                                                  const T (&s)[12],
                                                  const WingAuxT<T> &x,
                                                  const T (&sd)[12], KT &k) {
  NoWingParamGrads none;
  wing_step_adjoint(lam, ga, s, x, sd, k, none);
}

}  // namespace
}  // namespace apg
