/*
 * ORACLE - test infrastructure, never the product path.
 *
 * Plain-C restatement of the reference's fixed-wing and cart-pole APG paths
 * in the reference's own MATRIX form (rotation matrices assembled entry by
 * entry, matrix-vector products, cross products) with a hand-written reverse
 * sweep over that same op sequence - an independent derivation from the
 * scalarised adjoint the HIP kernels execute (csrc/wing_math.h,
 * csrc/cartpole_math.h):
 *   FixedWingDynamics.simulate_fixed_wing
 *       neural_control/dynamics/fixed_wing_dynamics.py:95-267
 *     normalize_action :41-46, body_wind_function :48-63,
 *     inertial_body_function :65-93, config_fixed_wing.json
 *   fixed_wing_mpc_loss           neural_control/drone_loss.py:72-82
 *   the k-step unroll             scripts/train_fixed_wing.py:90-116
 *   CartpoleDynamics.simulate_cartpole
 *       neural_control/dynamics/cartpole_dynamics.py:53-119 (friction 0.5 :34)
 *   cartpole_loss_mpc             neural_control/drone_loss.py:136-145
 *   make_reference + unroll       scripts/train_cartpole.py:103-150
 * Compiled twice (REAL = float / double) into liboracle.so; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may load it.  Pinned
 * against tests/golden/wing.npz and cartpole.npz by tests/test_oracle_c.py,
 * and against central finite differences in fp64.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
#ifndef SUFFIX
#define SUFFIX f32
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* ------------------------------------------------------------------ wing */
typedef struct OracleWingCfg { /* config_fixed_wing.json after modified_params */
  double mass, I_xx, I_yy, I_zz, I_xz, rho, S, c, b, g;
  double CL0, CL_alpha, CL_q, CL_del_e;
  double CD0, CD_alpha, CD_q, CD_del_e;
  double CY0, CY_beta, CY_p, CY_r, CY_del_a, CY_del_r;
  double Cl0, Cl_beta, Cl_p, Cl_r, Cl_del_a, Cl_del_r;
  double Cm0, Cm_alpha, Cm_q, Cm_del_e;
  double Cn0, Cn_beta, Cn_p, Cn_r, Cn_del_a, Cn_del_r;
  double epsilon;
} OracleWingCfg;

#define PI_D 3.14159265358979323846

static void matvec(const REAL M[3][3], const REAL v[3], REAL o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = M[i][0] * v[0] + M[i][1] * v[1] + M[i][2] * v[2];
}
/* o = M v  =>  gM += go v^T, gv += M^T go */
static void matvec_vjp(const REAL M[3][3], const REAL v[3], const REAL go[3],
                       REAL gM[3][3], REAL gv[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      gM[i][j] += go[i] * v[j];
      gv[j] += M[i][j] * go[i];
    }
}
static void cross(const REAL a[3], const REAL b[3], REAL o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
/* o = a x b  =>  ga += b x go, gb += go x a */
static void cross_vjp(const REAL a[3], const REAL b[3], const REAL go[3],
                      REAL ga[3], REAL gb[3]) {
  REAL t[3];
  cross(b, go, t);
  for (int i = 0; i < 3; ++i) ga[i] += t[i];
  cross(go, a, t);
  for (int i = 0; i < 3; ++i) gb[i] += t[i];
}

/* rows m1, m2, m3 of inertial_body_function BEFORE its final transpose (:78-91) */
static void rows_ib(REAL ph, REAL th, REAL ps, REAL M[3][3]) {
  REAL sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps),
       cps = cos(ps);
  M[0][0] = cth * cps, M[0][1] = cth * sps, M[0][2] = -sth;
  M[1][0] = -cph * sps + sph * sth * cps, M[1][1] = cph * cps + sph * sth * sps;
  M[1][2] = sph * cth;
  M[2][0] = sph * sps + cph * sth * cps, M[2][1] = -sph * cps + cph * sth * sps;
  M[2][2] = cph * cth;
}
/* cotangent of every entry -> cotangent of (phi, theta, psi), via the six
 * trig values */
static void rows_ib_vjp(REAL ph, REAL th, REAL ps, const REAL gM[3][3], REAL *gph,
                        REAL *gth, REAL *gps) {
  REAL sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps),
       cps = cos(ps);
  REAL gsph = 0, gcph = 0, gsth = 0, gcth = 0, gsps = 0, gcps = 0;
  /* M00 = cth*cps */ gcth += gM[0][0] * cps; gcps += gM[0][0] * cth;
  /* M01 = cth*sps */ gcth += gM[0][1] * sps; gsps += gM[0][1] * cth;
  /* M02 = -sth    */ gsth -= gM[0][2];
  /* M10 = -cph*sps + sph*sth*cps */
  gcph -= gM[1][0] * sps; gsps -= gM[1][0] * cph;
  gsph += gM[1][0] * sth * cps; gsth += gM[1][0] * sph * cps; gcps += gM[1][0] * sph * sth;
  /* M11 = cph*cps + sph*sth*sps */
  gcph += gM[1][1] * cps; gcps += gM[1][1] * cph;
  gsph += gM[1][1] * sth * sps; gsth += gM[1][1] * sph * sps; gsps += gM[1][1] * sph * sth;
  /* M12 = sph*cth */ gsph += gM[1][2] * cth; gcth += gM[1][2] * sph;
  /* M20 = sph*sps + cph*sth*cps */
  gsph += gM[2][0] * sps; gsps += gM[2][0] * sph;
  gcph += gM[2][0] * sth * cps; gsth += gM[2][0] * cph * cps; gcps += gM[2][0] * cph * sth;
  /* M21 = -sph*cps + cph*sth*sps */
  gsph -= gM[2][1] * cps; gcps -= gM[2][1] * sph;
  gcph += gM[2][1] * sth * sps; gsth += gM[2][1] * cph * sps; gsps += gM[2][1] * cph * sth;
  /* M22 = cph*cth */ gcph += gM[2][2] * cth; gcth += gM[2][2] * cph;
  *gph += gsph * cph - gcph * sph;
  *gth += gsth * cth - gcth * sth;
  *gps += gsps * cps - gcps * sps;
}

typedef struct {
  REAL mass, g_m, half_rho, S, c, ceps, seps, ab;
  REAL CL0, CLa, CLqc, CLde, CD0, CDa, CDqc, CDde;
  REAL CY0, CYb, CYpb, CYrb, CYda, CYdr, Cl0, Clb, Clpb, Clrb, Clda, Cldr;
  REAL Cm0, Cma, Cmqc, Cmde, Cn0, Cnb, Cnpb, Cnrb, Cnda, Cndr;
  REAL I[3][3], Iinv[3][3];
} WPar;

static void wing_par(const OracleWingCfg *k, WPar *p) {
  /* python-double arithmetic of the reference before it meets a tensor */
  p->mass = (REAL)k->mass;
  p->g_m = (REAL)(float)(k->g * k->mass);       /* torch.tensor(g_m): float32 */
  p->half_rho = (REAL)(0.5 * k->rho);
  p->S = (REAL)k->S, p->c = (REAL)k->c;
  p->ceps = (REAL)(float)cosf((float)k->epsilon);  /* torch.cos of a float32 tensor */
  p->seps = (REAL)(float)sinf((float)k->epsilon);
  if (sizeof(REAL) == 8) p->ceps = cos(k->epsilon), p->seps = sin(k->epsilon);
  p->ab = (REAL)(10.0 / 180.0 * PI_D);
  p->CL0 = k->CL0, p->CLa = k->CL_alpha, p->CLqc = (REAL)(k->CL_q * k->c), p->CLde = k->CL_del_e;
  p->CD0 = k->CD0, p->CDa = k->CD_alpha, p->CDqc = (REAL)(k->CD_q * k->c), p->CDde = k->CD_del_e;
  p->CY0 = k->CY0, p->CYb = k->CY_beta, p->CYpb = (REAL)(k->CY_p * k->b);
  p->CYrb = (REAL)(k->CY_r * k->b), p->CYda = k->CY_del_a, p->CYdr = k->CY_del_r;
  p->Cl0 = k->Cl0, p->Clb = k->Cl_beta, p->Clpb = (REAL)(k->Cl_p * k->b);
  p->Clrb = (REAL)(k->Cl_r * k->b), p->Clda = k->Cl_del_a, p->Cldr = k->Cl_del_r;
  p->Cm0 = k->Cm0, p->Cma = k->Cm_alpha, p->Cmqc = (REAL)(k->Cm_q * k->c), p->Cmde = k->Cm_del_e;
  p->Cn0 = k->Cn0, p->Cnb = k->Cn_beta, p->Cnpb = (REAL)(k->Cn_p * k->b);
  p->Cnrb = (REAL)(k->Cn_r * k->b), p->Cnda = k->Cn_del_a, p->Cndr = k->Cn_del_r;
  memset(p->I, 0, sizeof(p->I));
  p->I[0][0] = (REAL)(float)k->I_xx, p->I[0][2] = (REAL)(float)(-k->I_xz);
  p->I[1][1] = (REAL)(float)k->I_yy;
  p->I[2][0] = (REAL)(float)(-k->I_xz), p->I[2][2] = (REAL)(float)k->I_zz;
  /* torch.inverse(I): closed form of this sparsity pattern */
  const double a = p->I[0][0], b = p->I[0][2], d = p->I[2][2],
               det = a * d - b * b;
  memset(p->Iinv, 0, sizeof(p->Iinv));
  p->Iinv[0][0] = (REAL)(d / det), p->Iinv[0][2] = (REAL)(-b / det);
  p->Iinv[2][0] = (REAL)(-b / det), p->Iinv[2][2] = (REAL)(a / det);
  p->Iinv[1][1] = (REAL)(1.0 / p->I[1][1]);
}

/* one explicit-Euler step; if gn != NULL also the vector-Jacobian products */
static void wing_step(const WPar *p, const REAL *s, const REAL *a, REAL dt,
                      REAL *o, const REAL *gn, REAL *gs, REAL *ga) {
  const REAL *vel = s + 3, *om = s + 9;
  const REAL u = s[3], v = s[4], w = s[5], ph = s[6], th = s[7], ps = s[8];
  const REAL pi = (REAL)PI_D;
  /* normalize_action :41-46 */
  const REAL T = a[0] * 7;
  const REAL de = pi * (a[1] * 40 - 20) / 180, da = pi * (a[2] * 5 - (REAL)2.5) / 180,
             dr = pi * (a[3] * 40 - 20) / 180;
  /* :130-134 */
  const REAL V = sqrt(u * u + v * v + w * w);
  const REAL al0 = atan(w / u), be0 = atan(v / V);
  const int al_in = al0 >= -p->ab && al0 <= p->ab, be_in = be0 >= -p->ab && be0 <= p->ab;
  const REAL al = al0 < -p->ab ? -p->ab : (al0 > p->ab ? p->ab : al0);
  const REAL be = be0 < -p->ab ? -p->ab : (be0 > p->ab ? p->ab : be0);
  const REAL i2V = 1 / (2 * V);
  /* :139-164 */
  const REAL CL = p->CL0 + p->CLa * al + p->CLqc * i2V * om[1] + p->CLde * de;
  const REAL CD = p->CD0 + p->CDa * al + p->CDqc * i2V * om[1] + p->CDde * de;
  const REAL CY = p->CY0 + p->CYb * be + p->CYpb * i2V * om[0] + p->CYrb * i2V * om[2] +
                  p->CYda * da + p->CYdr * dr;
  const REAL Cl = p->Cl0 + p->Clb * be + p->Clpb * i2V * om[0] + p->Clrb * i2V * om[2] +
                  p->Clda * da + p->Cldr * dr;
  const REAL Cm = p->Cm0 + p->Cma * al + p->Cmqc * i2V * om[1] + p->Cmde * de;
  const REAL Cn = p->Cn0 + p->Cnb * be + p->Cnpb * i2V * om[0] + p->Cnrb * i2V * om[2] +
                  p->Cnda * da + p->Cndr * dr;
  /* :167-175 */
  const REAL qS = p->half_rho * (V * V) * p->S;
  const REAL L = qS * CL, D = qS * CD, Y = qS * CY;
  const REAL Mb[3] = {qS * p->c * Cl, qS * p->c * Cm, qS * p->c * Cn};
  /* :185-204 */
  const REAL sa = sin(al), ca = cos(al), sb = sin(be), cb = cos(be);
  const REAL Rbw[3][3] = {{ca * cb, -ca * sb, -sa}, {sb, cb, 0}, {sa * cb, -sa * sb, ca}};
  const REAL aero[3] = {-D, Y, -L};
  REAL fa[3], fg[3], M0[3][3], Mi[3][3];
  matvec(Rbw, aero, fa);
  rows_ib(ph, th, 0, M0);  /* transpose(inertial_body(phi, theta, 0)) = the rows */
  const REAL grav[3] = {0, 0, p->g_m};
  matvec(M0, grav, fg);
  const REAL f[3] = {fa[0] + fg[0] + T * p->ceps, fa[1] + fg[1], fa[2] + fg[2] + T * p->seps};
  /* :213-216 pos_dot = R_ib vel, R_ib = rows^T */
  rows_ib(ph, th, ps, Mi);
  REAL sd[12];
  for (int j = 0; j < 3; ++j) sd[j] = Mi[0][j] * vel[0] + Mi[1][j] * vel[1] + Mi[2][j] * vel[2];
  /* :220-221 */
  REAL oxv[3];
  cross(om, vel, oxv);
  for (int i = 0; i < 3; ++i) sd[3 + i] = (1 / p->mass) * f[i] - oxv[i];
  /* :225-245 */
  const REAL sph = sin(ph), cph = cos(ph), tth = tan(th), cth = cos(th);
  const REAL E[3][3] = {{1, sph * tth, cph * tth}, {0, cph, -sph}, {0, sph / cth, cph / cth}};
  matvec(E, om, sd + 6);
  /* :250-255 */
  REAL h[3], oxh[3], cp[3];
  matvec(p->I, om, h);
  cross(om, h, oxh);
  for (int i = 0; i < 3; ++i) cp[i] = Mb[i] - oxh[i];
  matvec(p->Iinv, cp, sd + 9);
  if (o)
    for (int i = 0; i < 12; ++i) o[i] = s[i] + dt * sd[i];
  if (!gn) return;

  /* ---------------------------------------------------- reverse sweep ---- */
  REAL g[12], gvel[3] = {0, 0, 0}, gom[3] = {0, 0, 0};
  REAL gph = 0, gth = 0, gps = 0;
  for (int i = 0; i < 12; ++i) g[i] = dt * gn[i];
  /* omega_dot = Iinv cp */
  REAL gcp[3] = {0, 0, 0}, gdummy[3][3];
  memset(gdummy, 0, sizeof(gdummy));
  matvec_vjp(p->Iinv, cp, g + 9, gdummy, gcp);
  /* cp = Mb - om x h */
  REAL gMb[3] = {gcp[0], gcp[1], gcp[2]}, gx[3] = {-gcp[0], -gcp[1], -gcp[2]}, gh[3] = {0, 0, 0};
  cross_vjp(om, h, gx, gom, gh);
  matvec_vjp(p->I, om, gh, gdummy, gom);
  /* eul_dot = E om */
  REAL gE[3][3];
  memset(gE, 0, sizeof(gE));
  matvec_vjp(E, om, g + 6, gE, gom);
  {
    const REAL gsph = gE[0][1] * tth - gE[1][2] + gE[2][1] / cth;
    const REAL gcph = gE[0][2] * tth + gE[1][1] + gE[2][2] / cth;
    const REAL gtth = gE[0][1] * sph + gE[0][2] * cph;
    const REAL gicth = gE[2][1] * sph + gE[2][2] * cph; /* cotangent of 1/cos(theta) */
    gph += gsph * cph - gcph * sph;
    gth += gtth / (cth * cth) + gicth * sin(th) / (cth * cth);
  }
  /* uvw_dot = f / m - om x vel */
  REAL gf[3], gx2[3];
  for (int i = 0; i < 3; ++i) gf[i] = g[3 + i] / p->mass, gx2[i] = -g[3 + i];
  cross_vjp(om, vel, gx2, gom, gvel);
  /* pos_dot_j = sum_i Mi[i][j] vel_i */
  {
    REAL gMi[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        gMi[i][j] = g[j] * vel[i];
        gvel[i] += Mi[i][j] * g[j];
      }
    rows_ib_vjp(ph, th, ps, gMi, &gph, &gth, &gps);
  }
  /* f = Rbw aero + M0 grav + thrust */
  const REAL gT = gf[0] * p->ceps + gf[2] * p->seps;
  {
    REAL gM0[3][3], gg[3] = {0, 0, 0}, gps0 = 0;
    memset(gM0, 0, sizeof(gM0));
    matvec_vjp(M0, grav, gf, gM0, gg);
    rows_ib_vjp(ph, th, 0, gM0, &gph, &gth, &gps0); /* psi = 0 is a constant */
  }
  REAL gRbw[3][3], gaero[3] = {0, 0, 0};
  memset(gRbw, 0, sizeof(gRbw));
  matvec_vjp(Rbw, aero, gf, gRbw, gaero);
  const REAL gD = -gaero[0], gY = gaero[1], gL = -gaero[2];
  /* entries of Rbw -> sin / cos of alpha, beta */
  REAL gsa = 0, gca = 0, gsb = 0, gcb = 0;
  gca += gRbw[0][0] * cb; gcb += gRbw[0][0] * ca;
  gca -= gRbw[0][1] * sb; gsb -= gRbw[0][1] * ca;
  gsa -= gRbw[0][2];
  gsb += gRbw[1][0]; gcb += gRbw[1][1];
  gsa += gRbw[2][0] * cb; gcb += gRbw[2][0] * sa;
  gsa -= gRbw[2][1] * sb; gsb -= gRbw[2][1] * sa;
  gca += gRbw[2][2];
  REAL gal = gsa * ca - gca * sa, gbe = gsb * cb - gcb * sb;
  /* forces and moments */
  REAL gqS = gL * CL + gD * CD + gY * CY + p->c * (gMb[0] * Cl + gMb[1] * Cm + gMb[2] * Cn);
  const REAL gCL = gL * qS, gCD = gD * qS, gCY = gY * qS;
  const REAL gCl = gMb[0] * qS * p->c, gCm = gMb[1] * qS * p->c, gCn = gMb[2] * qS * p->c;
  /* coefficients */
  gal += p->CLa * gCL + p->CDa * gCD + p->Cma * gCm;
  gbe += p->CYb * gCY + p->Clb * gCl + p->Cnb * gCn;
  const REAL gq_c = p->CLqc * gCL + p->CDqc * gCD + p->Cmqc * gCm;
  const REAL gp_c = p->CYpb * gCY + p->Clpb * gCl + p->Cnpb * gCn;
  const REAL gr_c = p->CYrb * gCY + p->Clrb * gCl + p->Cnrb * gCn;
  gom[0] += gp_c * i2V, gom[1] += gq_c * i2V, gom[2] += gr_c * i2V;
  const REAL gi2V = gq_c * om[1] + gp_c * om[0] + gr_c * om[2];
  const REAL gde = p->CLde * gCL + p->CDde * gCD + p->Cmde * gCm;
  const REAL gda = p->CYda * gCY + p->Clda * gCl + p->Cnda * gCn;
  const REAL gdr = p->CYdr * gCY + p->Cldr * gCl + p->Cndr * gCn;
  /* qS = half_rho V^2 S, i2V = 1 / (2V) */
  REAL gV = gqS * p->half_rho * p->S * 2 * V - gi2V / (2 * V * V);
  /* clamp, atan */
  if (be_in) {
    const REAL t = v / V, gt = gbe / (1 + t * t);
    gvel[1] += gt / V;
    gV -= gt * v / (V * V);
  }
  if (al_in) {
    const REAL t = w / u, gt = gal / (1 + t * t);
    gvel[2] += gt / u;
    gvel[0] -= gt * w / (u * u);
  }
  /* V = sqrt(u^2 + v^2 + w^2) */
  gvel[0] += gV * u / V, gvel[1] += gV * v / V, gvel[2] += gV * w / V;
  /* actions */
  ga[0] = 7 * gT;
  ga[1] = gde * pi * 40 / 180, ga[2] = gda * pi * 5 / 180, ga[3] = gdr * pi * 40 / 180;
  for (int i = 0; i < 3; ++i) {
    gs[i] = gn[i];
    gs[3 + i] = gn[3 + i] + gvel[i];
    gs[9 + i] = gn[9 + i] + gom[i];
  }
  gs[6] = gn[6] + gph, gs[7] = gn[7] + gth, gs[8] = gn[8] + gps;
}

void FN(oracle_wing_step)(const OracleWingCfg *cfg, const REAL *state,
                          const REAL *action, REAL dt, int B, REAL *next) {
  WPar p;
  wing_par(cfg, &p);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    wing_step(&p, state + 12 * b, action + 4 * b, dt, next + 12 * b, NULL, NULL, NULL);
}

void FN(oracle_wing_step_vjp)(const OracleWingCfg *cfg, const REAL *state,
                              const REAL *action, REAL dt, int B, const REAL *gnext,
                              REAL *gstate, REAL *gaction) {
  WPar p;
  wing_par(cfg, &p);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    wing_step(&p, state + 12 * b, action + 4 * b, dt, NULL, gnext + 12 * b,
              gstate + 12 * b, gaction + 4 * b);
}

/* H-step unroll + fixed_wing_mpc_loss (10 pos, 0.1 action[1:]) + reverse
 * sweep; state0[B,12], actions[B,H,4], ref[B,H,3], states[B,H,12]. */
double FN(oracle_wing_rollout_fwd_bwd)(const OracleWingCfg *cfg, const REAL *state0,
                                       const REAL *actions, const REAL *ref, REAL dt,
                                       int B, int H, REAL *states, REAL *gactions,
                                       REAL *gstate0) {
  WPar p;
  wing_par(cfg, &p);
  const REAL wp = 10, wa = (REAL)0.1;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int b = 0; b < B; ++b) {
    REAL *st = (REAL *)malloc(sizeof(REAL) * 12 * (H + 1));
    memcpy(st, state0 + 12 * b, sizeof(REAL) * 12);
    double loss = 0.0;
    for (int k = 0; k < H; ++k) {
      const REAL *a = actions + ((size_t)b * H + k) * 4, *r = ref + ((size_t)b * H + k) * 3;
      REAL *n = st + 12 * (k + 1);
      wing_step(&p, st + 12 * k, a, dt, n, NULL, NULL, NULL);
      for (int i = 0; i < 3; ++i) {
        const REAL dp = n[i] - r[i], d = a[1 + i] - (REAL).5;
        loss += wp * dp * dp + wa * d * d;
      }
    }
    total += loss;
    if (states) memcpy(states + (size_t)b * H * 12, st + 12, sizeof(REAL) * 12 * H);
    REAL lam[12] = {0}, gs[12], ga[4];
    for (int k = H - 1; k >= 0; --k) {
      const REAL *a = actions + ((size_t)b * H + k) * 4, *r = ref + ((size_t)b * H + k) * 3;
      const REAL *n = st + 12 * (k + 1);
      for (int i = 0; i < 3; ++i) lam[i] += 2 * wp * (n[i] - r[i]);
      wing_step(&p, st + 12 * k, a, dt, NULL, lam, gs, ga);
      memcpy(lam, gs, sizeof(lam));
      REAL *g = gactions + ((size_t)b * H + k) * 4;
      g[0] = ga[0];
      for (int i = 1; i < 4; ++i) g[i] = ga[i] + 2 * wa * (a[i] - (REAL).5);
    }
    if (gstate0) memcpy(gstate0 + 12 * b, lam, sizeof(lam));
    free(st);
  }
  return total;
}

/* -------------------------------------------------------------- cartpole */
typedef struct OracleCartpoleCfg {
  double masscart, masspole, length, max_force_mag, friction, gravity;
} OracleCartpoleCfg;

static void cart_step(const OracleCartpoleCfg *c, const REAL *s, const REAL *a,
                      REAL dt, REAL *o, const REAL *gn, REAL *gs, REAL *ga) {
  /* python-double constants of the reference */
  const REAL total_mass = (REAL)(c->masspole + c->masscart);
  const REAL pml = (REAL)(c->masspole * c->length);
  const REAL mp = (REAL)c->masspole, len = (REAL)c->length, mu = (REAL)c->friction,
             grav = (REAL)c->gravity;
  const REAL x = s[0], xd = s[1], th = s[2], thd = s[3];
  const REAL F = a[0] * (REAL)c->max_force_mag * (REAL)0.5; /* :60 */
  const REAL sn = sin(th), cs = cos(th);
  /* _calculate_xdot_update :86-98 */
  const REAL nx = -2 * pml * (thd * thd) * sn + 3 * mp * grav * sn * cs + 4 * F - 4 * mu * xd;
  const REAL dx = 4 * total_mass - 3 * mp * cs * cs;
  const REAL xacc = nx / dx;
  /* _calculate_thetadot_update :100-112 */
  const REAL nt = -3 * pml * (thd * thd) * sn * cs + 6 * total_mass * grav * sn +
                  6 * (F - mu * xd) * cs;
  const REAL dth = 4 * len * total_mass - 3 * pml * cs * cs;
  const REAL tacc = nt / dth;
  /* _calculate_theta_update :114-119, atan2 :76 */
  const REAL sd = sin(thd * dt), cd = cos(thd * dt);
  const REAL ns = sn * cd + cs * sd, nc = cs * cd - sn * sd;
  if (o) {
    o[0] = x + xd * dt;
    o[1] = xd + xacc * dt;
    o[2] = atan2(ns, nc);
    o[3] = thd + tacc * dt;
  }
  if (!gn) return;
  REAL gx = gn[0], gxd = gn[0] * dt + gn[1], gth = 0, gthd = gn[3], gF = 0;
  REAL gsn = 0, gcs = 0;
  /* atan2(ns, nc) */
  const REAL r2 = ns * ns + nc * nc;
  const REAL gns = gn[2] * nc / r2, gnc = -gn[2] * ns / r2;
  gsn += gns * cd - gnc * sd;
  gcs += gns * sd + gnc * cd;
  const REAL gsd = gns * cs - gnc * sn, gcd = gns * sn + gnc * cs;
  gthd += (gsd * cd - gcd * sd) * dt;
  /* xacc = nx / dx */
  const REAL gxacc = gn[1] * dt, gnx = gxacc / dx, gdx = -gxacc * nx / (dx * dx);
  gthd += gnx * (-2 * pml * 2 * thd * sn);
  gsn += gnx * (-2 * pml * thd * thd + 3 * mp * grav * cs);
  gcs += gnx * (3 * mp * grav * sn) + gdx * (-3 * mp * 2 * cs);
  gF += gnx * 4;
  gxd += gnx * (-4 * mu);
  /* tacc = nt / dth */
  const REAL gtacc = gn[3] * dt, gnt = gtacc / dth, gdth = -gtacc * nt / (dth * dth);
  gthd += gnt * (-3 * pml * 2 * thd * sn * cs);
  gsn += gnt * (-3 * pml * thd * thd * cs + 6 * total_mass * grav);
  gcs += gnt * (-3 * pml * thd * thd * sn + 6 * (F - mu * xd)) + gdth * (-3 * pml * 2 * cs);
  gF += gnt * 6 * cs;
  gxd += gnt * (-6 * mu * cs);
  gth += gsn * cs - gcs * sn;
  gs[0] = gx, gs[1] = gxd, gs[2] = gth, gs[3] = gthd;
  ga[0] = gF * (REAL)c->max_force_mag * (REAL)0.5;
}

void FN(oracle_cartpole_step)(const OracleCartpoleCfg *cfg, const REAL *state,
                              const REAL *action, REAL dt, int B, REAL *next) {
  for (int b = 0; b < B; ++b)
    cart_step(cfg, state + 4 * b, action + b, dt, next + 4 * b, NULL, NULL, NULL);
}

void FN(oracle_cartpole_step_vjp)(const OracleCartpoleCfg *cfg, const REAL *state,
                                  const REAL *action, REAL dt, int B,
                                  const REAL *gnext, REAL *gstate, REAL *gaction) {
  for (int b = 0; b < B; ++b)
    cart_step(cfg, state + 4 * b, action + b, dt, NULL, gnext + 4 * b, gstate + 4 * b,
              gaction + b);
}

/* make_reference (ref_k = s0 (1 - k/(H-1)) for k < H-1, last row 0) + H-step
 * unroll + cartpole_loss_mpc (weights 0, 3, 10, 1; 0.01 sum a^2) + reverse
 * sweep (the reference depends on state0: with ref_grad its cotangent flows
 * into gstate0, as autograd does in the trainer's graph; without, the
 * reference is a constant).  state0[B,4], actions[B,H,1], states[B,H,4]. */
double FN(oracle_cartpole_rollout_fwd_bwd)(const OracleCartpoleCfg *cfg,
                                           const REAL *state0, const REAL *actions,
                                           REAL dt, int B, int H, int ref_grad,
                                           REAL *states, REAL *gactions,
                                           REAL *gstate0) {
  const REAL wq[4] = {0, 3, 10, 1}, wa = (REAL)0.01;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int b = 0; b < B; ++b) {
    REAL *st = (REAL *)malloc(sizeof(REAL) * 4 * (H + 1));
    const REAL *s0 = state0 + 4 * b;
    memcpy(st, s0, sizeof(REAL) * 4);
    double loss = 0.0;
    for (int k = 0; k < H; ++k) {
      const REAL *a = actions + (size_t)b * H + k;
      REAL *n = st + 4 * (k + 1);
      cart_step(cfg, st + 4 * k, a, dt, n, NULL, NULL, NULL);
      const REAL sc = k < H - 1 ? 1 - (REAL)1 / (REAL)(H - 1) * (REAL)k : 0;
      for (int i = 0; i < 4; ++i) {
        const REAL d = n[i] - s0[i] * sc;
        loss += wq[i] * d * d;
      }
      loss += wa * a[0] * a[0];
    }
    total += loss;
    if (states) memcpy(states + (size_t)b * H * 4, st + 4, sizeof(REAL) * 4 * H);
    REAL lam[4] = {0, 0, 0, 0}, gref0[4] = {0, 0, 0, 0}, gs[4], ga[1];
    for (int k = H - 1; k >= 0; --k) {
      const REAL *a = actions + (size_t)b * H + k;
      const REAL *n = st + 4 * (k + 1);
      const REAL sc = k < H - 1 ? 1 - (REAL)1 / (REAL)(H - 1) * (REAL)k : 0;
      for (int i = 0; i < 4; ++i) {
        const REAL gd = 2 * wq[i] * (n[i] - s0[i] * sc);
        lam[i] += gd;
        gref0[i] -= gd * sc;
      }
      cart_step(cfg, st + 4 * k, a, dt, NULL, lam, gs, ga);
      memcpy(lam, gs, sizeof(lam));
      gactions[(size_t)b * H + k] = ga[0] + 2 * wa * a[0];
    }
    if (gstate0)
      for (int i = 0; i < 4; ++i)
        gstate0[4 * b + i] = lam[i] + (ref_grad ? gref0[i] : 0);
    free(st);
  }
  return total;
}
