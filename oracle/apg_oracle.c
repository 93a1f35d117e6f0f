/*
 * ORACLE - test infrastructure, never the product path.
 *
 * Plain-C restatement of the reference's quadrotor APG hot path, following
 * the reference's op sequence in its matrix form (NOT the closed form the HIP
 * kernels use), with a hand-written reverse sweep of that same op sequence:
 *   FlightmareDynamics.simulate_quadrotor
 *       neural_control/dynamics/quad_dynamics_flightmare.py:128-216
 *   run_flight_control / linear_dynamics                       :74-117
 *   Dynamics.world_to_body_matrix / to_euler_matrix / euler_rate
 *       neural_control/dynamics/quad_dynamics_base.py:59-127
 *   quad_mpc_loss            neural_control/drone_loss.py:12-39
 *   the k-step unroll        scripts/train_drone.py:181-197
 * Compiled twice (REAL = float / double) into liboracle.so; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline may load it.  Pinned
 * against tests/golden/quad_*.npz by tests/test_oracle_c.py.
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

typedef struct OracleQuadCfg { /* config_quad.json after modified_params */
  double mass, arm_length;
  double frame_inertia[3], gravity[3], kinv[3], rot_drag[3], trans_drag[3];
} OracleQuadCfg;

typedef struct {
  REAL mass, J[3], Jinv[3], K[3], g[3], dr[3], dt_[3];
} Par;

static void make_par(const OracleQuadCfg *c, Par *p) {
  p->mass = (REAL)c->mass;
  for (int i = 0; i < 3; ++i) {
    /* inertia_vector is computed in double, then .float() (base.py:33-47) */
    float J = (float)(c->mass / 12.0 * c->arm_length * c->arm_length *
                      c->frame_inertia[i]);
    p->J[i] = (REAL)J;
    p->Jinv[i] = (REAL)1 / p->J[i];          /* torch.inverse of a diagonal */
    p->K[i] = (REAL)(float)c->kinv[i];
    p->g[i] = (REAL)c->gravity[i];
    p->dr[i] = (REAL)(float)c->rot_drag[i];
    p->dt_[i] = (REAL)(float)c->trans_drag[i];
  }
}

/* world_to_body_matrix, quad_dynamics_base.py:59-94 */
static void w2b(const REAL att[3], REAL M[3][3]) {
  REAL sr = sin(att[0]), cr = cos(att[0]), sp = sin(att[1]), cp = cos(att[1]),
       sy = sin(att[2]), cy = cos(att[2]);
  M[0][0] = cy * cp, M[0][1] = sy * cp, M[0][2] = -sp;
  M[1][0] = cy * sp * sr - cr * sy, M[1][1] = cr * cy + sr * sy * sp, M[1][2] = cp * sr;
  M[2][0] = cy * sp * cr + sr * sy, M[2][1] = cr * sy * sp - cy * sr, M[2][2] = cr * cp;
}

/* reverse of w2b: gatt += (dM/datt)^T gM, entry by entry */
static void w2b_vjp(const REAL att[3], const REAL gM[3][3], REAL gatt[3]) {
  REAL sr = sin(att[0]), cr = cos(att[0]), sp = sin(att[1]), cp = cos(att[1]),
       sy = sin(att[2]), cy = cos(att[2]);
  /* cotangents of the six trig values */
  REAL gsr = 0, gcr = 0, gsp = 0, gcp = 0, gsy = 0, gcy = 0;
  /* M00 = cy*cp */            gcy += gM[0][0] * cp; gcp += gM[0][0] * cy;
  /* M01 = sy*cp */            gsy += gM[0][1] * cp; gcp += gM[0][1] * sy;
  /* M02 = -sp */              gsp -= gM[0][2];
  /* M10 = cy*sp*sr - cr*sy */ gcy += gM[1][0] * sp * sr; gsp += gM[1][0] * cy * sr;
                               gsr += gM[1][0] * cy * sp; gcr -= gM[1][0] * sy; gsy -= gM[1][0] * cr;
  /* M11 = cr*cy + sr*sy*sp */ gcr += gM[1][1] * cy; gcy += gM[1][1] * cr;
                               gsr += gM[1][1] * sy * sp; gsy += gM[1][1] * sr * sp; gsp += gM[1][1] * sr * sy;
  /* M12 = cp*sr */            gcp += gM[1][2] * sr; gsr += gM[1][2] * cp;
  /* M20 = cy*sp*cr + sr*sy */ gcy += gM[2][0] * sp * cr; gsp += gM[2][0] * cy * cr;
                               gcr += gM[2][0] * cy * sp; gsr += gM[2][0] * sy; gsy += gM[2][0] * sr;
  /* M21 = cr*sy*sp - cy*sr */ gcr += gM[2][1] * sy * sp; gsy += gM[2][1] * cr * sp;
                               gsp += gM[2][1] * cr * sy; gcy -= gM[2][1] * sr; gsr -= gM[2][1] * cy;
  /* M22 = cr*cp */            gcr += gM[2][2] * cp; gcp += gM[2][2] * cr;
  gatt[0] += gsr * cr - gcr * sr;
  gatt[1] += gsp * cp - gcp * sp;
  gatt[2] += gsy * cy - gcy * sy;
}

/* one step; state = [pos, att, vel, omega] */
static void step(const Par *p, const REAL *s, const REAL *a, REAL dt, REAL *o) {
  const REAL *pos = s, *att = s + 3, *vel = s + 6, *om = s + 9;
  REAL thrust = a[0] * 15 - (REAL)7.5 + (REAL)9.81;            /* :139 */
  REAL rates[3] = {a[1] - (REAL).5, a[2] - (REAL).5, a[3] - (REAL).5};
  REAL Jo[3] = {p->J[0] * om[0], p->J[1] * om[1], p->J[2] * om[2]};
  REAL cross[3] = {om[1] * Jo[2] - om[2] * Jo[1], om[2] * Jo[0] - om[0] * Jo[2],
                   om[0] * Jo[1] - om[1] * Jo[0]};              /* :146-149 */
  REAL force = p->mass * thrust;                                /* :101 */
  REAL tau[3], M[3][3];
  for (int i = 0; i < 3; ++i)                                   /* :104-112 */
    tau[i] = p->J[i] * (p->K[i] * (rates[i] - om[i])) + cross[i] + p->dr[i];
  w2b(att, M);
  for (int i = 0; i < 3; ++i) {                                 /* :84-92, :172-175 */
    REAL acc = ((REAL)1 / p->mass) * (M[2][i] * force) + p->g[i] + p->dt_[i];
    o[i] = pos[i] + (REAL)0.5 * dt * dt * acc + (REAL)0.5 * dt * vel[i];
    o[6 + i] = vel[i] + dt * acc;
    o[9 + i] = om[i] + dt * (p->Jinv[i] * (tau[i] - cross[i])); /* :178-183 */
  }
  REAL sr = sin(att[0]), cr = cos(att[0]), sp = sin(att[1]), cp = cos(att[1]);
  /* to_euler_matrix @ omega, base.py:96-127, old omega (:210) */
  o[3] = att[0] + dt * (om[0] - sp * om[2]);
  o[4] = att[1] + dt * (cr * om[1] + cp * sr * om[2]);
  o[5] = att[2] + dt * (-sr * om[1] + cp * cr * om[2]);
}

/* reverse sweep of `step`: gs = J_s^T gn, ga = J_a^T gn (overwritten) */
static void step_vjp(const Par *p, const REAL *s, const REAL *a, REAL dt,
                     const REAL *gn, REAL *gs, REAL *ga) {
  const REAL *att = s + 3, *om = s + 9;
  REAL thrust = a[0] * 15 - (REAL)7.5 + (REAL)9.81;
  REAL force = p->mass * thrust;
  REAL M[3][3], gM[3][3] = {{0}};
  w2b(att, M);
  REAL gforce = 0, gatt[3] = {gn[3], gn[4], gn[5]}, gom[3], gvel[3], gcross[3], gtau[3];
  for (int i = 0; i < 3; ++i) {
    REAL gacc = (REAL)0.5 * dt * dt * gn[i] + dt * gn[6 + i];
    gs[i] = gn[i];
    gvel[i] = gn[6 + i] + (REAL)0.5 * dt * gn[i];
    REAL gthr = ((REAL)1 / p->mass) * gacc;         /* cotangent of M[2][i]*force */
    gM[2][i] = gthr * force;
    gforce += gthr * M[2][i];
    REAL gangacc = dt * gn[9 + i];
    gtau[i] = p->Jinv[i] * gangacc;
    gcross[i] = -gtau[i] + gtau[i];                 /* subtracted :181, added :110 */
    gom[i] = gn[9 + i] - p->J[i] * p->K[i] * gtau[i];
    ga[1 + i] = p->J[i] * p->K[i] * gtau[i];
  }
  (void)gcross;
  ga[0] = p->mass * gforce * 15;
  /* attitude update */
  REAL sr = sin(att[0]), cr = cos(att[0]), sp = sin(att[1]), cp = cos(att[1]);
  REAL e0 = dt * gn[3], e1 = dt * gn[4], e2 = dt * gn[5];
  gom[0] += e0;
  gom[1] += e1 * cr - e2 * sr;
  gom[2] += -e0 * sp + e1 * cp * sr + e2 * cp * cr;
  gatt[0] += e1 * (-sr * om[1] + cp * cr * om[2]) + e2 * (-cr * om[1] - cp * sr * om[2]);
  gatt[1] += e0 * (-cp * om[2]) + e1 * (-sp * sr * om[2]) + e2 * (-sp * cr * om[2]);
  w2b_vjp(att, gM, gatt);
  for (int i = 0; i < 3; ++i) gs[3 + i] = gatt[i], gs[6 + i] = gvel[i], gs[9 + i] = gom[i];
}

void FN(oracle_quad_step)(const OracleQuadCfg *cfg, const REAL *state,
                          const REAL *action, REAL dt, int B, REAL *next) {
  Par p;
  make_par(cfg, &p);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) step(&p, state + 12 * b, action + 4 * b, dt, next + 12 * b);
}

void FN(oracle_quad_step_vjp)(const OracleQuadCfg *cfg, const REAL *state,
                              const REAL *action, REAL dt, int B,
                              const REAL *gnext, REAL *gstate, REAL *gaction) {
  Par p;
  make_par(cfg, &p);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b)
    step_vjp(&p, state + 12 * b, action + 4 * b, dt, gnext + 12 * b, gstate + 12 * b,
             gaction + 4 * b);
}

/* H-step unroll + quad_mpc_loss + reverse sweep.  All tensors row-major as in
 * the reference: state0[B,12], actions[B,H,4], ref[B,H,9], states[B,H,12].
 * Returns the loss (sum over batch and horizon) accumulated in double. */
double FN(oracle_quad_rollout_fwd_bwd)(const OracleQuadCfg *cfg,
                                       const REAL *state0, const REAL *actions,
                                       const REAL *ref, REAL dt, int B, int H,
                                       REAL *states, REAL *gactions,
                                       REAL *gstate0) {
  Par p;
  make_par(cfg, &p);
  const REAL wp = 10, wv = 1, wa = (REAL)0.1, wr = (REAL)0.1, wt = 5;
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int b = 0; b < B; ++b) {
    REAL *st = (REAL *)malloc(sizeof(REAL) * 12 * (H + 1));
    memcpy(st, state0 + 12 * b, sizeof(REAL) * 12);
    double loss = 0.0;
    for (int k = 0; k < H; ++k) {
      const REAL *a = actions + ((size_t)b * H + k) * 4, *r = ref + ((size_t)b * H + k) * 9;
      REAL *n = st + 12 * (k + 1);
      step(&p, st + 12 * k, a, dt, n);
      for (int i = 0; i < 3; ++i) {
        REAL dp = n[i] - r[i], dv = n[6 + i] - r[6 + i], da = a[1 + i] - (REAL).5;
        loss += wp * dp * dp + wv * dv * dv + wa * n[9 + i] * n[9 + i] + wr * da * da;
      }
      loss += wt * (a[0] - (REAL).5) * (a[0] - (REAL).5);
    }
    total += loss;
    if (states) memcpy(states + (size_t)b * H * 12, st + 12, sizeof(REAL) * 12 * H);
    REAL lam[12] = {0}, gs[12], ga[4];
    for (int k = H - 1; k >= 0; --k) {
      const REAL *a = actions + ((size_t)b * H + k) * 4, *r = ref + ((size_t)b * H + k) * 9;
      const REAL *n = st + 12 * (k + 1);
      for (int i = 0; i < 3; ++i) {
        lam[i] += 2 * wp * (n[i] - r[i]);
        lam[6 + i] += 2 * wv * (n[6 + i] - r[6 + i]);
        lam[9 + i] += 2 * wa * n[9 + i];
      }
      step_vjp(&p, st + 12 * k, a, dt, lam, gs, ga);
      memcpy(lam, gs, sizeof(lam));
      REAL *g = gactions + ((size_t)b * H + k) * 4;
      g[0] = ga[0] + 2 * wt * (a[0] - (REAL).5);
      for (int i = 1; i < 4; ++i) g[i] = ga[i] + 2 * wr * (a[i] - (REAL).5);
    }
    if (gstate0) memcpy(gstate0 + 12 * b, lam, sizeof(lam));
    free(st);
  }
  return total;
}
