// Test infrastructure: the per-trajectory rollout of csrc/quad_lane_pk.h (the
// very code the packed rollout kernel runs per lane) compiled for the HOST,
// so that its arithmetic can be checked against the golden vectors without a
// GPU.  Row-major [B,...] tensors as in the reference.
#include "quad_lane_pk.h"

namespace {
struct HostIO {
  const float *s0_, *act_, *ref_;
  float *ga_, *gs_, *states_;
  int ref_cols, vel_col;
  float s0(int i) const { return s0_[i]; }
  float act(int k, int i) const { return act_[k * 4 + i]; }
  float ref_p(int k, int i) const { return ref_[k * ref_cols + i]; }
  float ref_v(int k, int i) const { return ref_[k * ref_cols + vel_col + i]; }
  void ga(int k, int i, float v) { ga_[k * 4 + i] = v; }
  void gs(int i, float v) { gs_[i] = v; }
  void state(int k, int i, float v) { states_[k * 12 + i] = v; }
};

template <int HT>
double run(const float *state0, const float *actions, const float *ref,
           int ref_cols, float dt, const ApgQuadParams *p,
           const ApgQuadLossWeights *w, int B, float *states, float *ga,
           float *gs) {
  const apg::pk::Const c = apg::pk::make_const(*p, dt);
  double loss = 0.0;
  for (int b = 0; b < B; ++b) {
    HostIO io{state0 + b * 12, actions + b * HT * 4, ref + b * HT * ref_cols,
              ga + b * HT * 4, gs + b * 12, states + b * HT * 12, ref_cols,
              ref_cols == 9 ? 6 : 3};
    loss += apg::pk::rollout_lane<HT, true>(io, c, *w, true);
  }
  return loss;
}
}  // namespace

extern "C" double quad_lane_rollout_host(const float *state0, const float *actions,
                                         const float *ref, int ref_cols, float dt,
                                         const ApgQuadParams *p,
                                         const ApgQuadLossWeights *w, int B, int H,
                                         float *states, float *ga, float *gs) {
  if (H == 10) return run<10>(state0, actions, ref, ref_cols, dt, p, w, B, states, ga, gs);
  if (H == 5) return run<5>(state0, actions, ref, ref_cols, dt, p, w, B, states, ga, gs);
  return -1.0;
}

// one step + VJP (golden G1)
extern "C" void quad_lane_step_host(const float *state, const float *action, float dt,
                                    const ApgQuadParams *p, int B, const float *cot,
                                    float *next, float *gstate, float *gaction) {
  using namespace apg::pk;
  const Const c = make_const(*p, dt);
  for (int b = 0; b < B; ++b) {
    const float *r = state + b * 12, *a = action + b * 4;
    State s;
    s.p01 = (f2){r[0], r[1]}, s.p2 = r[2], s.phi = r[3], s.tp = (f2){r[4], r[5]};
    s.v01 = (f2){r[6], r[7]}, s.v2 = r[8], s.w01 = (f2){r[9], r[10]}, s.w2 = r[11];
    const Trig t = make_trig(s.phi, s.tp);
    const f2 w01 = s.w01;
    const float w2 = s.w2;
    step(s, a[0], (f2){a[1], a[2]}, a[3], c, t);
    float *o = next + b * 12;
    o[0] = s.p01.x, o[1] = s.p01.y, o[2] = s.p2, o[3] = s.phi, o[4] = s.tp.x;
    o[5] = s.tp.y, o[6] = s.v01.x, o[7] = s.v01.y, o[8] = s.v2, o[9] = s.w01.x;
    o[10] = s.w01.y, o[11] = s.w2;
    if (!cot) continue;
    const float *q = cot + b * 12;
    Adj l;
    l.p01 = (f2){q[0], q[1]}, l.p2 = q[2], l.phi = q[3], l.tp = (f2){q[4], q[5]};
    l.v01 = (f2){q[6], q[7]}, l.v2 = q[8], l.w01 = (f2){q[9], q[10]}, l.w2 = q[11];
    float g0 = 0.f, g3 = 0.f;
    f2 g12 = bc(0.f);
    step_adjoint(l, g0, g12, g3, a[0], w01, w2, c, t);
    float *gs = gstate + b * 12, *ga = gaction + b * 4;
    gs[0] = l.p01.x, gs[1] = l.p01.y, gs[2] = l.p2, gs[3] = l.phi, gs[4] = l.tp.x;
    gs[5] = l.tp.y, gs[6] = l.v01.x, gs[7] = l.v01.y, gs[8] = l.v2, gs[9] = l.w01.x;
    gs[10] = l.w01.y, gs[11] = l.w2;
    ga[0] = g0, ga[1] = g12.x, ga[2] = g12.y, ga[3] = g3;
  }
}
