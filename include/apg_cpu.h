/*
 * apg_cpu.h - C ABI of libapg_cpu.so: the HOST twins of the dynamics entry
 * points of apg.h ("CPU twins ..._cpu with identical signatures
 * minus the stream").
 *
 * What it is.  The per-trajectory arithmetic of every GPU kernel lives in
 * headers that compile for both sides (csrc/quad_math.h, wing_math.h,
 * cartpole_math.h: step, adjoint, constants).  libapg_cpu.so is those headers
 * compiled for the host behind the signatures of apg.h: the same functions a
 * lane of the GPU kernels runs, looped over the batch.  It restates nothing -
 * it is not the oracle (oracle/ is an independent restatement of the
 * REFERENCE and never part of the product) - and it shares no code path with
 * the GPU library at run time.
 *
 * What it is not.  A fallback.  The Python package never loads this library:
 * a CPU tensor handed to the package raises, a missing libapg_hip.so raises.
 * The twins exist for callers of the C ABI that have no GPU at hand (a
 * reference-side script that evaluates a controller on a laptop, a debugger
 * session on one trajectory), who opt in by linking a differently named
 * library and calling differently named symbols.
 *
 * Differences from apg.h:
 *  - all pointers are HOST pointers; every call is synchronous and
 *    single-threaded (no stream argument);
 *  - results agree with the GPU entry points to fp32 rounding (the host
 *    compiler contracts and orders a few operations differently; the special
 *    functions are the same sincos_fast / rational approximations);
 *  - loss_partials holds one partial per 64 trajectories, as on the GPU
 *    (apg_loss_partials_count(B) = ceil(B / 64)), summed in trajectory order;
 *    `loss` is their fixed-order sum;
 *  - the policy-inside entry points (apg_quad_mlp_*, apg_quad_lstm_*,
 *    apg_wing_policy_*), the learnt simulators, the matrix products and the
 *    layout helpers have no twin: they are matrix-core kernels without
 *    shared per-lane headers.
 * Errors: same codes; apg_cpu_last_error_string().
 */
#ifndef APG_CPU_H_
#define APG_CPU_H_

#include "apg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* apg_quad_step_fwd / _bwd (neural_control/dynamics/quad_dynamics_flightmare.py:125-216) */
int apg_quad_step_fwd_cpu(const float *state, const float *action, float dt,
                          const ApgQuadParams *params, int B, int layout,
                          float *next_state);
int apg_quad_step_bwd_cpu(const float *state, const float *action, float dt,
                          const ApgQuadParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action);
/* apg_quad_rollout_fwd_bwd (scripts/train_drone.py:181-197); all three layouts */
int apg_quad_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                 const float *ref, int ref_cols, float dt,
                                 const ApgQuadParams *params,
                                 const ApgQuadLossWeights *weights, int B, int H,
                                 int layout, float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out,
                                 const ApgDeferredLoss *deferred);
int apg_quad_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                             const ApgQuadParams *params, int B, int H, int layout,
                             float *states_out);

/* apg_wing_* (neural_control/dynamics/fixed_wing_dynamics.py:95-267,
 * scripts/train_fixed_wing.py:90-110) */
int apg_wing_step_fwd_cpu(const float *state, const float *action, float dt,
                          const ApgWingParams *params, int B, int layout,
                          float *next_state);
int apg_wing_step_bwd_cpu(const float *state, const float *action, float dt,
                          const ApgWingParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action);
int apg_wing_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                 const float *ref, float dt,
                                 const ApgWingParams *params,
                                 const ApgWingLossWeights *weights, int B, int H,
                                 int layout, float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out,
                                 const ApgDeferredLoss *deferred);
int apg_wing_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                             const ApgWingParams *params, int B, int H, int layout,
                             float *states_out);

/* apg_cartpole_* (neural_control/dynamics/cartpole_dynamics.py:50-119,
 * scripts/train_cartpole.py:103-150) */
int apg_cartpole_step_fwd_cpu(const float *state, const float *action, float dt,
                              const ApgCartpoleParams *params, int B, int layout,
                              float *next_state);
int apg_cartpole_step_bwd_cpu(const float *state, const float *action, float dt,
                              const ApgCartpoleParams *params, int B, int layout,
                              const float *grad_next, float *grad_state,
                              float *grad_action);
int apg_cartpole_rollout_fwd_bwd_cpu(const float *state0, const float *actions,
                                     float dt, const ApgCartpoleParams *params,
                                     int B, int H, int layout,
                                     float *loss_partials, float *loss,
                                     float *grad_actions, float *grad_state0,
                                     float *states_out);
int apg_cartpole_rollout_fwd_cpu(const float *state0, const float *actions, float dt,
                                 const ApgCartpoleParams *params, int B, int H,
                                 int layout, float *states_out);

int apg_cpu_version(void);
const char *apg_cpu_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* APG_CPU_H_ */
