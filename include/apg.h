/*
 * apg.h - C ABI of libapg_hip.so: the MI355X (gfx950) implementation of the
 * APG training-rollout hot path of lis-epfl/apg_trajectory_tracking.
 *
 * The reference is pure Python and has no FFI of its own; the entry points
 * below are what a ctypes binding for this path binds (INTEGRATION.md shows
 * the stub).  Each one cites the reference code it replaces (paths relative
 * to the reference repository root).
 *
 * Conventions
 *  - All buffers are DEVICE pointers owned by the caller (e.g. the PyTorch
 *    caching allocator); the library allocates nothing and keeps no state
 *    except a thread-local error string.
 *  - All arithmetic is IEEE fp32, as in the reference
 *    (neural_control/dynamics/quad_dynamics_flightmare.py:216).
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t, may be NULL
 *    for the default stream) and never synchronises.
 *  - Return value: APG_OK (0) or a negative APG_ERR_* code;
 *    apg_last_error_string() describes the last failure on this thread.
 *  - Layouts.  B = batch, H = horizon, S = state size, A = action size.
 *      APG_LAYOUT_AOS  reference row-major tensors:
 *                      state[B][S], seq[B][H][C]   (C = A, or ref columns)
 *      APG_LAYOUT_SOA  device-native, batch fastest:
 *                      state[S][B], seq[H][C][B]
 *    The fused rollout kernels read one trajectory per lane; with SOA every
 *    wave-wide load/store is one fully coalesced 256-byte transaction.
 */
#ifndef APG_H_
#define APG_H_

#ifdef __cplusplus
extern "C" {
#endif

#define APG_VERSION_MAJOR 0
#define APG_VERSION_MINOR 1

typedef void *apg_stream_t; /* hipStream_t */

enum { APG_LAYOUT_SOA = 0, APG_LAYOUT_AOS = 1 };

enum {
  APG_OK = 0,
  APG_ERR_ARG = -1,      /* bad argument (null pointer, B<0, H out of range) */
  APG_ERR_HIP = -2,      /* a HIP runtime call failed (launch, attribute)    */
  APG_ERR_NO_DEVICE = -3 /* no gfx950 device visible                          */
};

/* Threads per workgroup of the fused rollout kernels; one loss partial is
 * produced per workgroup.  apg_loss_partials_count(B) = ceil(B / 64). */
#define APG_ROLLOUT_BLOCK 64
/* Largest horizon the fused rollout kernels accept. */
#define APG_MAX_HORIZON 48

/* Deferred loss reduction.  A fused-rollout launch can fold the fixed-order
 * sum of the loss partials left by an EARLIER launch on the same stream into
 * its own kernel (workgroup 0 does it while its first loads are in flight).
 * That takes the second kernel of the `loss != NULL` mode off the per-step
 * critical path: step i's scalar loss is materialised by step i+1's launch,
 * the last one by apg_reduce_loss_partials().  `prev_partials` must not alias
 * the `loss_partials` of the launch it is passed to (ping-pong two buffers). */
typedef struct ApgDeferredLoss {
  const float *prev_partials; /* [prev_count] written by an earlier launch */
  int prev_count;
  float *prev_loss;           /* [1] receives their sum */
} ApgDeferredLoss;

/* ---------------------------------------------------------------- quad --- */
/* Parameters of neural_control/dynamics/quad_dynamics_base.py:11-57 after
 * `cfg.update(modified_params)`; `inertia` is mass/12*arm_length^2*
 * frame_inertia (:33-36).  Mass cancels out of the translational dynamics
 * (quad_dynamics_flightmare.py:84,101) and is carried for completeness. */
typedef struct ApgQuadParams {
  float mass;
  float kinv[3];       /* kinv_ang_vel_tau            */
  float inertia[3];    /* diagonal of J               */
  float gravity[3];
  float trans_drag[3]; /* constant additive vector (sic, :89-92)   */
  float rot_drag[3];   /* constant additive torque  (sic, :110-112) */
} ApgQuadParams;

/* Weights of quad_mpc_loss, neural_control/drone_loss.py:12-39
 * (reference values: pos 10, vel 1, av 0.1, rates 0.1, thrust 5). */
typedef struct ApgQuadLossWeights {
  float pos, vel, av, rates, thrust;
} ApgQuadLossWeights;

/* One step of FlightmareDynamics.__call__/simulate_quadrotor
 * (neural_control/dynamics/quad_dynamics_flightmare.py:125-216).
 * state[B,12] = [p(3), euler rpy(3), v(3), omega(3)], action[B,4] in [0,1]. */
int apg_quad_step_fwd(const float *state, const float *action, float dt,
                      const ApgQuadParams *params, int B, int layout,
                      float *next_state, apg_stream_t stream);

/* Vector-Jacobian product of the step above (what torch.autograd computes
 * for the reference): grad_state = J_s^T grad_next, grad_action = J_a^T
 * grad_next.  grad_state / grad_action may be NULL to skip that output. */
int apg_quad_step_bwd(const float *state, const float *action, float dt,
                      const ApgQuadParams *params, int B, int layout,
                      const float *grad_next, float *grad_state,
                      float *grad_action, apg_stream_t stream);

/* Fused horizon-unrolled rollout + quad_mpc_loss + analytic adjoint:
 * replaces the bracketed region of TrainDrone.train_controller_model
 * (scripts/train_drone.py:181-197): H x dynamics, quad_mpc_loss,
 * loss.backward() down to dL/daction_seq (and dL/dstate0).
 *   state0   [B,12]            actions [B,H,4]
 *   ref      [B,H,ref_cols]    ref_cols = 9: reference rows
 *                              [pos, euler, vel] (cols 3:6 never read);
 *                              ref_cols = 6: packed [pos, vel]
 *   loss_partials [apg_loss_partials_count(B)]  per-workgroup loss sums
 *   loss          [1] or NULL; if given, a second tiny kernel sums the
 *                 partials in a fixed order (deterministic)
 *   grad_actions  [B,H,4]      dL/daction_seq
 *   grad_state0   [B,12] or NULL
 *   states_out    [B,H,12] or NULL  intermediate states
 *   deferred      NULL, or an earlier launch's partials to reduce (see above)
 * all in `layout`. */
int apg_quad_rollout_fwd_bwd(const float *state0, const float *actions,
                             const float *ref, int ref_cols, float dt,
                             const ApgQuadParams *params,
                             const ApgQuadLossWeights *weights, int B, int H,
                             int layout, float *loss_partials, float *loss,
                             float *grad_actions, float *grad_state0,
                             float *states_out,
                             const ApgDeferredLoss *deferred,
                             apg_stream_t stream);

/* No-grad unroll (eval / self-play): states_out[B,H,12] only. */
int apg_quad_rollout_fwd(const float *state0, const float *actions, float dt,
                         const ApgQuadParams *params, int B, int H, int layout,
                         float *states_out, apg_stream_t stream);

/* quad_mpc_loss alone (neural_control/drone_loss.py:12-39) on materialised
 * states[B,H,12], ref[B,H,ref_cols], actions[B,H,4]; also returns
 * dL/dstates and dL/dactions (either may be NULL). */
int apg_quad_loss_fwd_bwd(const float *states, const float *ref, int ref_cols,
                          const float *actions,
                          const ApgQuadLossWeights *weights, int B, int H,
                          int layout, float *loss_partials, float *loss,
                          float *grad_states, float *grad_actions,
                          apg_stream_t stream);

/* state_preprocessing (neural_control/dataset.py:207-220): state[B,12] ->
 * features[B,15] = [v_world, R_wb[:, :, :2] flattened (6), v_body, omega]. */
int apg_quad_features_fwd(const float *state, int B, int layout,
                          float *features, apg_stream_t stream);
int apg_quad_features_bwd(const float *state, const float *grad_features,
                          int B, int layout, float *grad_state,
                          apg_stream_t stream);

/* ---------------------------------------------------------- fixed wing --- */
/* Parameters of neural_control/dynamics/fixed_wing_dynamics.py:18-39 +
 * config_fixed_wing.json after `cfg.update(modified_params)`. */
typedef struct ApgWingParams {
  float mass, I_xx, I_yy, I_zz, I_xz, rho, S, c, b, g;
  float CL0, CL_alpha, CL_q, CL_del_e;
  float CD0, CD_alpha, CD_q, CD_del_e;
  float CY0, CY_beta, CY_p, CY_r, CY_del_a, CY_del_r;
  float Cl0, Cl_beta, Cl_p, Cl_r, Cl_del_a, Cl_del_r;
  float Cm0, Cm_alpha, Cm_q, Cm_del_e;
  float Cn0, Cn_beta, Cn_p, Cn_r, Cn_del_a, Cn_del_r;
  float epsilon;
} ApgWingParams;

/* Weights of fixed_wing_mpc_loss, neural_control/drone_loss.py:72-82
 * (reference values: pos 10, action 0.1). */
typedef struct ApgWingLossWeights {
  float pos, action;
} ApgWingLossWeights;

/* FixedWingDynamics.__call__/simulate_fixed_wing
 * (neural_control/dynamics/fixed_wing_dynamics.py:95-267).
 * state[B,12] = [pos NED(3), vel body uvw(3), euler(3), omega pqr(3)]. */
int apg_wing_step_fwd(const float *state, const float *action, float dt,
                      const ApgWingParams *params, int B, int layout,
                      float *next_state, apg_stream_t stream);
int apg_wing_step_bwd(const float *state, const float *action, float dt,
                      const ApgWingParams *params, int B, int layout,
                      const float *grad_next, float *grad_state,
                      float *grad_action, apg_stream_t stream);

/* Fused rollout of TrainFixedWing.train_controller_model
 * (scripts/train_fixed_wing.py:90-110) with fixed_wing_mpc_loss.
 *   ref [B,H,3] linear reference (WingDataset._compute_target_pos,
 *   neural_control/dataset.py:309-320). */
int apg_wing_rollout_fwd_bwd(const float *state0, const float *actions,
                             const float *ref, float dt,
                             const ApgWingParams *params,
                             const ApgWingLossWeights *weights, int B, int H,
                             int layout, float *loss_partials, float *loss,
                             float *grad_actions, float *grad_state0,
                             float *states_out,
                             const ApgDeferredLoss *deferred,
                             apg_stream_t stream);
int apg_wing_rollout_fwd(const float *state0, const float *actions, float dt,
                         const ApgWingParams *params, int B, int H, int layout,
                         float *states_out, apg_stream_t stream);

/* ------------------------------------------------------------ cartpole --- */
/* neural_control/dynamics/cartpole_dynamics.py:23-43 + config_cartpole.json
 * (friction is forced to 0.5 at :34). */
typedef struct ApgCartpoleParams {
  float masscart, masspole, length, max_force_mag, friction, gravity;
} ApgCartpoleParams;

/* CartpoleDynamics.__call__/simulate_cartpole
 * (neural_control/dynamics/cartpole_dynamics.py:50-119).
 * state[B,4] = [x, x_dot, theta, theta_dot], action[B,1]. */
int apg_cartpole_step_fwd(const float *state, const float *action, float dt,
                          const ApgCartpoleParams *params, int B, int layout,
                          float *next_state, apg_stream_t stream);
int apg_cartpole_step_bwd(const float *state, const float *action, float dt,
                          const ApgCartpoleParams *params, int B, int layout,
                          const float *grad_next, float *grad_state,
                          float *grad_action, apg_stream_t stream);

/* Fused rollout of TrainCartpole.run_epoch's controller branch
 * (scripts/train_cartpole.py:131-150): make_reference (:103-110, derived
 * in-kernel from state0, gradient flows through it as in the reference),
 * H x dynamics, cartpole_loss_mpc (neural_control/drone_loss.py:136-145). */
int apg_cartpole_rollout_fwd_bwd(const float *state0, const float *actions,
                                 float dt, const ApgCartpoleParams *params,
                                 int B, int H, int layout,
                                 float *loss_partials, float *loss,
                                 float *grad_actions, float *grad_state0,
                                 float *states_out, apg_stream_t stream);

/* --------------------------------------------------------------- misc --- */
/* loss[0] = fixed-order sum of partials[0..n) (one small kernel). */
int apg_reduce_loss_partials(const float *partials, int n, float *loss,
                             apg_stream_t stream);
/* Number of floats `loss_partials` must hold for a batch of B. */
int apg_loss_partials_count(int B);
/* APG_VERSION_MAJOR * 1000 + APG_VERSION_MINOR. */
int apg_version(void);
/* Description of the last error on the calling thread ("" if none). */
const char *apg_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* APG_H_ */
