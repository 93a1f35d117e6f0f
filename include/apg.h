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
 *      APG_LAYOUT_PACKED  device-native, rows of one trajectory's floats with
 *                      the batch as the next-faster dimension (accepted by
 *                      apg_quad_rollout_fwd_bwd only, H = 5 or 10):
 *                      state[S/4][B][4], seq[H][B][C]  (actions C = 4, packed
 *                      reference rows C = 6, states_out [H][3][B][4])
 *    The fused rollout kernels read one trajectory per lane; with SOA every
 *    wave-wide load/store is one fully coalesced 256-byte transaction, with
 *    PACKED a lane moves a whole row with one 16-byte access (1 KiB per wave
 *    instruction) - a quarter of the memory instructions, which is what the
 *    one-wave-per-SIMD rollout kernel is paced by.
 *  - Operand range.  The dynamics, losses and adjoints are plain fp32 and take
 *    any finite input.  The entry points that run a POLICY NETWORK inside the
 *    kernel (apg_quad_mlp_*, apg_quad_lstm_*, apg_wing_policy_*,
 *    apg_wing_mlp_closed_loop) evaluate its layers on the 16-bit matrix pipe
 *    with every fp32 operand split into two fp16 terms (fp32 accuracy,
 *    csrc/policy_mfma16.h).  Cotangents are rescaled per trajectory inside the
 *    kernels; the first-layer inputs are not: normalised features, initial
 *    state, reference windows / trajectories and network weights must be
 *    FINITE with |x| < 16 384 (2^14; fp16 overflows at 65 504, the margin
 *    covers what a rollout adds to a state).  Smaller is always fine: the low
 *    term keeps an absolute accuracy of 2^-25.  The library cannot see device
 *    data at enqueue time - the caller checks (the Python host does, once per
 *    tensor version: functional._guard_policy_inputs raises ValueError); a
 *    violation yields inf / NaN losses, never a wrong finite number.
 *    apg_planes_gemm / apg_linear_wgrad split into bf16 terms (fp32's exponent
 *    range): finite operands of any magnitude; a non-finite operand gives a
 *    non-finite (NaN) result where fp32 arithmetic would give inf.
 */
#ifndef APG_H_
#define APG_H_

#ifdef __cplusplus
extern "C" {
#endif

#define APG_VERSION_MAJOR 0
#define APG_VERSION_MINOR 1

typedef void *apg_stream_t; /* hipStream_t */
typedef void *apg_event_t;  /* hipEvent_t */

enum { APG_LAYOUT_SOA = 0, APG_LAYOUT_AOS = 1, APG_LAYOUT_PACKED = 2 };

enum {
  APG_OK = 0,
  APG_ERR_ARG = -1,      /* bad argument (null pointer, B<0, H out of range) */
  APG_ERR_HIP = -2,      /* a HIP runtime call failed (launch, attribute)    */
  APG_ERR_NO_DEVICE = -3 /* no gfx950 device visible                          */
};

/* Threads per workgroup of the fused rollout kernels.  One loss partial is
 * produced per WAVE (64 trajectories): apg_loss_partials_count(B) =
 * ceil(B / 64) independently of this value. */
#ifndef APG_ROLLOUT_BLOCK
#define APG_ROLLOUT_BLOCK 64
#endif
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
 *   loss_partials [apg_loss_partials_count(B)]  per-WAVE loss sums (one per
 *                 64 trajectories, whatever the workgroup size)
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

/* The same fused unroll through the LEARNT simulator of
 * neural_control/dynamics/quad_dynamics_trained.py:10-69 (LearntDynamics):
 *   a' = linear_at a;  s' = simulate_quadrotor(a', s) + W2 relu(W1 [s; a'] + b1) + b2
 * for the controller phase of TrainBase.run_dynamics (scripts/train_base.py:
 * 334-375, scripts/train_drone.py:185-191): H x LearntDynamics.forward,
 * quad_mpc_loss, backward down to dL/daction_seq and dL/dstate0.  The
 * simulator's parameters are frozen in that phase: no parameter gradient is
 * produced.  `params` = the physical parameters the module simulates with (its
 * construction-time kinv / inertia); the tensors of `model` are the module's
 * own (row-major, device memory): linear_at [4,4], linear_state_1.weight
 * [64,16] / .bias [64], linear_state_2.weight [12,64] / .bias [12].
 * Layouts: APG_LAYOUT_AOS or APG_LAYOUT_SOA. */
typedef struct ApgLearntResidual {
  const float *linear_at, *w1, *b1, *w2, *b2;
} ApgLearntResidual;
int apg_quad_learnt_rollout_fwd_bwd(const float *state0, const float *actions,
                                    const float *ref, int ref_cols, float dt,
                                    const ApgQuadParams *params,
                                    const ApgLearntResidual *model,
                                    const ApgQuadLossWeights *weights, int B, int H,
                                    int layout, float *loss_partials, float *loss,
                                    float *grad_actions, float *grad_state0,
                                    float *states_out, apg_stream_t stream);

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

/* ------------------------------------------- quad, LSTM policy in-kernel --- */
/* LSTM_NEW with the conv branch (neural_control/models/rnn.py:8-51) for
 * state_dim 15, horizon 10, ref_dim 9, 4 actions: conv1d(9 -> 20, k = 3) over
 * the 10-row reference window, LSTMCell(15 + 160 -> 8), Linear(8 -> 4).
 * Device pointers to the plain row-major torch parameters (the kernels
 * gather them into matrix-core operand order themselves). */
typedef struct ApgLstmPolicy {
  const float *conv_w;  /* [20][9][3] conv_ref.weight */
  const float *conv_b;  /* [20]       conv_ref.bias   */
  const float *w_ih;    /* [32][175]  lstm.weight_ih (gate order i,f,g,o) */
  const float *w_hh;    /* [32][8]    lstm.weight_hh */
  const float *b_ih;    /* [32]       lstm.bias_ih */
  const float *b_hh;    /* [32]       lstm.bias_hh */
  const float *w_out;   /* [4][8]     fc_out.weight */
  const float *b_out;   /* [4]        fc_out.bias */
} ApgLstmPolicy;

/* Fused LSTM-mode unroll, forward sweep: replaces the loop of
 * TrainDrone.train_recurrent_model (scripts/train_drone.py:134-157) for
 * train_mode "LSTM" - per step: window made relative to the current position
 * (copied), state_preprocessing, LSTM_NEW.forward, sigmoid,
 * FlightmareDynamics - all H = 10 steps in one launch, gate projection and
 * conv on the matrix cores (one wave = 32 trajectories), hidden / cell state
 * and the sliding window in registers.
 * SoA only:  state0 [12][B], in_ref [2H][9][B], h0 / c0 [8][B].
 * Outputs (caller-allocated, N = H*B, plane index = step*B + trajectory):
 *   states [H][12][B], actions [H][4][B],
 *   x [15][N]    the state features (LSTM inputs 0..14; round 6: the 160
 *                relu(conv) inputs are NOT stored - apg_quad_lstm_gate_wgrad
 *                recomputes them from the window for the weight gradient),
 *   gates [32][N] activated gates (i, f, g, o), hc [16][N] = h_prev, c_prev,
 *   hnew [8][N], relu_mask [5][N] (bit ch*8+pos = conv output > 0).
 * `workspace`: apg_quad_lstm_workspace_floats() floats of scratch (weights in
 * operand order); forward and reverse call may share it (same stream). */
int apg_quad_lstm_workspace_floats(void);
int apg_quad_lstm_rollout_fwd(const float *state0, const float *in_ref,
                              const float *h0, const float *c0, float dt,
                              const ApgQuadParams *params,
                              const ApgLstmPolicy *policy, int B, int H,
                              float *states, float *actions, float *x,
                              float *gates, float *hc, float *hnew,
                              unsigned *relu_mask, float *workspace,
                              apg_stream_t stream);

/* SURVEY.md 8a A4 `legacy_inplace_ref`: the same forward sweep with the loop of
 * scripts/train_drone.py:138-142 AS SHIPPED - the reference window is a view of
 * the batch and the relative-position subtraction writes through it, so every
 * step shifts the rows its window holds AGAIN.  Forward only (the reference
 * cannot back-propagate through the in-place write; there is no gradient to
 * match); in_ref itself is not modified.  apg_quad_mlp_rollout_fwd_inplace_ref is
 * the autoregressive-MLP counterpart. */
int apg_quad_lstm_rollout_fwd_inplace_ref(const float *state0, const float *in_ref,
                                          const float *h0, const float *c0, float dt,
                                          const ApgQuadParams *params,
                                          const ApgLstmPolicy *policy, int B, int H,
                                          float *states, float *actions, float *x,
                                          float *gates, float *hc, float *hnew,
                                          unsigned *relu_mask, float *workspace,
                                          apg_stream_t stream);

/* Reverse sweep (BPTT) of the above + quad_mpc_loss on ref[:, :H]
 * (scripts/train_drone.py:159-168).  ref [H][ref_cols][B].  Writes the loss
 * (loss_partials: apg_quad_lstm_loss_partials_count(B) floats) and the
 * cotangent planes from which the host forms the weight gradients:
 *   d_gates [32][N] (gate pre-activations), d_zout [4][N] (head
 *   pre-activations), d_conv [720][B]: the conv pre-activation cotangents
 *   d[ch][pos][k] summed along the window diagonals - the window of (step k,
 *   position pos, tap t) is reference row k + pos + t, so 17 diagonal sums per
 *   channel carry what 80 (pos, k) planes would:
 *     planes [0, 520):   G[ch][hi][tau] = sum_{k + pos - 4 hi = tau} d[ch][pos][k]
 *                        (hi = pos / 4: the half-wave that holds the position;
 *                        tau in [0, 13)), plane = ch * 26 + hi * 13 + tau
 *     planes [520, 720): P[ch][k] = sum_pos d[ch][pos][k], plane = 520 + ch*10 + k
 * optional grad_state0 [12][B], grad_h0 / grad_c0 [8][B]; optional cot_amax
 * [apg_quad_lstm_cot_amax_floats(B)] = per group of 32 trajectories the largest
 * |d_gates| and |d_zout| of the unroll (apg_quad_lstm_gate_wgrad scales its fp16
 * operands by them).
 *   dW_ih = d_gates x^T, dW_hh = d_gates h_prev^T, db_ih = db_hh = sum d_gates,
 *   dW_out = d_zout hnew^T, db_out = sum d_zout  (apg_quad_lstm_gate_wgrad),
 *   dconv_w[ch][c][t] = sum_{hi,tau,n} G[ch][hi][tau][n] ref[n][4 hi + tau + t][c]
 *                       - (c < 3) sum_{k,n} P[ch][k][n] pos_k[n][c],
 *   dconv_b[ch] = sum_{k,n} P[ch][k][n]. */
int apg_quad_lstm_loss_partials_count(int B);
int apg_quad_lstm_cot_amax_floats(int B);
int apg_quad_lstm_rollout_bwd(const float *state0, const float *states,
                              const float *actions, const float *ref,
                              int ref_cols, const unsigned *relu_mask,
                              const float *gates, const float *hc, float dt,
                              const ApgQuadParams *params,
                              const ApgQuadLossWeights *weights,
                              const ApgLstmPolicy *policy, int B, int H,
                              float *loss_partials, float *loss, float *d_gates,
                              float *d_zout, float *d_conv, float *grad_state0,
                              float *grad_h0, float *grad_c0, float *cot_amax,
                              float *workspace, apg_stream_t stream);

/* Round 6: the gate and head weight gradients of the LSTM unroll from the
 * cotangent planes of the reverse sweep - what `loss.backward()` leaves in
 * lstm.weight_ih / weight_hh / bias_ih and fc_out.weight / bias
 * (scripts/train_base.py:200-204 for train_mode "LSTM") - without the conv
 * outputs in memory: one kernel recomputes relu(conv(window)) per step on the
 * matrix cores (operands swapped: trajectories in the registers) and multiplies
 * it, the stored features / h_prev and a ones column with d_gates, h_new and
 * ones with d_zout, trajectory-major (one wave per SIMD: all eight window
 * positions of 64 trajectories); a second one sums the workgroups in index order
 * (bit-reproducible).
 *   state0 [12][B], states [H][12][B], in_ref [2H][9][B] as for the sweeps;
 *   acts [39][N] = x (15) | hc (16) | hnew (8) of apg_quad_lstm_rollout_fwd as
 *   ONE buffer; d_gates [32][N], d_zout [4][N], cot_amax of
 *   apg_quad_lstm_rollout_bwd.
 *   `policy` given: its forward tables are packed into `tables_fwd`
 *   (apg_quad_lstm_workspace_floats() floats) first; NULL: `tables_fwd` holds them
 *   (apg_quad_lstm_pack_tables / apg_quad_lstm_step_tail).
 *   partials: apg_quad_lstm_gate_wgrad_partials_floats(B) floats of scratch.
 * Outputs: ih_hh [32][183] = [dW_ih | dW_hh], b_ih [32] (= db_hh),
 *   w_out [4][8], b_out [4]. */
int apg_quad_lstm_gate_wgrad_partials_floats(int B);
int apg_quad_lstm_gate_wgrad(const float *state0, const float *states, const float *in_ref,
                             const float *acts, const float *d_gates, const float *d_zout,
                             const float *cot_amax, const ApgLstmPolicy *policy,
                             float *tables_fwd, int B, int H,
                             float *partials, float *ih_hh, float *b_ih, float *w_out,
                             float *b_out, apg_stream_t stream);

/* Round 6: the conv weights' gradient of the LSTM unroll (what `loss.backward()`
 * leaves in conv_ref.weight / bias, scripts/train_drone.py:168) from the diagonal
 * sums d_conv [720][B] of apg_quad_lstm_rollout_bwd, the reference planes in_ref
 * [2H][9][B] and st_all [(H + 1)][12][B] = [state0; states]: one kernel, both
 * operands of a segment straight out of planes, trajectory-major products on
 * fp16-split operands; a second one sums the workgroups in index order.
 *   conv_w [20][27] = sum G . R (index c * 3 + t: conv_ref.weight's own layout,
 *   BEFORE the position part is subtracted), conv_pos [20][3] = sum_k P . pos_k,
 *   conv_b [20] = sum_k P;  dconv_w[ch][c][t] = conv_w[ch][c][t] - (c < 3) conv_pos[ch][c]
 *   (apg_quad_lstm_step_tail does the subtraction).
 *   partials: apg_quad_lstm_conv_wgrad_partials_floats(B) floats of scratch. */
int apg_quad_lstm_conv_wgrad_partials_floats(int B);
int apg_quad_lstm_conv_wgrad(const float *d_conv, const float *in_ref, const float *st_all, int B,
                             int H, float *partials, float *conv_w, float *conv_pos,
                             float *conv_b, apg_stream_t stream);

/* Both weight-gradient products of the LSTM unroll with ONE sum launch behind them
 * (what the training step calls): apg_quad_lstm_gate_wgrad's and
 * apg_quad_lstm_conv_wgrad's product kernels back to back, then one kernel that
 * adds up both sets of partials side by side (the same fixed orders: results are
 * bit-identical to the two calls).  Arguments as there; gate_partials /
 * conv_partials: the two scratch buffers.
 *   finish (or NULL): the step's ApgLstmStepTail (declared below).  The thread
 *   that holds a gradient element's final sum then finishes the step for it: the
 *   element into finish->grad (conv_ref.weight minus its position part,
 *   lstm.weight_ih / weight_hh as their own tensors, bias_hh = bias_ih's) and - if
 *   finish->update - momentum SGD on its parameter, the arithmetic of
 *   apg_quad_lstm_step_tail; call that with `applied = 1` afterwards. */
struct ApgLstmStepTail;
int apg_quad_lstm_wgrads(const float *state0, const float *states, const float *in_ref,
                         const float *acts, const float *d_gates, const float *d_zout,
                         const float *cot_amax, const float *d_conv, const float *st_all,
                         const ApgLstmPolicy *policy, float *tables_fwd, int B, int H,
                         float *gate_partials, float *conv_partials, float *ih_hh, float *b_ih,
                         float *w_out, float *b_out, float *conv_w, float *conv_pos,
                         float *conv_b, const struct ApgLstmStepTail *finish,
                         apg_stream_t stream);

/* The LSTM training step without its small launches (round 6; the loop body of
 * scripts/train_base.py:198-214 for train_mode "LSTM": loss.backward() +
 * optimizer.step()).  The operand tables of the two sweeps live in caller-owned
 * buffers (apg_quad_lstm_tables_floats(0 | 1) floats) that apg_quad_lstm_pack_tables
 * fills from the parameters in ONE launch and apg_quad_lstm_step_tail refreshes
 * after its update: the `_packed` sweeps take the tables instead of the
 * parameters and launch nothing but themselves (loss may be NULL: the tail
 * reduces the partials). */
int apg_quad_lstm_tables_floats(int reverse);
int apg_quad_lstm_pack_tables(const ApgLstmPolicy *policy, float *tables_fwd,
                              float *tables_bwd, apg_stream_t stream);
int apg_quad_lstm_rollout_fwd_packed(const float *state0, const float *in_ref,
                                     const float *h0, const float *c0, float dt,
                                     const ApgQuadParams *params, const float *tables_fwd,
                                     int B, int H, float *states, float *actions, float *x,
                                     float *gates, float *hc, float *hnew,
                                     unsigned *relu_mask, apg_stream_t stream);
int apg_quad_lstm_rollout_bwd_packed(const float *state0, const float *states,
                                     const float *actions, const float *ref, int ref_cols,
                                     const unsigned *relu_mask, const float *gates,
                                     const float *hc, float dt, const ApgQuadParams *params,
                                     const ApgQuadLossWeights *weights,
                                     const float *tables_bwd, int B, int H,
                                     float *loss_partials, float *loss, float *d_gates,
                                     float *d_zout, float *d_conv, float *grad_state0,
                                     float *grad_h0, float *grad_c0, float *cot_amax,
                                     apg_stream_t stream);
/* What follows the weight-gradient products of the step, in one launch of one
 * workgroup: the gradients into their tensors (conv_ref.weight = grad.conv_w as
 * the window product left it minus the position part conv_pos [20][3];
 * lstm.weight_ih / weight_hh out of ih_hh [32][183]; lstm.bias_hh = bias_ih's;
 * the others are where the products wrote them), then - if `update` - momentum
 * SGD on the eight tensors with torch's arithmetic (buf = momentum buf + grad,
 * p -= lr buf, each in double with one rounding), the tables of the next step's
 * sweeps from the updated parameters (tables_fwd / tables_bwd, both or none),
 * and the loss (loss = sum of n_partials floats; loss_sum += loss if given). */
typedef struct ApgLstmPolicyGrads {
  float *conv_w, *conv_b, *w_ih, *w_hh, *b_ih, *b_hh, *w_out, *b_out;
} ApgLstmPolicyGrads;
typedef struct ApgLstmStepTail {
  ApgLstmPolicyGrads grad;      /* b_hh may alias b_ih */
  const float *ih_hh;           /* [32][183] = [dW_ih | dW_hh] */
  const float *conv_pos;        /* [20][3] */
  int update;
  double lr, momentum;
  ApgLstmPolicyGrads param, mom;
  float *tables_fwd, *tables_bwd;
  const float *loss_partials;
  int n_partials;
  float *loss, *loss_sum;
  /* 1: apg_quad_lstm_wgrads(finish = this) already put the gradients in place and
   * applied the update: the tail packs the tables and sums the loss, nothing else
   * (on ~60 workgroups instead of one) */
  int applied;
} ApgLstmStepTail;
int apg_quad_lstm_step_tail(const ApgLstmStepTail *tail, apg_stream_t stream);

/* ------------------------------------------- quad, MLP policy in-kernel ---- */
/* hutter_model.Net with the conv branch (neural_control/models/
 * hutter_model.py:6-49) for state_dim 15, horizon 10, ref_dim 9, 4 actions:
 * Linear(15 -> 64), conv1d(9 -> 20, k = 3), fc1 (224 -> 64), fc2, fc3
 * (64 -> 64), fc_out (64 -> 4).  Device pointers to the plain row-major
 * torch parameters (the kernels gather them into matrix-core operand order
 * themselves). */
typedef struct ApgMlpPolicy {
  const float *w_s;    /* [64][15]   states_in.weight */
  const float *b_s;    /* [64] */
  const float *conv_w; /* [20][9][3] conv_ref.weight */
  const float *conv_b; /* [20] */
  const float *w_1;    /* [64][224]  fc1.weight (inputs: s1, then conv ch-major) */
  const float *b_1;    /* [64] */
  const float *w_2;    /* [64][64]   fc2.weight */
  const float *b_2;    /* [64] */
  const float *w_3;    /* [64][64]   fc3.weight */
  const float *b_3;    /* [64] */
  const float *w_out;  /* [4][64]    fc_out.weight */
  const float *b_out;  /* [4] */
} ApgMlpPolicy;

/* Fused AUTOREGRESSIVE unroll, forward sweep: the loop of
 * TrainDrone.train_recurrent_model (scripts/train_drone.py:134-157) for
 * train_mode "autoregressive" - per step: window relative to the current
 * position (copied), state_preprocessing, Net.forward,
 * sigmoid, FlightmareDynamics.  SoA only: state0 [12][B], in_ref [2H][9][B].
 * Outputs (N = H*B, plane index = step*B + trajectory):
 *   states [H][12][B], actions [H][4][B], feat [15][N],
 *   x1 [224][N] (tanh state branch, relu conv), h [192][N] (h1, h2, h3),
 *   relu_mask [5][N].
 * `workspace`: apg_quad_mlp_workspace_floats() floats of scratch (the weights
 * re-ordered into matrix-core operand order by a small pre-kernel); the
 * forward and the reverse call may share it (same stream).  B <= 419 430. */
int apg_quad_mlp_workspace_floats(void);
int apg_quad_mlp_rollout_fwd(const float *state0, const float *in_ref, float dt,
                             const ApgQuadParams *params,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *states, float *actions, float *feat,
                             float *x1, float *h, unsigned *relu_mask,
                             float *workspace, apg_stream_t stream);
/* (forward only: the loop as shipped, see apg_quad_lstm_rollout_fwd_inplace_ref) */
int apg_quad_mlp_rollout_fwd_inplace_ref(const float *state0, const float *in_ref, float dt,
                                         const ApgQuadParams *params,
                                         const ApgMlpPolicy *policy, int B, int H,
                                         float *states, float *actions, float *feat,
                                         float *x1, float *h, unsigned *relu_mask,
                                         float *workspace, apg_stream_t stream);

/* apg_quad_mlp_loss_partials_count(B): floats of loss partials of the MLP-policy
 * sweeps.  (The plane-writing reverse sweep of rounds 1-4, apg_quad_mlp_rollout_bwd,
 * and the plane form of the concurrent step, apg_quad_mlp_concurrent_fwd_bwd, are
 * in include/apg_planes.h / libapg_planes.so since round 6: a test library - the
 * product's reverse sweeps accumulate the weight gradients themselves:
 * apg_quad_mlp_rollout_train_step, apg_quad_mlp_concurrent_step below.) */
int apg_quad_mlp_loss_partials_count(int B);

/* Concurrent-mode training step with the policy inside (BASELINE config 2):
 * TrainBase.run_epoch's concurrent branch (scripts/train_base.py:198-204:
 * actions = sigmoid(net(in_state, in_ref)) reshaped [B, H, 4]) +
 * TrainDrone.train_controller_model (scripts/train_drone.py:175-203); `policy` is a
 * Net(15, 10, 9, 40, conv=1): w_out [40][64], b_out [40].  Round 4: the weight
 * gradients are accumulated INSIDE the
 * reverse pass (no cotangent planes, no second pass of products): pack, forward
 * + rollout + adjoint, reverse pass with the products of every layer on the
 * matrix cores (cotangents transposed through LDS, activations read from the
 * forward kernel's planes), one fixed-order second stage that also sums the
 * loss.  One loss.backward() of the reference produces every parameter
 * gradient (scripts/train_drone.py:175-203); so does this call.
 *   acts [521][B]: planes 0..14 = feat (in), 431..520 = in_ref [H][9] (in), the
 *     224 + 192 planes between them are written (x1, h1, h2, h3);
 *   state0 [12][B], ref [H][ref_cols][B] (in); relu_mask [5][B], d_zout [40][B]
 *     (scratch); loss_partials (apg_quad_mlp_loss_partials_count(B)), loss [1]
 *     or NULL, states [H][12][B] or NULL;
 *   grads: where each parameter's gradient goes (12 device pointers, shapes of
 *     the policy's tensors; typically views of one flat buffer);
 *   workspace: apg_quad_mlp_step_workspace_floats() floats,
 *   partials:  apg_quad_mlp_step_partials_floats(B) floats;
 *   after_reverse: optional hipEvent_t recorded on `stream` once the reverse
 *     kernel is enqueued - the last reader of acts / state0 / ref; a caller
 *     that pipelines batches refills the next batch's buffers behind it while
 *     the second stage and the optimizer run (NULL: none). */
typedef struct ApgMlpPolicyGrads {
  float *w_s, *b_s;        /* [64][15], [64]  */
  float *conv_w, *conv_b;  /* [20][9][3], [20] */
  float *w_1, *b_1;        /* [64][224], [64] */
  float *w_2, *b_2, *w_3, *b_3; /* [64][64], [64] */
  float *w_out, *b_out;    /* [40][64], [40]  */
} ApgMlpPolicyGrads;
int apg_quad_mlp_step_workspace_floats(void);
long long apg_quad_mlp_step_partials_floats(int B);
/* The reverse kernel of the two step calls below (csrc/mlp_concurrent.hip,
 * mlp_concurrent_bwd_tm_kernel): every wave multiplies its own 32 trajectories'
 * cotangents - obtained in [trajectory][feature] form by issuing the layer's
 * matrix instructions with the operands swapped - against its own x and adds
 * the 32 x 32 blocks into 32-bit FIXED-POINT accumulators in LDS (ds_add_u32:
 * order-independent, so results are bit-reproducible); one barrier per layer.
 * Round 5: the chain's operands are scaled per trajectory, the head's rows
 * have their own exponents, the biases are per-wave float sums.  A non-finite
 * cotangent or activation yields NaN gradients.  Nothing in the environment
 * influences the kernel.  (Round 4's staged kernel: tools/patches/.) */
int apg_quad_mlp_concurrent_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, apg_event_t after_reverse,
    apg_stream_t stream);

/* The same call with the optimizer inside: torch.optim.SGD(lr, momentum = 0.9)
 * as the reference builds it (scripts/train_base.py:140-143) and applies it
 * after loss.backward() (scripts/train_drone.py:200-203),
 *     buf = momentum * buf + grad;   param -= lr * buf
 * done by the second stage's threads right where they have summed a gradient
 * element (one launch and one pass over the parameters less per step).
 * `update->param` are the tensors `policy` points at (the pack launch at the
 * head of the call has read them before they change), `update->momentum_buf`
 * the optimizer's momentum buffers (all zero before the first step).  The
 * gradients are still written to `grads`.  update NULL: no update, exactly
 * apg_quad_mlp_concurrent_step.  Single-process training only: with more than
 * one rank the gradients must be all-reduced BEFORE the update - call the
 * plain step, reduce, then step the optimizer. */
/* Stream events of a pipelined caller (each may be NULL; hipEvent_t):
 *   inputs_ready   waited on (hipStreamWaitEvent) after the pack launch, before
 *                  the forward kernel - the first reader of acts / state0 / ref:
 *                  the producer of these buffers (a gather on another stream)
 *                  may still be running while the tables are packed;
 *   after_forward  recorded once the forward kernel is enqueued;
 *   after_reverse  recorded once the reverse kernel - the last reader of the
 *                  inputs - is enqueued (as apg_quad_mlp_concurrent_step's). */
typedef struct ApgStepEvents {
  apg_event_t inputs_ready, after_forward, after_reverse;
} ApgStepEvents;
/* resident (apg_quad_mlp_concurrent_train_step[_rows] only; 0 elsewhere): the
 * kernels read the policy from packed operand tables in `workspace`, normally
 * re-packed from the parameters by a launch at the head of every call.  A
 * caller that keeps the SAME workspace from call to call and changes the
 * parameters through this update only may let the second stage keep the tables
 * current instead: 1 - pack now, build the table map (first call on a
 * workspace), scatter the updated parameters into the tables; 2 - the tables of
 * the previous call are current (no parameter was written by anyone else
 * since): no pack launch; 3 - pack now (somebody else wrote the parameters), the
 * workspace's map is still the one of an earlier call with 1.  0: pack every
 * call, leave nothing behind. */
typedef struct ApgMlpSgdUpdate {
  double lr, momentum;     /* (torch's fused SGD computes in double, rounds once) */
  ApgMlpPolicyGrads param;
  ApgMlpPolicyGrads momentum_buf;
  int resident;
} ApgMlpSgdUpdate;
int apg_quad_mlp_concurrent_train_step(
    const float *state0, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *acts, unsigned *relu_mask,
    float *d_zout, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *states, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    const ApgStepEvents *events, apg_stream_t stream);

/* The same step with the minibatch selection inside (round 5; the batch
 * selection of TrainBase.run_epoch, scripts/train_base.py:191-194, without a
 * gather pass): the forward kernel reads its trajectories' rows of the DATA SET's
 * tensors through `index` - normed [N][ld_normed] (15 columns read), state0
 * [N][ld_state0] (12), in_ref [N][ld_in_ref] (90), ref [N][ld_ref] (H x
 * ref_cols), float32 device memory, row strides in floats, every tensor below
 * 4 GiB; the reverse kernel reads the features and windows it multiplies with
 * from the same rows.  The feature / window planes of `acts` (0..14, 431..520)
 * are not used in this form: the caller fills nothing.  index [B]: int64 row
 * numbers in [0, n_rows), device memory.
 * running_loss (or NULL): one device float the step's loss is ADDED to - the
 * epoch loop's `running_loss += loss` (scripts/train_base.py:212) without a
 * launch of its own; needs `loss`. */
typedef struct ApgBatchRows {
  const long long *index;
  const float *normed, *state0, *in_ref, *ref;
  int ld_normed, ld_state0, ld_in_ref, ld_ref;
  long long n_rows;
  float *running_loss;
} ApgBatchRows;
int apg_quad_mlp_concurrent_train_step_rows(
    const ApgBatchRows *rows, int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H,
    float *acts, unsigned *relu_mask, float *d_zout, float *loss_partials, float *loss,
    const ApgMlpPolicyGrads *grads, float *states, float *workspace, float *partials,
    const ApgMlpSgdUpdate *update, const ApgStepEvents *events, apg_stream_t stream);

/* Round 6: the LSTM sweeps on a minibatch named by ROW NUMBERS of the whole data
 * set's tensors (TrainBase.run_epoch's batch selection,
 * scripts/train_base.py:191-194: `batch = data[index]`) - no gather pass.  The
 * forward sweep reads rows->state0 [n][ld >= 12] and rows->in_ref [n][ld >= 2H x 9]
 * through rows->index [B] (range-checked: a row number beyond n_rows reads zeros)
 * and WRITES their planes state0 [12][B], in_ref [2H][9][B] for its followers
 * (the reverse sweep, apg_quad_lstm_gate_wgrad, the conv-weight products); the
 * reverse sweep reads rows->ref [n][ld >= H x ref_cols] the same way.  Packed
 * tables only; everything else as apg_quad_lstm_rollout_fwd_packed / _bwd_packed.
 * (rows->normed, rows->running_loss are not used.) */
int apg_quad_lstm_rollout_fwd_rows(const ApgBatchRows *rows, const float *h0, const float *c0,
                                   float dt, const ApgQuadParams *params,
                                   const float *tables_fwd, int B, int H, float *state0,
                                   float *in_ref, float *states, float *actions, float *x,
                                   float *gates, float *hc, float *hnew, unsigned *relu_mask,
                                   apg_stream_t stream);
int apg_quad_lstm_rollout_bwd_rows(const ApgBatchRows *rows, int ref_cols, const float *state0,
                                   const float *states, const float *actions,
                                   const unsigned *relu_mask, const float *gates,
                                   const float *hc, float dt, const ApgQuadParams *params,
                                   const ApgQuadLossWeights *weights, const float *tables_bwd,
                                   int B, int H, float *loss_partials, float *loss,
                                   float *d_gates, float *d_zout, float *d_conv,
                                   float *grad_state0, float *grad_h0, float *grad_c0,
                                   float *cot_amax, apg_stream_t stream);

/* The AUTOREGRESSIVE training step in one call (round 5; configs[2] per rank):
 * TrainDrone.train_recurrent_model's unroll, loss and loss.backward()
 * (scripts/train_drone.py:113-173) for Net(15, 10, 9, 4, conv=1) - the forward
 * sweep (as apg_quad_mlp_rollout_fwd), then a reverse sweep that accumulates
 * EVERY parameter gradient inside (trajectory-major block products per step
 * and layer into 32-bit fixed-point LDS accumulators, flushed per phase into a
 * workgroup-owned float accumulator in `partials`; csrc/mlp_rollout.hip,
 * mlp_rollout_bwd_tm_kernel), then the fixed-order second stage of
 * apg_quad_mlp_concurrent_step.  No cotangent planes, no apg_planes_gemm.
 * Bit-reproducible run to run.
 *   state0 [12][B], in_ref [2H][9][B], ref [H][ref_cols][B] (planes);
 *   out: states [H][12][B], actions [H][4][B], acts [431][H*B] (feat | x1 |
 *     h1 h2 h3: scratch the reverse sweep reads), relu_mask [5][H*B] (scratch),
 *     loss_partials (apg_quad_mlp_loss_partials_count(B)), loss [1],
 *     grads (every pointer set; fc_out is [4][64]), grad_state0 [12][B] or NULL;
 *   workspace: apg_quad_mlp_rollout_step_workspace_floats() floats,
 *   partials:  apg_quad_mlp_rollout_step_partials_floats(B) floats;
 *   update: as apg_quad_mlp_concurrent_train_step (NULL: gradients only).
 * Operand range: as the other in-kernel policies (finite, |x| < 2^14). */
int apg_quad_mlp_rollout_step_workspace_floats(void);
long long apg_quad_mlp_rollout_step_partials_floats(int B);
int apg_quad_mlp_rollout_train_step(
    const float *state0, const float *in_ref, const float *ref, int ref_cols, float dt,
    const ApgQuadParams *params, const ApgQuadLossWeights *weights,
    const ApgMlpPolicy *policy, int B, int H, float *states, float *actions, float *acts,
    unsigned *relu_mask, float *loss_partials, float *loss, const ApgMlpPolicyGrads *grads,
    float *grad_state0, float *workspace, float *partials, const ApgMlpSgdUpdate *update,
    apg_stream_t stream);

/* Batched closed-loop evaluation: the loop of
 * QuadEvaluator.follow_trajectory("rand") (scripts/evaluate_drone.py:81-194)
 * for B reference trajectories in one launch - per step the H-row reference
 * window (Random.get_ref_traj, neural_control/trajectory/random_traj.py:60-79),
 * QuadDataset.prepare_data (dataset.py:155-204), the policy (first 4 outputs of
 * `policy->w_out`, so a concurrent-mode Net works unchanged), sigmoid, clip,
 * FlightmareDynamics, the divergence from the projected reference and the
 * attitude check (drone_env.py:59-72); on failure either stop (test_time) or
 * reset to the reference state.  traj [L][9][B] = (position, euler, velocity)
 * rows, L > H.  T = min(max_steps, L + 1) iterations.
 * Outputs: div [T][B] (rows >= steps[b] are not written), steps [B];
 * optional (NULL to skip) drone [T+1][12][B] (state after each step, row 0 =
 * start), actions [T][4][B], start_states [T][12][B] (state the policy saw,
 * i.e. after a reset).  workspace: apg_quad_mlp_workspace_floats(). */
int apg_quad_mlp_closed_loop(const float *traj, int L, float dt,
                             const ApgQuadParams *params,
                             const ApgMlpPolicy *policy, int B, int H,
                             int max_steps, float thresh_div,
                             float thresh_stable, int test_time, float *div,
                             int *steps, float *drone, float *actions,
                             float *start_states, float *workspace,
                             apg_stream_t stream);
/* ... with the environment a caller chooses: `learnt` NULL = FlightmareDynamics
 * (the function above), else the LEARNT simulator of train_dynamics()
 * (LearntDynamics, neural_control/dynamics/quad_dynamics_trained.py:10-69:
 * action transform, the analytic step on `params`, the relu residual) - what
 * QuadEvaluator flies after scripts/train_drone.py:44-45 swapped
 * eval_env.dynamics for the train dynamics. */
int apg_quad_mlp_closed_loop_env(const float *traj, int L, float dt,
                                 const ApgQuadParams *params,
                                 const ApgLearntResidual *learnt,
                                 const ApgMlpPolicy *policy, int B, int H,
                                 int max_steps, float thresh_div,
                                 float thresh_stable, int test_time, float *div,
                                 int *steps, float *drone, float *actions,
                                 float *start_states, float *workspace,
                                 apg_stream_t stream);

/* The same closed loop for the LSTM controller (LSTM_NEW): hidden / cell
 * state h0 / c0 [8][B] carried through all steps (QuadEvaluator resets it once,
 * scripts/evaluate_drone.py:56-58).  workspace:
 * apg_quad_lstm_workspace_floats(). */
int apg_quad_lstm_closed_loop(const float *traj, int L, const float *h0,
                              const float *c0, float dt,
                              const ApgQuadParams *params,
                              const ApgLstmPolicy *policy, int B, int H,
                              int max_steps, float thresh_div,
                              float thresh_stable, int test_time, float *div,
                              int *steps, float *drone, float *actions,
                              float *start_states, float *workspace,
                              apg_stream_t stream);
int apg_quad_lstm_closed_loop_env(const float *traj, int L, const float *h0,
                                  const float *c0, float dt,
                                  const ApgQuadParams *params,
                                  const ApgLearntResidual *learnt,
                                  const ApgLstmPolicy *policy, int B, int H,
                                  int max_steps, float thresh_div,
                                  float thresh_stable, int test_time, float *div,
                                  int *steps, float *drone, float *actions,
                                  float *start_states, float *workspace,
                                  apg_stream_t stream);

/* "Planes x planes" reduction GEMM on the matrix cores (fp32 accuracy):
 *   C[m*ldc + j] = sum_{s<S} sum_{n<N} A[(m*S + s)*N + n] * B[bplane(j,s)*N + n]
 *   bplane(j, s) = bdesc[j] + (s / sdiv) * bdesc[J + j] + (s % sdiv) * bdesc[2J + j]
 * for m < M <= 64, j < J (J + with_ones <= 192, <= 128 when M > 32);
 * with_ones appends one column C[m][J] = sum_{s,n} A[...] (row sums).  Turns
 * the cotangent planes of apg_quad_lstm_rollout_bwd / apg_quad_mlp_rollout_bwd
 * into weight gradients - the role torch.autograd plays for the parameters in
 * scripts/train_drone.py:168.  The per-column two-level segment stride reads
 * the sliding reference windows of the conv branch in place (segment =
 * (window position, step), sdiv = H).  `bdesc` is a DEVICE int array [3][J];
 * `workspace` holds apg_planes_gemm_workspace_floats(M, J, with_ones, num_wg)
 * floats; C has row stride ldc >= J + with_ones.  With `bias_out` (and
 * with_ones) the row sums go to bias_out[M] instead of column J, so that
 * weight and bias gradients are both contiguous (ldc >= J then).  B holds
 * `b_planes` planes; A (M*S planes) and B must each stay below 4 GiB (unsigned
 * 32-bit buffer offsets): hand B over from the first plane the product uses
 * (bdesc relative to it) when the tensor behind it is larger.
 * Two kernels behind it: plain products (S = 1; and segmented ones with J < 16)
 * stream their operands global ->
 * registers -> v_mfma_f32_16x16x32_bf16 on three-term bf16 splits of the
 * fp32 operands, six products per tile (split-K over all waves, no LDS tile);
 * the other segmented products (S > 1, the conv windows) go through LDS tiles filled by
 * direct-to-LDS DMA and v_mfma_f32_32x32x2_f32.  apg_planes_gemm_default_wgs: the num_wg measured best for
 * the shape (what apg_planes_gemm_multi uses). */
int apg_planes_gemm_workspace_floats(int M, int J, int with_ones, int num_wg);
int apg_planes_gemm_default_wgs(int M, int S, int J, int with_ones);
int apg_planes_gemm(const float *A, int M, int S, const float *B,
                    const int *bdesc, int J, int sdiv, int with_ones,
                    int b_planes, long long N, float *workspace, int num_wg,
                    float *C, int ldc, float *bias_out, apg_stream_t stream);

/* Several of the products above in one launch pair (all weight gradients of a
 * training step): n <= 8 problems, each M <= 64 and J + with_ones <= 128.
 * workspace: num_wg * 64 * 128 floats; num_wg >= n workgroups are divided
 * among the problems in proportion to the planes they stream. */
typedef struct ApgGemmProblem {
  const float *A;
  const float *B;
  const int *bdesc;   /* device int [3][J] */
  float *C;
  float *bias_out;    /* or NULL */
  long long N;
  int M, S, J, sdiv, with_ones, b_planes, ldc;
} ApgGemmProblem;
int apg_planes_gemm_grouped(const ApgGemmProblem *problems, int n,
                            float *workspace, int num_wg, apg_stream_t stream);

/* The same n <= 8 products as n launches, each with the tile shape and
 * occupancy that fit it (long planes: H columns per trajectory), followed by
 * ONE second-stage launch for all of them.  J + with_ones <= 192 (<= 128 when
 * M > 32) as for apg_planes_gemm.  workspace:
 * apg_planes_gemm_multi_workspace_floats(problems, n) floats. */
long long apg_planes_gemm_multi_workspace_floats(const ApgGemmProblem *problems,
                                                 int n);
int apg_planes_gemm_multi(const ApgGemmProblem *problems, int n, float *workspace,
                          apg_stream_t stream);

/* ---------------------------------------------------------- fixed wing --- */
/* Weight gradient of a PyTorch-side policy layer y = x W^T + b (where the
 * policy is not inside a kernel: scripts/train_base.py:198-209 ->
 * loss.backward() on torch.nn.Linear): dW[m][n] = sum_b dY[b][m] X[b][n] and,
 * if db is not NULL, db[m] = sum_b dY[b][m], for row-major dY [B, M], X [B, N]
 * (each below 4 GiB).  Split-K over the whole chip on the fp32 matrix
 * instruction, fixed summation order.  workspace:
 * apg_linear_wgrad_workspace_floats(M, N) device floats. */
long long apg_linear_wgrad_workspace_floats(int M, int N);
int apg_linear_wgrad(const float *dY, const float *X, long long B, int M, int N,
                     float *dW, float *db, float *workspace, apg_stream_t stream);

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

/* The fixed-wing controller hutter_model.Net(9, 1, 3, 80, conv=False)
 * (neural_control/models/hutter_model.py:6-49; scripts/train_fixed_wing.py:
 * 68-76): device pointers to the plain row-major torch parameters. */
typedef struct ApgWingPolicy {
  const float *w_s;    /* [64][9]   states_in.weight */
  const float *b_s;    /* [64] */
  const float *w_r;    /* [64][3]   ref_in.weight */
  const float *b_r;    /* [64] */
  const float *w_1;    /* [64][128] fc1.weight (inputs: state branch, ref branch) */
  const float *b_1;    /* [64] */
  const float *w_2;    /* [64][64]  fc2.weight */
  const float *b_2;    /* [64] */
  const float *w_3;    /* [64][64]  fc3.weight */
  const float *b_3;    /* [64] */
  const float *w_out;  /* [4 H][64] fc_out.weight */
  const float *b_out;  /* [4 H] */
} ApgWingPolicy;

/* Policy part of the fixed-wing concurrent step on the matrix cores
 * (scripts/train_base.py:198-204: actions = sigmoid(net(in_state,
 * in_ref_state))) for Net(9, 1, 3, 4 H, conv=False), H = 20 (BASELINE config
 * 4) or 10 (the horizon of the reference's shipped configs/wing_config.json;
 * w_out / b_out then have 40 rows).  Forward: feat [9][B] (normed_states),
 * ref_in [3][B] -> actions [4 H][B] = [H][4][B], the SoA action sequence of
 * apg_wing_rollout_fwd_bwd; saved x1 [128][B], h [192][B] (h1, h2, h3).
 * Reverse: from grad_actions [4 H][B] (that kernel's dL/dactions) -> d_zout
 * [4 H][B] and d_pre [320][B] (fc1, fc2, fc3 pre-activation cotangents, 64
 * planes each, then the first layer's 128); weight gradients by
 * apg_planes_gemm(_grouped).  workspace: apg_wing_policy_workspace_floats(). */
int apg_wing_policy_workspace_floats(void);
int apg_wing_policy_fwd(const float *feat, const float *ref_in,
                        const ApgWingPolicy *policy, int B, int H, float *actions,
                        float *x1, float *h, float *workspace,
                        apg_stream_t stream);
int apg_wing_policy_bwd(const float *actions, const float *grad_actions,
                        const float *x1, const float *h,
                        const ApgWingPolicy *policy, int B, int H, float *d_zout,
                        float *d_pre, float *workspace, apg_stream_t stream);

/* Closed-loop evaluation of the fixed-wing controller - beyond SURVEY.md §8:
 * FixedWingEvaluator.fly_to_point (scripts/evaluate_fixed_wing.py:45-131) for B
 * target lists in one launch.  Per step: WingDataset.prepare_data
 * (neural_control/dataset.py:322-350; mean / std: HOST arrays of 12, entries
 * 3..11 used; data_dt / data_horizon: the data set's dt and horizon, which fix
 * the length 12 * dt * horizon of the reference vector the policy sees),
 * FixedWingNetWrapper.predict_actions (neural_control/controllers/
 * network_wrapper.py:81-98: sigmoid, first action; only rows 0..3 of
 * policy->w_out / b_out are read, so a head of any horizon serves),
 * SimpleWingEnv.step (neural_control/environments/wing_env.py:44-57, dt),
 * project_to_line (neural_control/trajectory/q_funcs.py:6-18), the target
 * switch, and on divergence the break (test_time) or the reset onto the line
 * at 11.5 m/s.  targets [n_targets][3][B]; state0 [12][B] or NULL (zero_reset:
 * zeros, u = 11.5).  Outputs, all SoA: div_linear [T][B] (div_to_linear),
 * div_pass [T][B] / div_fail [T][B] = the values fly_to_point appends to
 * div_target when a target is passed / on divergence at that step (-1: none;
 * the caller appends thresh_div for a run with steps == max_steps), steps [B]
 * = len(drone_traj); optional drone [T][16][B] (state after the step, action)
 * and seen [T][15][B] (state the policy saw - after a reset still the last
 * simulated one, as in the reference - and its target).
 * workspace: apg_wing_policy_workspace_floats(). */
int apg_wing_mlp_closed_loop(const float *targets, int n_targets,
                             const float *state0, float dt,
                             const ApgWingParams *params,
                             const ApgWingPolicy *policy, const float *mean,
                             const float *std, float data_dt, int data_horizon,
                             int B, int max_steps, float thresh_div,
                             float thresh_stable, int test_time,
                             float *div_linear, float *div_pass, float *div_fail,
                             int *steps, float *drone, float *seen,
                             float *workspace, apg_stream_t stream);
/* ... with a LEARNT environment (LearntFixedWingDynamics.forward,
 * neural_control/dynamics/fixed_wing_dynamics.py:270-326 - what
 * SimpleWingEnv(train_dynamics) steps with after train_dynamics(),
 * scripts/train_fixed_wing.py:42-43): `params` = the CURRENT values of the
 * module's parameters, `inertia` = its 3x3 parameter `I` (HOST array of 9,
 * row-major, used in full, as apg_wing_learnt_step_fwd), `learnt` = the residual
 * network on [state, action] (linear_at is not read).  inertia and learnt both
 * NULL: the function above. */
int apg_wing_mlp_closed_loop_env(const float *targets, int n_targets,
                                 const float *state0, float dt,
                                 const ApgWingParams *params, const float *inertia,
                                 const ApgLearntResidual *learnt,
                                 const ApgWingPolicy *policy, const float *mean,
                                 const float *std, float data_dt, int data_horizon,
                                 int B, int max_steps, float thresh_div,
                                 float thresh_stable, int test_time,
                                 float *div_linear, float *div_pass, float *div_fail,
                                 int *steps, float *drone, float *seen,
                                 float *workspace, apg_stream_t stream);

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

/* The physics step of LearntFixedWingDynamics (beyond SURVEY.md §8;
 * neural_control/dynamics/fixed_wing_dynamics.py:270-326): simulate_fixed_wing
 * (:98-267) with the CURRENT values of its trainable parameters - `params` (the
 * I_* fields are ignored) and `inertia`, a HOST array of 9 = the 3x3 parameter
 * `I` row-major, used in full (no symmetry or sparsity assumed).  AoS
 * state [B,12] / action [B,4].  The reverse call returns dL/dstate, dL/daction
 * (either may be NULL) and grad_params: apg_wing_learnt_param_count() = 50
 * device floats, the batch-summed cotangents in the order of ApgWingParams'
 * 41 fields (I_xx..I_xz and g: 0 - the reference's weight g*mass is a
 * detached copy, :197, so mass gets its gradient through 1/mass only)
 * followed by dL/dI row-major.  workspace:
 * apg_wing_learnt_workspace_floats(B) device floats. */
int apg_wing_learnt_param_count(void);
int apg_wing_learnt_workspace_floats(int B);
int apg_wing_learnt_step_fwd(const float *state, const float *action, float dt,
                             const ApgWingParams *params, const float *inertia,
                             int B, float *next_state, apg_stream_t stream);
int apg_wing_learnt_step_bwd(const float *state, const float *action, float dt,
                             const ApgWingParams *params, const float *inertia,
                             int B, const float *grad_next, float *grad_state,
                             float *grad_action, float *grad_params,
                             float *workspace, apg_stream_t stream);

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
/* Test / measurement hook, process-wide: which of the two fused fixed-wing
 * kernels apg_wing_rollout_fwd_bwd launches for plane-layout batches.
 *   2 (default) two trajectories per lane once the batch exceeds one wave per
 *     SIMD, 1 whenever the batch is even and 8-byte aligned, 0 never.
 * Nothing in the environment influences the choice. */
int apg_wing_set_two_per_lane(int mode);

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

/* No-grad unroll (evaluation: CartPoleEnv._step in a loop,
 * neural_control/environments/cartpole_env.py:72-74): states_out[B,H,4]. */
int apg_cartpole_rollout_fwd(const float *state0, const float *actions, float dt,
                             const ApgCartpoleParams *params, int B, int H,
                             int layout, float *states_out, apg_stream_t stream);

/* --------------------------------------------------------------- misc --- */
/* loss[0] = fixed-order sum of partials[0..n) (one small kernel). */
int apg_reduce_loss_partials(const float *partials, int n, float *loss,
                             apg_stream_t stream);
/* Layout change at the boundary: src [B][R] row-major with row stride ld >= R
 * (the reference's batch-major tensors, trailing dims flattened into R; ld > R
 * reads the leading part of longer rows) -> dst [R][B], the plane layout every
 * APG_LAYOUT_SOA entry point reads.  With `index` (device int64 [B], or NULL)
 * row b of the result is src row index[b]: the minibatch gather of the
 * training loop (scripts/train_base.py:132-137, 191-194) in the same pass. */
int apg_to_soa(const float *src, const long long *index, int B, int R, int ld,
               float *dst, apg_stream_t stream);
/* The same for up to APG_SOA_MAX_ITEMS tensors of one batch in ONE launch (a
 * training step converts 3-5 of them: features, reference windows, state,
 * reference, hidden state). */
#define APG_SOA_MAX_ITEMS 6
typedef struct ApgSoaItem {
  const float *src;
  const long long *index;   /* device int64 [B] or NULL */
  float *dst;
  int R, ld;
} ApgSoaItem;
int apg_to_soa_multi(const ApgSoaItem *items, int n, int B, apg_stream_t stream);
/* Number of floats `loss_partials` must hold for a batch of B. */
int apg_loss_partials_count(int B);
/* Measurement aid: device-to-device stream copy (16-byte accesses, grid-
 * strided, non-temporal stores) - the copy bandwidth this GPU delivers, quoted
 * next to the 8 TB/s datasheet peak in bench.py's roofline.
 * 16-byte aligned pointers, bytes a multiple of 16. */
int apg_stream_copy(const void *src, void *dst, long long bytes, apg_stream_t stream);
/* The same copy through other launch shapes (0 = apg_stream_copy; 1 one element
 * per thread, whole-array grid; 2 capped grid, four elements in flight per
 * thread; 3 as 2 with non-temporal stores; 4 as 1 with 1 024-thread blocks):
 * measurement aid for bench.py's `roofline.copy_GBps_measured`. */
int apg_stream_copy_shape(const void *src, void *dst, long long bytes, int shape,
                          apg_stream_t stream);
/* Measurement aid for `roofline.stream_floor_us`: the bytes of ONE headline
 * launch (in_bytes read - state0 + actions + reference rows - and
 * out_bytes written - dL/dactions rows) moved in the fastest copy shape this GPU
 * has (one 16-byte element per thread, the whole input as the grid), with no
 * arithmetic.  shape 1 / 2: the first out_bytes/16 threads store (plain /
 * non-temporal); 3 / 4: the stores spread evenly between the loads; 5 / 6: as
 * 1 / 2 with one wave per workgroup; 7: the rollout's own pattern - one
 * trajectory per lane, one wave per workgroup, 28 row loads then 10 non-temporal
 * row stores of 16 bytes per lane (in_bytes = 28 x B x 16, out_bytes = 10 x B x
 * 16, B a multiple of 64; the shape check of shape 7 is not verified by stores
 * equal to loads: it writes sums). */
int apg_stream_rows_probe(const void *in, long long in_bytes, void *out, long long out_bytes,
                          int shape, apg_stream_t stream);

/* APG_VERSION_MAJOR * 1000 + APG_VERSION_MINOR. */
int apg_version(void);
/* Description of the last error on the calling thread ("" if none). */
const char *apg_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* APG_H_ */
