/* apg_planes.h - C ABI of libapg_planes.so, a TEST library (round 6).
 *
 * The reverse kernels of rounds 1-4 that write COTANGENT PLANES for planes_gemm
 * products - the autoregressive sweep (scripts/train_drone.py:159-168) and the
 * concurrent step (scripts/train_base.py:198-204 + scripts/train_drone.py:
 * 175-203) - behind the entry points they always had.  The product library
 * (include/apg.h, libapg_hip.so) has ONE reverse kernel per training mode, with
 * the weight gradients accumulated inside the sweep; these stay as an independent
 * implementation of the same sums (exact float accumulation in the products) for
 * tests/plane_path.py, the per-row arbiter and the in-sweep tests.  Built by
 * apg_trajectory_tracking_amd/build.py next to libapg_cpu.so; the package never
 * loads it.  Types, layouts, error convention: include/apg.h. */
#ifndef APG_PLANES_H_
#define APG_PLANES_H_
#include "apg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Reverse sweep of the above + quad_mpc_loss on ref[:, :H]
 * (scripts/train_drone.py:159-168).  loss_partials:
 * apg_quad_mlp_loss_partials_count(B) floats.
 * Cotangent planes for the weight gradients (apg_planes_gemm):
 *   d_pre [256][N] = pre-activation cotangents of fc1, fc2, fc3, states_in
 *   (64 planes each, in this order), d_zout [4][N], d_conv [720][B] (the
 *   window-diagonal sums G / P described at apg_quad_lstm_rollout_bwd);
 *   dW_1 = d_pre1 x1^T, dW_2 = d_pre2 h1^T, dW_3 = d_pre3 h2^T,
 *   dW_s = d_pre_s feat^T, dW_out = d_zout h3^T, biases = row sums,
 *   dconv_w as for the LSTM policy.  Optional grad_state0 [12][B]. */
int apg_quad_mlp_rollout_bwd(const float *state0, const float *states,
                             const float *actions, const float *ref,
                             int ref_cols, const float *x1, const float *h,
                             const unsigned *relu_mask, float dt,
                             const ApgQuadParams *params,
                             const ApgQuadLossWeights *weights,
                             const ApgMlpPolicy *policy, int B, int H,
                             float *loss_partials, float *loss, float *d_pre,
                             float *d_zout, float *d_conv, float *grad_state0,
                             float *workspace, apg_stream_t stream);


/* Concurrent-mode training step with the policy inside (BASELINE config 2):
 * TrainBase.run_epoch's concurrent branch (scripts/train_base.py:198-204:
 * actions = sigmoid(net(in_state, in_ref)) reshaped [B, H, 4]) +
 * TrainDrone.train_controller_model (scripts/train_drone.py:175-203) in two
 * launches: network forward once per trajectory + register-resident rollout,
 * quad_mpc_loss and adjoint; then the network's reverse pass.  `policy` is a
 * Net(15, 10, 9, 40, conv=1): w_out [40][64], b_out [40].
 * In (SoA): feat [15][B] (the data set's normed_states), in_ref [H][9][B],
 * state0 [12][B], ref [H][ref_cols][B].
 * Out: planes for apg_planes_gemm - x1 [224][B], h [192][B], relu_mask [5][B],
 * d_zout [40][B] (head pre-activation cotangents), d_pre [256][B], d_conv
 * [160][B]; loss_partials (apg_quad_mlp_loss_partials_count(B)), loss [1] or
 * NULL, states [H][12][B] or NULL.
 *   dW_out = d_zout h3^T, the rest as for apg_quad_mlp_rollout_bwd with N = B.
 * workspace: apg_quad_mlp_concurrent_workspace_floats(). */
int apg_quad_mlp_concurrent_workspace_floats(void);
int apg_quad_mlp_concurrent_fwd_bwd(
    const float *feat, const float *in_ref, const float *state0, const float *ref,
    int ref_cols, float dt, const ApgQuadParams *params,
    const ApgQuadLossWeights *weights, const ApgMlpPolicy *policy, int B, int H,
    float *x1, float *h, unsigned *relu_mask, float *d_zout, float *d_pre,
    float *d_conv, float *loss_partials, float *loss, float *states,
    float *workspace, apg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* APG_PLANES_H_ */
