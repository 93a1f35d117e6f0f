"""Drop-in for neural_control.dynamics.cartpole_dynamics.CartpoleDynamics
(reference: neural_control/dynamics/cartpole_dynamics.py:21-119): same
constructor and `dyn(state, action, dt)` surface; HIP step kernel
(apg_cartpole_step_fwd / _bwd)."""
from .. import functional as F

# neural_control/dynamics/config_cartpole.json:1-11
DEFAULT_CONFIG = {
    "masscart": 1.0, "masspole": 0.1, "length": 0.5, "max_force_mag": 30.0,
    "muc": 0.0005, "mup": 0.000002, "wind": 0.0, "vel_drag": 0.0,
    "contact": 0.0, "delay": 0.0,
}
gravity = 9.81


class CartpoleDynamics:

    def __init__(self, modified_params={}, test_time=0, batch_size=1):
        self.batch_size = batch_size
        self.cfg = dict(DEFAULT_CONFIG)
        self.test_time = test_time
        self.cfg.update(modified_params)
        self.cfg["friction"] = .5      # cartpole_dynamics.py:34 (sic)
        self.cfg["total_mass"] = self.cfg["masspole"] + self.cfg["masscart"]
        self.cfg["polemass_length"] = self.cfg["masspole"] * self.cfg["length"]
        self.timestamp = 0
        self.params = F.cartpole_params(self.cfg)

    def __call__(self, state, action, dt):
        return self.simulate_cartpole(state, action, dt)

    def simulate_cartpole(self, state, action, delta_t):
        """state [B,4] = [x, x_dot, theta, theta_dot], action [B,1] ->
        next state [B,4]."""
        self.timestamp += .05          # side effect kept (:57)
        return F.cartpole_step(state, action, delta_t, self.params)

    def rollout(self, state0, action_seq, dt):
        """H-step no-grad unroll in one kernel: states [B, H, 4]."""
        self.timestamp += .05 * action_seq.shape[1]
        return F.cartpole_rollout_fwd(F._f32c(state0), F._f32c(action_seq), dt,
                                      self.params)
