"""Drop-in for neural_control.dynamics.quad_dynamics_flightmare.FlightmareDynamics
(reference: neural_control/dynamics/quad_dynamics_flightmare.py:7-216 and
quad_dynamics_base.py:9-57): same constructor, same `dyn(state, action, dt)`
/ `dyn.simulate_quadrotor(action, state, dt)` call surface, same parameter
keys in `modified_params`.  The step itself is one HIP kernel
(apg_quad_step_fwd) with an analytic VJP (apg_quad_step_bwd)."""
import numpy as np

from .. import functional as F

# neural_control/dynamics/config_quad.json:1-29
DEFAULT_CONFIG = {
    "mass": 0.723,
    "rotational_drag": [0, 0, 0],
    "translational_drag": [0, 0, 0],
    "arm_length": 0.31,
    "frame_inertia": [4.5, 4.5, 7.0],
    "gravity": [0, 0, -9.81],
    "kinv_ang_vel_tau": [16.6, 16.6, 5.0],
}


class Dynamics:
    """Parameter holder (quad_dynamics_base.py:9-57)."""

    def __init__(self, modified_params={}):
        self.cfg = {k: (list(v) if isinstance(v, list) else v)
                    for k, v in DEFAULT_CONFIG.items()}
        self.cfg.update(modified_params)
        self.mass = self.cfg["mass"]
        self.arm_length = self.cfg["arm_length"]
        self.kinv_ang_vel_tau = np.array(self.cfg["kinv_ang_vel_tau"])
        self.inertia_vector = (
            self.mass / 12.0 * self.arm_length**2 *
            np.array(self.cfg["frame_inertia"])
        )
        self.params = F.quad_params(self.cfg)


class FlightmareDynamics(Dynamics):

    def __init__(self, modified_params={}, simulate_rotors=False):
        super().__init__(modified_params=modified_params)
        if simulate_rotors:
            # the reference keeps the rotor model commented out (:157-165)
            raise NotImplementedError("simulate_rotors is not implemented")
        self.simulate_rotors = simulate_rotors

    def __call__(self, state, action, dt):
        return self.simulate_quadrotor(action, state, dt)

    def simulate_quadrotor(self, action, state, dt):
        """state [B,12] = [pos, euler rpy, vel, body rates], action [B,4] in
        [0,1] -> next state [B,12] (fp32, differentiable)."""
        return F.quad_step(state, action, dt, self.params)

    def rollout(self, state0, action_seq, dt):
        """No-grad H-step unroll -> states [B,H,12] (apg_quad_rollout_fwd)."""
        return F.quad_rollout_fwd(
            F._f32c(state0), F._f32c(action_seq), dt, self.params)
