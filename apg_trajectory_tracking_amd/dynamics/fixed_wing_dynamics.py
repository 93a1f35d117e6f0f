"""Drop-in for neural_control.dynamics.fixed_wing_dynamics.FixedWingDynamics
(reference: neural_control/dynamics/fixed_wing_dynamics.py:13-267): same
constructor and `dyn(state, action, dt)` / `dyn.simulate_fixed_wing(state,
action, dt)` surface; the step is one HIP kernel (apg_wing_step_fwd) with an
analytic VJP (apg_wing_step_bwd)."""
import numpy as np

from .. import functional as F

# neural_control/dynamics/config_fixed_wing.json:1-42
DEFAULT_CONFIG = {
    "mass": 1.01, "I_xx": 0.04766, "I_yy": 0.05005, "I_zz": 0.09558,
    "I_xz": -0.00105, "rho": 1.225, "S": 0.276, "c": 0.185, "b": 1.54,
    "g": 9.81,
    "CL0": 0.39, "CL_alpha": 4.5321, "CL_q": 0.318, "CL_del_e": 0.527,
    "CD0": 0.0765, "CD_alpha": 0.3346, "CD_q": 0.354, "CD_del_e": 0.004,
    "CY0": 0.0, "CY_beta": -0.033, "CY_p": -0.1, "CY_r": 0.039,
    "CY_del_a": 0.0, "CY_del_r": 0.225,
    "Cl0": 0.0, "Cl_beta": -0.081, "Cl_p": -0.529, "Cl_r": 0.159,
    "Cl_del_a": -0.453, "Cl_del_r": 0.005,
    "Cm0": 0.02, "Cm_alpha": -1.4037, "Cm_q": -0.1324, "Cm_del_e": -0.4236,
    "Cn0": 0.0, "Cn_beta": 0.189, "Cn_p": -0.083, "Cn_r": -0.948,
    "Cn_del_a": -0.041, "Cn_del_r": -0.077,
    "epsilon": 0.16534698176788384,
}

alpha_bound = float(10 / 180 * np.pi)


class FixedWingDynamics:

    def __init__(self, modified_params={}):
        self.cfg = dict(DEFAULT_CONFIG)
        self.pi = np.pi
        self.cfg.update(modified_params)
        self.params = F.wing_params(self.cfg)

    def __call__(self, state, action, dt):
        return self.simulate_fixed_wing(state, action, dt)

    def simulate_fixed_wing(self, state, action, dt):
        """state [B,12] = [pos NED, vel body, euler, body rates],
        action [B,4] -> next state [B,12] (fp32, differentiable)."""
        return F.wing_step(state, action, dt, self.params)

    def rollout(self, state0, action_seq, dt):
        return F.wing_rollout_fwd(
            F._f32c(state0), F._f32c(action_seq), dt, self.params)
