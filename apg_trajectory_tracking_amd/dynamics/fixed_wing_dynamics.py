"""Drop-in for neural_control.dynamics.fixed_wing_dynamics.FixedWingDynamics
(reference: neural_control/dynamics/fixed_wing_dynamics.py:13-267): same
constructor and `dyn(state, action, dt)` / `dyn.simulate_fixed_wing(state,
action, dt)` surface; the step is one HIP kernel (apg_wing_step_fwd) with an
analytic VJP (apg_wing_step_bwd)."""
import ctypes

import numpy as np
import torch
from torch import nn

from .. import _capi
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


class _LearntWingStep(torch.autograd.Function):
    """simulate_fixed_wing with live parameters: apg_wing_learnt_step_fwd /
    _bwd.  theta: the 41 ApgWingParams entries as ONE tensor (the I_* slots
    are placeholders), inertia: the 3x3 parameter."""

    @staticmethod
    def forward(ctx, state, action, theta, inertia, dt):
        s, a = F._f32c(state), F._f32c(action)
        _capi.require_device(s, a)
        if s.dim() != 2 or s.shape[1] != 12 or a.shape != (s.shape[0], 4):
            raise ValueError("state [B,12] / action [B,4] expected")
        # ONE device-to-host read of the 50 numbers per step: this is the
        # cold dynamics-fitting path (scripts/train_base.py:160-186)
        host = torch.cat((theta.detach().reshape(-1).float(),
                          inertia.detach().reshape(-1).float())).cpu().tolist()
        params = _capi.ApgWingParams(*host[:41])
        I9 = (ctypes.c_float * 9)(*host[41:])
        out = torch.empty_like(s)
        _capi.check(_capi.lib().apg_wing_learnt_step_fwd(
            _capi.ptr(s), _capi.ptr(a), float(dt), ctypes.byref(params), I9,
            s.shape[0], _capi.ptr(out), _capi.stream_of(s)),
            "apg_wing_learnt_step_fwd")
        ctx.save_for_backward(s, a)
        ctx.meta = (float(dt), params, I9)
        return out

    @staticmethod
    def backward(ctx, grad_next):
        s, a = ctx.saved_tensors
        dt, params, I9 = ctx.meta
        g = F._f32c(grad_next)
        B = s.shape[0]
        gs = torch.empty_like(s) if ctx.needs_input_grad[0] else None
        ga = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        lib = _capi.lib()
        gp = torch.empty(lib.apg_wing_learnt_param_count(), device=s.device)
        ws = torch.empty(max(1, lib.apg_wing_learnt_workspace_floats(B)),
                         device=s.device)
        _capi.check(lib.apg_wing_learnt_step_bwd(
            _capi.ptr(s), _capi.ptr(a), dt, ctypes.byref(params), I9, B,
            _capi.ptr(g), _capi.ptr(gs), _capi.ptr(ga), _capi.ptr(gp),
            _capi.ptr(ws), _capi.stream_of(s)), "apg_wing_learnt_step_bwd")
        return gs, ga, gp[:41], gp[41:].view(3, 3), None


class LearntFixedWingDynamics(nn.Module, FixedWingDynamics):
    """Drop-in for neural_control.dynamics.fixed_wing_dynamics.
    LearntFixedWingDynamics (fixed_wing_dynamics.py:270-326; beyond
    SURVEY.md §8): every physical parameter of the fixed-wing model is
    trainable (`cfg`, a ParameterDict of [1] tensors, and the 3x3 inertia
    matrix `I`), and a zero-initialised residual MLP (16 -> 64 -> 12) is
    added to the simulated next state.  Same parameter names as the
    reference, so its state_dicts load.  The step and its VJP - including
    the cotangents of the 37 + 9 physical parameters - are HIP kernels
    (csrc/wing_learnt.hip)."""

    def __init__(self, modified_params={}):
        FixedWingDynamics.__init__(self, modified_params)
        nn.Module.__init__(self)
        c = self.cfg
        self.I = nn.Parameter(torch.tensor(
            [[c["I_xx"], 0, -c["I_xz"]], [0, c["I_yy"], 0],
             [-c["I_xz"], 0, c["I_zz"]]]))
        self.cfg = nn.ParameterDict({
            key: nn.Parameter(torch.tensor([val]))
            for key, val in c.items() if "I_" not in key})
        self.linear_state_1 = nn.Linear(16, 64)
        nn.init.constant_(self.linear_state_1.weight, 0)
        nn.init.constant_(self.linear_state_1.bias, 0)
        self.linear_state_2 = nn.Linear(64, 12)
        nn.init.constant_(self.linear_state_2.weight, 0)
        nn.init.constant_(self.linear_state_2.bias, 0)

    def _theta(self):
        zero = self.I.new_zeros(1)
        return torch.cat([self.cfg[k] if k in self.cfg else zero
                          for k in _capi.WING_PARAM_FIELDS])

    def simulate_fixed_wing(self, state, action, dt):
        return _LearntWingStep.apply(state, action, self._theta(), self.I, dt)

    def state_transformer(self, state, action):
        state_action = torch.cat((state, action), dim=1)
        layer_1 = torch.relu(self.linear_state_1(state_action))
        return self.linear_state_2(layer_1)

    def forward(self, state, action, dt):
        new_state = self.simulate_fixed_wing(state, action, dt)
        return new_state + self.state_transformer(state, action)

    def __call__(self, *args, **kwargs):      # nn.Module.__call__, not the
        return nn.Module.__call__(self, *args, **kwargs)  # dynamics' shortcut

