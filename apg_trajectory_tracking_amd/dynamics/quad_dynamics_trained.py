"""Drop-in for neural_control.dynamics.quad_dynamics_trained.LearntDynamics
(reference: quad_dynamics_trained.py:10-69): the Flightmare
step with a learnable 4x4 action transform in front, a residual MLP
(16 -> 64 -> 12, zero-initialised) behind and learnable physical parameters
(mass, inertia vector, kinv).  Used by TrainBase.train_dynamics_model
(scripts/train_base.py:160-186) to fit the simulator to another dynamics.

The step itself stays the HIP kernel; its VJP gives dL/dstate and dL/daction,
and the physical-parameter gradients follow in closed form from the same
cotangent (mass cancels out of the dynamics - its gradient is exactly zero in
the reference too):
    w'_i = w_i + dt (K_i (a_i - 1/2 - w_i) + d_r,i / J_i)
    dL/dK_i = sum_b lam_w'_i dt (a_i - 1/2 - w_i),
    dL/dJ_i = -sum_b lam_w'_i dt d_r,i / J_i^2.

Reference quirk kept on purpose (pinned by golden G10 `steps.*`): the matrices
the simulated step multiplies with, `torch_inertia_J` / `torch_kinv_ang_vel_tau`,
are `torch.diag(...)` COPIES made once in `__init__`
(quad_dynamics_trained.py:48-50).  `torch_inertia_vector` / `torch_kinv_vector`
receive gradients and are moved by the optimizer (and by `load_state_dict`),
but the step keeps integrating with the kinv / inertia of construction time.
The kernel's parameter block is therefore filled once, in `__init__`, and the
closed-form gradients above are evaluated at those initial values.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _capi
from .. import functional as F
from .quad_dynamics_flightmare import FlightmareDynamics


class _LearntStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, action, kinv, inertia, mass, dt, params, rot_drag,
                inertia0):
        s, a = F._f32c(state), F._f32c(action)
        _capi.require_device(s, a)
        out = torch.empty_like(s)
        _capi.check(_capi.lib().apg_quad_step_fwd(
            _capi.ptr(s), _capi.ptr(a), float(dt), ctypes.byref(params),
            s.shape[0], _capi.LAYOUT_AOS, _capi.ptr(out), _capi.stream_of(s)),
            "apg_quad_step_fwd")
        # `inertia` here is the value the step was simulated with (the
        # construction-time copy), not the live parameter
        ctx.save_for_backward(s, a, inertia0.detach().to(s.device))
        ctx.meta = (float(dt), params, rot_drag)
        return out

    @staticmethod
    def backward(ctx, grad_next):
        s, a, inertia = ctx.saved_tensors
        dt, params, rot_drag = ctx.meta
        g = F._f32c(grad_next)
        gs, ga = torch.empty_like(s), torch.empty_like(a)
        _capi.check(_capi.lib().apg_quad_step_bwd(
            _capi.ptr(s), _capi.ptr(a), dt, ctypes.byref(params), s.shape[0],
            _capi.LAYOUT_AOS, _capi.ptr(g), _capi.ptr(gs), _capi.ptr(ga),
            _capi.stream_of(s)), "apg_quad_step_bwd")
        lam_w = g[:, 9:12]
        g_kinv = (lam_w * (dt * ((a[:, 1:] - 0.5) - s[:, 9:12]))).sum(0)
        g_inertia = -(lam_w.sum(0)) * dt * rot_drag.to(s.device) / inertia**2
        g_mass = torch.zeros(1, device=s.device)
        return gs, ga, g_kinv, g_inertia, g_mass, None, None, None, None


class LearntDynamics(nn.Module, FlightmareDynamics):

    def __init__(self, initial_params={}):
        FlightmareDynamics.__init__(self, initial_params)
        nn.Module.__init__(self)
        self.linear_at = nn.Parameter(torch.diag(torch.ones(4)))
        self.linear_state_1 = nn.Linear(16, 64)
        nn.init.constant_(self.linear_state_1.weight, 0)
        nn.init.constant_(self.linear_state_1.bias, 0)
        self.linear_state_2 = nn.Linear(64, 12)
        nn.init.constant_(self.linear_state_2.weight, 0)
        nn.init.constant_(self.linear_state_2.bias, 0)
        mass = float(self.mass)
        self.mass = nn.Parameter(torch.tensor([mass]))
        self.torch_inertia_vector = nn.Parameter(
            torch.from_numpy(np.asarray(self.inertia_vector)).float())
        self.torch_kinv_vector = nn.Parameter(
            torch.tensor(np.asarray(self.kinv_ang_vel_tau)).float())
        # constants of the step, as NON-persistent buffers: they follow
        # module.to(device) (no host-to-device copy per forward) and stay out
        # of the state_dict (the reference's has no such keys)
        self.register_buffer("_rot_drag", torch.tensor(
            [float(v) for v in self.cfg["rotational_drag"]]), persistent=False)
        # the reference's torch.diag copies (:48-50): what the step uses from
        # now on, whatever happens to the parameters above
        self.register_buffer("_inertia0",
                             self.torch_inertia_vector.detach().clone(),
                             persistent=False)
        self._snapshot_params()

    def _snapshot_params(self):
        """Fill the kernel's parameter block from the (initial) physical
        parameters - called once, from __init__."""
        kinv = self.torch_kinv_vector.detach().cpu().tolist()
        inertia = self.torch_inertia_vector.detach().cpu().tolist()
        self.params.mass = float(self.mass.detach().cpu())
        for i in range(3):
            self.params.kinv[i] = kinv[i]
            self.params.inertia[i] = inertia[i]

    def state_transformer(self, state, action):
        state_action = torch.cat((state, action), dim=1)
        layer_1 = torch.relu(self.linear_state_1(state_action))
        return self.linear_state_2(layer_1)

    def forward(self, state, action, dt):
        action_transformed = torch.matmul(
            self.linear_at, torch.unsqueeze(action, 2))[:, :, 0]
        new_state = _LearntStep.apply(
            state, action_transformed, self.torch_kinv_vector,
            self.torch_inertia_vector, self.mass, dt, self.params,
            self._rot_drag, self._inertia0)
        return new_state + self.state_transformer(state, action_transformed)

    def __call__(self, *args, **kwargs):        # nn.Module.__call__, not the
        return nn.Module.__call__(self, *args, **kwargs)  # dynamics' shortcut
