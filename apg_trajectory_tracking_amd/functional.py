"""Differentiable torch-facing wrappers over the C ABI (include/apg.h).

Each `torch.autograd.Function` here stands where the reference relies on
torch.autograd over ~60 eager ops per step: forward enqueues one HIP kernel,
backward enqueues its hand-derived adjoint kernel.  Tensors are the
reference's row-major ones (`layout="aos"`) unless stated; the fused rollouts
also accept the device-native SoA layout (`layout="soa"`, batch fastest).
"""
import ctypes

import torch

from . import _capi
from ._capi import (LAYOUT_AOS, LAYOUT_PACKED, LAYOUT_SOA, check, lib, ptr,
                    require_device, stream_of)

# "packed" (rows [rows][B][C], include/apg.h) is understood by the fused
# quadrotor rollout only
_LAYOUTS = {"aos": LAYOUT_AOS, "soa": LAYOUT_SOA, "packed": LAYOUT_PACKED}


def _layout(layout):
    try:
        return _LAYOUTS[layout]
    except KeyError:
        raise ValueError(
            f"layout must be 'aos', 'soa' or 'packed', got {layout!r}")


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------- parameters
QUAD_LOSS_WEIGHTS = dict(pos=10.0, vel=1.0, av=0.1, rates=0.1, thrust=5.0)
WING_LOSS_WEIGHTS = dict(pos=10.0, action=0.1)


def quad_params(cfg):
    """cfg: the dict of neural_control/dynamics/config_quad.json after
    `update(modified_params)` -> ApgQuadParams (inertia as in
    quad_dynamics_base.py:33-36)."""
    p = _capi.ApgQuadParams()
    p.mass = float(cfg["mass"])
    scale = float(cfg["mass"]) / 12.0 * float(cfg["arm_length"])**2
    for i in range(3):
        p.kinv[i] = float(cfg["kinv_ang_vel_tau"][i])
        p.inertia[i] = scale * float(cfg["frame_inertia"][i])
        p.gravity[i] = float(cfg["gravity"][i])
        p.trans_drag[i] = float(cfg["translational_drag"][i])
        p.rot_drag[i] = float(cfg["rotational_drag"][i])
    return p


def quad_loss_weights(**kw):
    w = dict(QUAD_LOSS_WEIGHTS)
    w.update(kw)
    return _capi.ApgQuadLossWeights(**{k: float(v) for k, v in w.items()})


def wing_params(cfg):
    p = _capi.ApgWingParams()
    for n in _capi.WING_PARAM_FIELDS:
        setattr(p, n, float(cfg[n]))
    return p


def wing_loss_weights(**kw):
    w = dict(WING_LOSS_WEIGHTS)
    w.update(kw)
    return _capi.ApgWingLossWeights(**{k: float(v) for k, v in w.items()})


def cartpole_params(cfg):
    p = _capi.ApgCartpoleParams()
    p.masscart = float(cfg["masscart"])
    p.masspole = float(cfg["masspole"])
    p.length = float(cfg["length"])
    p.max_force_mag = float(cfg["max_force_mag"])
    p.friction = float(cfg["friction"])
    p.gravity = float(cfg.get("gravity", 9.81))
    return p


# ------------------------------------------------------- generic step op
class _StepFn(torch.autograd.Function):
    """next_state = dyn(state, action, dt) with an analytic VJP."""

    @staticmethod
    def forward(ctx, state, action, dt, params, fwd_name, bwd_name):
        s, a = _f32c(state), _f32c(action)
        require_device(s, a)
        if s.dim() != 2 or a.dim() != 2 or s.shape[0] != a.shape[0]:
            raise ValueError(
                f"state [B,S] / action [B,A] expected, got {tuple(s.shape)} "
                f"and {tuple(a.shape)}")
        out = torch.empty_like(s)
        check(getattr(lib(), fwd_name)(
            ptr(s), ptr(a), float(dt), ctypes.byref(params), s.shape[0],
            LAYOUT_AOS, ptr(out), stream_of(s)), fwd_name)
        ctx.save_for_backward(s, a)
        ctx.meta = (float(dt), params, bwd_name)
        return out

    @staticmethod
    def backward(ctx, grad_next):
        s, a = ctx.saved_tensors
        dt, params, bwd_name = ctx.meta
        g = _f32c(grad_next)
        need_s, need_a = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gs = torch.empty_like(s) if need_s else None
        ga = torch.empty_like(a) if need_a else None
        check(getattr(lib(), bwd_name)(
            ptr(s), ptr(a), dt, ctypes.byref(params), s.shape[0], LAYOUT_AOS,
            ptr(g), ptr(gs), ptr(ga), stream_of(s)), bwd_name)
        return gs, ga, None, None, None, None


def quad_step(state, action, dt, params):
    return _StepFn.apply(state, action, dt, params, "apg_quad_step_fwd",
                         "apg_quad_step_bwd")


def wing_step(state, action, dt, params):
    return _StepFn.apply(state, action, dt, params, "apg_wing_step_fwd",
                         "apg_wing_step_bwd")


def cartpole_step(state, action, dt, params):
    return _StepFn.apply(state, action, dt, params, "apg_cartpole_step_fwd",
                         "apg_cartpole_step_bwd")


# --------------------------------------------------- raw fused rollouts
def _seq_shape(t, layout):
    """-> (B, H, C) of a sequence tensor in the given layout."""
    if t.dim() != 3:
        raise ValueError(f"3-d sequence tensor expected, got {tuple(t.shape)}")
    if layout == LAYOUT_AOS:
        return t.shape[0], t.shape[1], t.shape[2]
    if layout == LAYOUT_PACKED:
        return t.shape[1], t.shape[0], t.shape[2]
    return t.shape[2], t.shape[0], t.shape[1]


def _state_batch(t, layout):
    if layout == LAYOUT_PACKED:
        if t.dim() != 3 or t.shape[2] != 4:
            raise ValueError(
                f"packed state tensor [S/4, B, 4] expected, got {tuple(t.shape)}")
        return t.shape[1]
    if t.dim() != 2:
        raise ValueError(f"2-d state tensor expected, got {tuple(t.shape)}")
    return t.shape[0] if layout == LAYOUT_AOS else t.shape[1]


def _states_shape(B, H, S, lay):
    if lay == LAYOUT_AOS:
        return (B, H, S)
    if lay == LAYOUT_PACKED:
        return (H, S // 4, B, 4)
    return (H, S, B)


def quad_rollout_fwd_bwd(state0, actions, ref, dt, params, weights=None,
                         layout="aos", want_grad_state0=True,
                         want_states=False, want_loss=True, out=None):
    """Fused H-step unroll + quad_mpc_loss + adjoint (apg_quad_rollout_fwd_bwd).

    Returns dict(loss [1] or None, loss_partials, grad_actions, grad_state0,
    states).  `out` may carry pre-allocated output tensors under the same
    keys (used by bench.py to keep allocation out of the timed region).
    """
    lay = _layout(layout)
    weights = weights or quad_loss_weights()
    require_device(state0, actions, ref)
    B, H, A = _seq_shape(actions, lay)
    Br, Hr, ref_cols = _seq_shape(ref, lay)
    if A != 4 or _state_batch(state0, lay) != B or Br != B or Hr != H:
        raise ValueError("inconsistent rollout shapes")
    out = dict(out or {})
    dev = state0.device

    def get(key, shape, wanted=True):
        if not wanted:
            return None
        t = out.get(key)
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
        return t
    partials = get("loss_partials", (_capi.loss_partials_count(B),))
    loss = get("loss", (1,), want_loss)
    ga = get("grad_actions", actions.shape)
    gs = get("grad_state0", state0.shape, want_grad_state0)
    states = get("states", _states_shape(B, H, 12, lay), want_states)
    check(lib().apg_quad_rollout_fwd_bwd(
        ptr(state0), ptr(actions), ptr(ref), ref_cols, float(dt),
        ctypes.byref(params), ctypes.byref(weights), B, H, lay, ptr(partials),
        ptr(loss), ptr(ga), ptr(gs), ptr(states), None, stream_of(state0)),
        "apg_quad_rollout_fwd_bwd")
    return dict(loss=loss, loss_partials=partials, grad_actions=ga,
                grad_state0=gs, states=states)


def quad_rollout_fwd(state0, actions, dt, params, layout="aos"):
    lay = _layout(layout)
    require_device(state0, actions)
    B, H, _ = _seq_shape(actions, lay)
    states = torch.empty((B, H, 12) if lay == LAYOUT_AOS else (H, 12, B),
                         dtype=torch.float32, device=state0.device)
    check(lib().apg_quad_rollout_fwd(
        ptr(state0), ptr(actions), float(dt), ctypes.byref(params), B, H, lay,
        ptr(states), stream_of(state0)), "apg_quad_rollout_fwd")
    return states


class _QuadRolloutLoss(torch.autograd.Function):
    """loss = quad_mpc_loss(unroll(dyn, state0, action_seq), ref, action_seq)
    as ONE kernel; the gradients w.r.t. action_seq / state0 are produced by
    the same launch and handed to autograd in backward()."""

    @staticmethod
    def forward(ctx, state0, action_seq, ref, dt, params, weights, layout):
        s, a, r = _f32c(state0), _f32c(action_seq), _f32c(ref)
        res = quad_rollout_fwd_bwd(
            s, a, r, dt, params, weights, layout=layout,
            want_grad_state0=ctx.needs_input_grad[0])
        ctx.save_for_backward(res["grad_actions"], res["grad_state0"])
        return res["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        ga, gs = ctx.saved_tensors
        ga = ga * g if ctx.needs_input_grad[1] else None
        gs = gs * g if (gs is not None and ctx.needs_input_grad[0]) else None
        return gs, ga, None, None, None, None, None


def quad_rollout_loss(state0, action_seq, ref, dt, params, weights=None,
                      layout="aos"):
    return _QuadRolloutLoss.apply(state0, action_seq, ref, dt, params,
                                  weights or quad_loss_weights(), layout)


# ------------------------------------------------------------ quad loss
# ------------------------------------------- rollout through LearntDynamics
def _learnt_model(dyn):
    """ApgLearntResidual over a LearntDynamics module's own tensors."""
    tensors = [dyn.linear_at, dyn.linear_state_1.weight, dyn.linear_state_1.bias,
               dyn.linear_state_2.weight, dyn.linear_state_2.bias]
    for t in tensors:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("LearntDynamics tensors must be contiguous fp32 "
                               "device tensors (module.to('cuda'))")
    if tuple(tensors[1].shape) != (64, 16) or tuple(tensors[3].shape) != (12, 64):
        raise ValueError("fused rollout expects the 16 -> 64 -> 12 residual network")
    return _capi.ApgLearntResidual(*[t.data_ptr() for t in tensors])


def quad_learnt_rollout_fwd_bwd(dyn, state0, actions, ref, dt, weights=None,
                                layout="aos", want_grad_state0=True,
                                want_states=False, out=None):
    """Fused H-step unroll through `dyn` (a LearntDynamics module: action
    transform + analytic step + residual network) + quad_mpc_loss + adjoint
    down to dL/dactions and dL/dstate0 (apg_quad_learnt_rollout_fwd_bwd).  The
    module's parameters are read, never differentiated."""
    lay = _layout(layout)
    if lay == LAYOUT_PACKED:
        raise ValueError("the learnt-dynamics rollout takes 'aos' or 'soa' tensors")
    weights = weights or quad_loss_weights()
    require_device(state0, actions, ref)
    B, H, A = _seq_shape(actions, lay)
    Br, Hr, ref_cols = _seq_shape(ref, lay)
    if A != 4 or _state_batch(state0, lay) != B or Br != B or Hr != H:
        raise ValueError("inconsistent rollout shapes")
    out = dict(out or {})
    dev = state0.device

    def get(key, shape, wanted=True):
        if not wanted:
            return None
        t = out.get(key)
        return t if t is not None else torch.empty(shape, dtype=torch.float32,
                                                   device=dev)
    partials = get("loss_partials", (_capi.loss_partials_count(B),))
    loss = get("loss", (1,))
    ga = get("grad_actions", actions.shape)
    gs = get("grad_state0", state0.shape, want_grad_state0)
    states = get("states", _states_shape(B, H, 12, lay), want_states)
    model = _learnt_model(dyn)
    check(lib().apg_quad_learnt_rollout_fwd_bwd(
        ptr(state0), ptr(actions), ptr(ref), ref_cols, float(dt),
        ctypes.byref(dyn.params), ctypes.byref(model), ctypes.byref(weights), B, H,
        lay, ptr(partials), ptr(loss), ptr(ga), ptr(gs), ptr(states),
        stream_of(state0)), "apg_quad_learnt_rollout_fwd_bwd")
    return dict(loss=loss, loss_partials=partials, grad_actions=ga,
                grad_state0=gs, states=states)


class _QuadLearntRolloutLoss(torch.autograd.Function):
    """loss = quad_mpc_loss(unroll(learnt_dyn, state0, action_seq), ref,
    action_seq) as ONE kernel; gradients w.r.t. action_seq / state0 only."""

    @staticmethod
    def forward(ctx, state0, action_seq, ref, dt, dyn, weights):
        s, a, r = _f32c(state0), _f32c(action_seq), _f32c(ref)
        res = quad_learnt_rollout_fwd_bwd(
            dyn, s, a, r, dt, weights, want_grad_state0=ctx.needs_input_grad[0])
        ctx.save_for_backward(res["grad_actions"], res["grad_state0"])
        return res["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        ga, gs = ctx.saved_tensors
        ga = ga * g if ctx.needs_input_grad[1] else None
        gs = gs * g if (gs is not None and ctx.needs_input_grad[0]) else None
        return gs, ga, None, None, None, None


def quad_learnt_rollout_loss(dyn, state0, action_seq, ref, dt, weights=None):
    return _QuadLearntRolloutLoss.apply(state0, action_seq, ref, dt, dyn,
                                        weights or quad_loss_weights())


class _QuadLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, states, ref, actions, weights):
        s, r, a = _f32c(states), _f32c(ref), _f32c(actions)
        require_device(s, r, a)
        B, H, C = s.shape
        if C != 12 or a.shape != (B, H, 4) or r.shape[:2] != (B, H):
            raise ValueError("states [B,H,12], ref [B,H,9|6], actions [B,H,4]")
        partials = torch.empty(_capi.loss_partials_count(B), device=s.device)
        loss = torch.empty(1, device=s.device)
        gs = torch.empty_like(s) if ctx.needs_input_grad[0] else None
        ga = torch.empty_like(a) if ctx.needs_input_grad[2] else None
        check(lib().apg_quad_loss_fwd_bwd(
            ptr(s), ptr(r), r.shape[2], ptr(a), ctypes.byref(weights), B, H,
            LAYOUT_AOS, ptr(partials), ptr(loss), ptr(gs), ptr(ga),
            stream_of(s)), "apg_quad_loss_fwd_bwd")
        ctx.save_for_backward(gs, ga)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        gs, ga = ctx.saved_tensors
        return (None if gs is None else gs * g, None,
                None if ga is None else ga * g, None)


def quad_loss(states, ref, actions, weights=None):
    return _QuadLoss.apply(states, ref, actions, weights or quad_loss_weights())


# -------------------------------------------------------- quad features
class _QuadFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state):
        s = _f32c(state)
        require_device(s)
        if s.dim() != 2 or s.shape[1] != 12:
            raise ValueError(f"state [B,12] expected, got {tuple(s.shape)}")
        f = torch.empty(s.shape[0], 15, device=s.device)
        check(lib().apg_quad_features_fwd(
            ptr(s), s.shape[0], LAYOUT_AOS, ptr(f), stream_of(s)),
            "apg_quad_features_fwd")
        ctx.save_for_backward(s)
        return f

    @staticmethod
    def backward(ctx, gf):
        (s,) = ctx.saved_tensors
        g = _f32c(gf)
        gs = torch.empty_like(s)
        check(lib().apg_quad_features_bwd(
            ptr(s), ptr(g), s.shape[0], LAYOUT_AOS, ptr(gs), stream_of(s)),
            "apg_quad_features_bwd")
        return gs


def quad_features(state):
    return _QuadFeatures.apply(state)


# ------------------------------------------------------------ fixed wing
def _alloc_outs(out, dev, B, H, S, A, lay, state0, actions, want_grad_state0,
                want_states, want_loss):
    out = dict(out or {})

    def get(key, shape, wanted=True):
        if not wanted:
            return None
        t = out.get(key)
        if t is None:
            t = torch.empty(shape, dtype=torch.float32, device=dev)
        return t
    return dict(
        loss_partials=get("loss_partials", (_capi.loss_partials_count(B),)),
        loss=get("loss", (1,), want_loss),
        grad_actions=get("grad_actions", actions.shape),
        grad_state0=get("grad_state0", state0.shape, want_grad_state0),
        states=get("states", (B, H, S) if lay == LAYOUT_AOS else (H, S, B),
                   want_states),
    )


def wing_rollout_fwd_bwd(state0, actions, ref, dt, params, weights=None,
                         layout="aos", want_grad_state0=True,
                         want_states=False, want_loss=True, out=None):
    """Fused H-step unroll + fixed_wing_mpc_loss + adjoint
    (apg_wing_rollout_fwd_bwd); ref is the [B,H,3] linear reference."""
    lay = _layout(layout)
    weights = weights or wing_loss_weights()
    require_device(state0, actions, ref)
    B, H, A = _seq_shape(actions, lay)
    Br, Hr, C = _seq_shape(ref, lay)
    if A != 4 or C != 3 or _state_batch(state0, lay) != B or (Br, Hr) != (B, H):
        raise ValueError("inconsistent rollout shapes")
    o = _alloc_outs(out, state0.device, B, H, 12, 4, lay, state0, actions,
                    want_grad_state0, want_states, want_loss)
    check(lib().apg_wing_rollout_fwd_bwd(
        ptr(state0), ptr(actions), ptr(ref), float(dt), ctypes.byref(params),
        ctypes.byref(weights), B, H, lay, ptr(o["loss_partials"]),
        ptr(o["loss"]), ptr(o["grad_actions"]), ptr(o["grad_state0"]),
        ptr(o["states"]), None, stream_of(state0)), "apg_wing_rollout_fwd_bwd")
    return o


def wing_rollout_fwd(state0, actions, dt, params, layout="aos"):
    lay = _layout(layout)
    require_device(state0, actions)
    B, H, _ = _seq_shape(actions, lay)
    states = torch.empty((B, H, 12) if lay == LAYOUT_AOS else (H, 12, B),
                         dtype=torch.float32, device=state0.device)
    check(lib().apg_wing_rollout_fwd(
        ptr(state0), ptr(actions), float(dt), ctypes.byref(params), B, H, lay,
        ptr(states), stream_of(state0)), "apg_wing_rollout_fwd")
    return states


class _WingRolloutLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state0, action_seq, ref, dt, params, weights, layout):
        s, a, r = _f32c(state0), _f32c(action_seq), _f32c(ref)
        res = wing_rollout_fwd_bwd(
            s, a, r, dt, params, weights, layout=layout,
            want_grad_state0=ctx.needs_input_grad[0])
        ctx.save_for_backward(res["grad_actions"], res["grad_state0"])
        return res["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        ga, gs = ctx.saved_tensors
        ga = ga * g if ctx.needs_input_grad[1] else None
        gs = gs * g if (gs is not None and ctx.needs_input_grad[0]) else None
        return gs, ga, None, None, None, None, None


def wing_rollout_loss(state0, action_seq, ref, dt, params, weights=None,
                      layout="aos"):
    return _WingRolloutLoss.apply(state0, action_seq, ref, dt, params,
                                  weights or wing_loss_weights(), layout)


# -------------------------------------------------------------- cartpole
def cartpole_rollout_fwd_bwd(state0, actions, dt, params, layout="aos",
                             want_grad_state0=True, want_states=False,
                             want_loss=True, out=None):
    """Fused make_reference + H-step unroll + cartpole_loss_mpc + adjoint
    (apg_cartpole_rollout_fwd_bwd)."""
    lay = _layout(layout)
    require_device(state0, actions)
    B, H, A = _seq_shape(actions, lay)
    if A != 1 or _state_batch(state0, lay) != B:
        raise ValueError("inconsistent rollout shapes")
    o = _alloc_outs(out, state0.device, B, H, 4, 1, lay, state0, actions,
                    want_grad_state0, want_states, want_loss)
    check(lib().apg_cartpole_rollout_fwd_bwd(
        ptr(state0), ptr(actions), float(dt), ctypes.byref(params), B, H, lay,
        ptr(o["loss_partials"]), ptr(o["loss"]), ptr(o["grad_actions"]),
        ptr(o["grad_state0"]), ptr(o["states"]), stream_of(state0)),
        "apg_cartpole_rollout_fwd_bwd")
    return o


def cartpole_rollout_fwd(state0, actions, dt, params, layout="aos"):
    """No-grad H-step unroll (apg_cartpole_rollout_fwd): states [B,H,4]."""
    lay = _layout(layout)
    require_device(state0, actions)
    B, H, A = _seq_shape(actions, lay)
    if A != 1 or _state_batch(state0, lay) != B:
        raise ValueError("inconsistent rollout shapes")
    states = torch.empty(_states_shape(B, H, 4, lay), dtype=torch.float32,
                         device=state0.device)
    check(lib().apg_cartpole_rollout_fwd(
        ptr(state0), ptr(actions), float(dt), ctypes.byref(params), B, H, lay,
        ptr(states), stream_of(state0)), "apg_cartpole_rollout_fwd")
    return states


class _CartpoleRolloutLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state0, action_seq, dt, params):
        s, a = _f32c(state0), _f32c(action_seq)
        res = cartpole_rollout_fwd_bwd(
            s, a, dt, params, want_grad_state0=ctx.needs_input_grad[0])
        ctx.save_for_backward(res["grad_actions"], res["grad_state0"])
        return res["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        ga, gs = ctx.saved_tensors
        ga = ga * g if ctx.needs_input_grad[1] else None
        gs = gs * g if (gs is not None and ctx.needs_input_grad[0]) else None
        return gs, ga, None, None


def cartpole_rollout_loss(state0, action_seq, dt, params):
    return _CartpoleRolloutLoss.apply(state0, action_seq, dt, params)


# ------------------------------------------------------ pre-bound launches
def reduce_loss_partials(partials, loss=None):
    """loss[0] = fixed-order sum of `partials` (apg_reduce_loss_partials)."""
    require_device(partials, loss)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=partials.device)
    check(lib().apg_reduce_loss_partials(
        ptr(partials), partials.numel(), ptr(loss), stream_of(partials)),
        "apg_reduce_loss_partials")
    return loss


class RolloutPlan:
    """A fused-rollout launch with every ctypes argument bound once.

    `launch()` is a single foreign call that enqueues the kernel(s) on the
    stream that was current when the plan was made - the hot loop of a
    trainer / bench pays no per-step Python argument marshalling, no
    allocation and no host synchronisation.  The input tensors are held by
    reference: refill them in place (copy_) between launches.

    loss_mode:
      "eager"    the launch is followed by the small fixed-order reduction
                 kernel; `out["loss"]` is valid after this launch.
      "deferred" the launch only leaves per-wave partials; the NEXT launch
                 given `after=<this plan>` folds their reduction into its own
                 kernel (ApgDeferredLoss), `flush()` reduces the last one.
      "none"     partials only.
    """

    def __init__(self, system, state0, actions, ref, dt, params, weights=None,
                 layout="soa", want_grad_state0=False, want_states=False,
                 loss_mode="eager", out=None):
        """`out`: optional pre-allocated output tensors (keys as returned by
        quad_/wing_rollout_fwd_bwd) - e.g. slices of one large allocation."""
        if loss_mode not in ("eager", "deferred", "none"):
            raise ValueError("loss_mode must be eager, deferred or none")
        self.system = system
        self.layout = layout
        self.loss_mode = loss_mode
        self.inputs = (state0, actions, ref)
        kw = dict(layout=layout, want_grad_state0=want_grad_state0,
                  want_states=want_states, want_loss=True, out=out)
        # one eager call allocates the outputs and validates the shapes
        if system == "quad":
            self.out = quad_rollout_fwd_bwd(state0, actions, ref, dt, params,
                                            weights, **kw)
            weights = weights or quad_loss_weights()
            fn = lib().apg_quad_rollout_fwd_bwd
        elif system == "wing":
            self.out = wing_rollout_fwd_bwd(state0, actions, ref, dt, params,
                                            weights, **kw)
            weights = weights or wing_loss_weights()
            fn = lib().apg_wing_rollout_fwd_bwd
        else:
            raise ValueError("system must be 'quad' or 'wing'")
        lay = _layout(layout)
        B, H, _ = _seq_shape(actions, lay)
        self.B, self.H = B, H
        self._keep = [params, weights]
        o = self.out
        head = [ptr(state0), ptr(actions), ptr(ref)]
        if system == "quad":
            head.append(_seq_shape(ref, lay)[2])
        self._fn = fn
        self._head = head + [
            float(dt), ctypes.byref(params), ctypes.byref(weights), B, H, lay,
            ptr(o["loss_partials"]),
            ptr(o["loss"]) if loss_mode == "eager" else None,
            ptr(o["grad_actions"]), ptr(o["grad_state0"]), ptr(o["states"])]
        self._stream = stream_of(state0)
        self._args = tuple(self._head + [None, self._stream])
        self._args_after = {}

    def _deferred_args(self, prev):
        key = id(prev)
        args = self._args_after.get(key)
        if args is None:
            d = _capi.ApgDeferredLoss(
                ptr(prev.out["loss_partials"]),
                prev.out["loss_partials"].numel(), ptr(prev.out["loss"]))
            self._keep.append(d)
            args = tuple(self._head + [ctypes.byref(d), self._stream])
            self._args_after[key] = args
        return args

    def launch(self, after=None):
        """Enqueue the rollout.  `after`: an earlier-launched *deferred* plan
        (not this one) whose loss this launch should reduce on the side."""
        args = self._args if after is None else self._deferred_args(after)
        code = self._fn(*args)
        if code != 0:
            check(code, "fused rollout launch")
        return self.out

    def flush(self):
        """Reduce this plan's partials now (end of a deferred chain)."""
        reduce_loss_partials(self.out["loss_partials"], self.out["loss"])
        return self.out["loss"]


# ------------------------------------- operand range of the in-kernel policies
# The policy layers inside the kernels multiply fp32 operands as two fp16 terms
# (csrc/policy_mfma16.h): a first-layer input of magnitude >= 65 504 would
# become inf in its high term.  Cotangents are rescaled per trajectory inside
# the kernels, activations are bounded by tanh / relu of bounded values; what
# the CALLER controls are the tensors the first layers read - the normalised
# state features, the initial state (velocities, body rates; in the recurrent
# modes the features are rebuilt from the evolving state, which a rollout of
# H = 10 steps moves by < 30 m/s and < 30 m) and the reference windows (metres,
# m/s).  Contract (include/apg.h): every such value is finite with magnitude
# < 2^14 = 16 384.  It is enforced here, on the host: the first time a tensor
# object (in this in-place version) is seen, its largest magnitude is read
# back ONCE - a data set that is stepped on for epochs costs one check, a
# captured step graph none (its warm-up steps have checked) - and a violation
# raises ValueError instead of training on inf.  Tiny values lose nothing that
# matters: the low term keeps an ABSOLUTE accuracy of 2^-25.
POLICY_INPUT_LIMIT = 16384.0
CHECK_POLICY_INPUT_RANGE = True


class _RangeGuard:
    def __init__(self, slots=32):
        self.slots, self.seen = slots, []

    def __call__(self, what, **tensors):
        if not CHECK_POLICY_INPUT_RANGE:
            return
        import weakref
        fresh = []
        for name, t in tensors.items():
            if t is None or not torch.is_tensor(t) or t.numel() == 0:
                continue
            if any(r() is t and v == t._version for r, v in self.seen):
                continue
            fresh.append((name, t))
        if not fresh or (fresh[0][1].is_cuda and torch.cuda.is_current_stream_capturing()):
            return
        amax = torch.stack([t.detach().abs().amax().float() for _, t in fresh]).tolist()
        for (name, t), m in zip(fresh, amax):
            if not m < POLICY_INPUT_LIMIT:          # also catches NaN / inf
                raise ValueError(
                    f"{what}: `{name}` holds a value of magnitude {m:g}; the "
                    f"in-kernel policy takes finite inputs below "
                    f"{POLICY_INPUT_LIMIT:g} (fp16-split operands, include/apg.h "
                    "\"operand range\") - normalise the data or use the "
                    "PyTorch policy path (fused_policy = False)")
            self.seen.append((weakref.ref(t), t._version))
        self.seen = [e for e in self.seen if e[0]() is not None][-self.slots:]


_guard_policy_inputs = _RangeGuard()


# ----------------------------------------- fused LSTM-policy unroll (K7)
_GEMM_WGS = None   # None: sized from the LDS footprint (workgroups per CU x 256)
_bdesc_cache = {}


class _BDesc:
    """Column descriptor of apg_planes_gemm: device int32 [3, J] = plane offset
    and the two segment strides of every column, the offsets RELATIVE to
    `base_plane` (the first plane the product touches): the kernel addresses
    its operands with unsigned 32-bit byte offsets, so B is handed over from
    that plane on and only the span the product really uses has to stay below
    4 GiB."""
    __slots__ = ("tensor", "base_plane", "J", "host")

    def __init__(self, tensor, base_plane, J, host):
        self.tensor, self.base_plane, self.J, self.host = tensor, base_plane, J, host

    def span(self, S, sdiv):
        """Planes from base_plane up to the last one segment s < S touches."""
        q, r = (S - 1) // sdiv, min(S, sdiv) - 1
        return 1 + max(o + q * a + r * b for o, a, b in zip(*self.host))

    @property
    def shape(self):
        return self.tensor.shape

    def data_ptr(self):
        return self.tensor.data_ptr()


def make_bdesc(dev, offsets, stride1=0, stride2=0, key=None):
    """Descriptor for the columns `offsets` (plane indices into B) with the
    per-column strides per s // sdiv and s % sdiv (scalars broadcast, >= 0).
    Cached per device when a `key` is given (building it is a host-to-device
    copy)."""
    k = (str(dev), key)
    if key is not None and k in _bdesc_cache:
        return _bdesc_cache[k]
    offsets = [int(o) for o in offsets]
    J = len(offsets)
    bc = lambda v: [int(x) for x in v] if hasattr(v, "__len__") else [int(v)] * J
    s1, s2 = bc(stride1), bc(stride2)
    if min(s1) < 0 or min(s2) < 0 or min(offsets) < 0:
        raise ValueError("make_bdesc: plane offsets and strides must be >= 0")
    base = min(offsets)
    host = ([o - base for o in offsets], s1, s2)
    t = torch.tensor(list(host), dtype=torch.int32, device=dev)
    d = _BDesc(t, base, J, host)
    if key is not None:
        _bdesc_cache[k] = d
    return d


def _b_operand(Bp, bdesc, N, S, sdiv):
    """(pointer, plane count) of B as the kernel gets it: the planes the
    product touches, from the descriptor's base plane on."""
    planes = bdesc.span(S, sdiv)
    if bdesc.base_plane + planes > Bp.numel() // N:
        raise ValueError("planes_gemm: descriptor points beyond B")
    return Bp.data_ptr() + bdesc.base_plane * N * 4, planes


def planes_gemm(A, M, S, Bp, bdesc, with_ones=True, sdiv=1, N=None, out=None,
                bias_out=None):
    """C[m][j] = sum_{s,n} A[m*S+s][n] * Bp[bdesc[0][j] + (s//sdiv)*bdesc[1][j]
    + (s%sdiv)*bdesc[2][j]][n] on the matrix cores (apg_planes_gemm).  A, Bp
    are fp32 tensors whose rows ("planes") hold N contiguous floats; `N`
    defaults to A.shape[1] (pass it to re-interpret a buffer as shorter
    planes).  `bdesc`: make_bdesc(...).  `out`: optional [M, >= J+ones] view
    (row stride = out.stride(0)) written in place; returns [M, J+ones].
    `bias_out` [M] (with_ones): the row sums go there instead of column J, so
    weight and bias gradients are separate contiguous tensors; returns [M, J]."""
    require_device(A, Bp)
    N = A.shape[1] if N is None else N
    J = bdesc.J
    Jt = J + int(with_ones)
    wgs = _GEMM_WGS or lib().apg_planes_gemm_default_wgs(M, S, J, int(with_ones))
    ws = torch.empty(lib().apg_planes_gemm_workspace_floats(
        M, J, int(with_ones), wgs), dtype=torch.float32, device=A.device)
    Jc = J if (bias_out is not None and with_ones) else Jt   # columns of C
    if out is None:
        out = torch.empty(M, Jc, dtype=torch.float32, device=A.device)
    if out.stride(1) != 1 or out.shape[0] < M or out.shape[1] < Jc:
        raise ValueError("planes_gemm: out must be [>=M, >=J+ones], unit column stride")
    b_ptr, b_planes = _b_operand(Bp, bdesc, N, S, sdiv)
    check(lib().apg_planes_gemm(
        ptr(A), M, S, b_ptr, bdesc.data_ptr(), J, sdiv, int(with_ones),
        b_planes, N, ptr(ws), wgs, out.data_ptr(), out.stride(0),
        ptr(bias_out), stream_of(A)), "apg_planes_gemm")
    return out[:M, :Jc]


_GROUP_WGS = 256
# the grouped launch pair wins while launch latency dominates; from ~8 k columns
# per product on, one register-streaming launch per product is faster (B =
# 65 536: 0.42 -> 0.36 ms per concurrent training step)
_GROUPED_MAX_COLUMNS = 8 * 8192


def _gemm_problems(problems):
    probs = (_capi.ApgGemmProblem * len(problems))()
    for q, d in zip(probs, problems):
        A, Bp, out = d["A"], d["Bp"], d["out"]
        require_device(A, Bp)
        N = d.get("N") or A.shape[1]
        ones = int(d.get("with_ones", True))
        bias = d.get("bias_out") if ones else None
        J = d["bdesc"].J
        if out.stride(1) != 1 or out.shape[0] < d["M"] \
                or out.shape[1] < (J if bias is not None else J + ones):
            raise ValueError("planes_gemm: bad `out` view")
        q.A, q.bdesc, q.C = ptr(A), d["bdesc"].data_ptr(), out.data_ptr()
        q.B, q.b_planes = _b_operand(Bp, d["bdesc"], N, d["S"], d.get("sdiv", 1))
        q.bias_out = ptr(bias)
        q.N, q.M, q.S, q.J = N, d["M"], d["S"], J
        q.sdiv, q.with_ones = d.get("sdiv", 1), ones
        q.ldc = out.stride(0)
    return probs


def planes_gemm_grouped(problems):
    """Several planes_gemm products in one launch pair
    (apg_planes_gemm_grouped).  `problems`: dicts with the planes_gemm
    arguments A, M, S, Bp, bdesc, out and optionally with_ones (True), sdiv
    (1), N, bias_out; each M <= 64 and J + ones <= 128, at most 8."""
    probs = _gemm_problems(problems)
    A0 = problems[0]["A"]
    ws = torch.empty(_GROUP_WGS * 64 * 128, dtype=torch.float32, device=A0.device)
    check(lib().apg_planes_gemm_grouped(probs, len(problems), ptr(ws), _GROUP_WGS,
                                        stream_of(A0)), "apg_planes_gemm_grouped")


def planes_gemm_multi(problems):
    """The same products, one launch each (own tile shape and occupancy) and
    ONE second-stage launch for all (apg_planes_gemm_multi); at most 8."""
    probs = _gemm_problems(problems)
    A0 = problems[0]["A"]
    n = len(problems)
    ws = torch.empty(lib().apg_planes_gemm_multi_workspace_floats(probs, n),
                     dtype=torch.float32, device=A0.device)
    check(lib().apg_planes_gemm_multi(probs, n, ptr(ws), stream_of(A0)),
          "apg_planes_gemm_multi")


def _run_products(problems):
    """A training step's weight-gradient products.  Few columns (small
    batches in the concurrent mode, one column per trajectory): launch latency
    dominates, so all products share one launch pair.  Otherwise each product
    gets its own launch with the kernel, tile shape and occupancy that fit it,
    and they share the second-stage launch."""
    n_max = max(d.get("N") or d["A"].shape[1] for d in problems)
    total = sum((d.get("N") or d["A"].shape[1]) * d["S"] for d in problems)
    fits = all(d["bdesc"].J + int(d.get("with_ones", True)) <= 128
               for d in problems) and len(problems) <= 8
    if fits and total <= _GROUPED_MAX_COLUMNS and n_max <= 131072:
        return planes_gemm_grouped(problems)
    for lo in range(0, len(problems), 8):
        planes_gemm_multi(problems[lo:lo + 8])


def _flat_grads(dev, shapes):
    """One flat fp32 buffer holding a contiguous gradient tensor per entry of
    `shapes` (name -> shape) plus ONE trailing slot (for the loss: the whole
    buffer is then the message of the data-parallel all-reduce); returns
    (flat, {name: view})."""
    sizes = [int(torch.Size(sh).numel()) for sh in shapes.values()]
    flat = torch.empty(sum(sizes) + 1, dtype=torch.float32, device=dev)
    views = {k: v.view(sh) for (k, sh), v in
             zip(shapes.items(), flat[:-1].split(sizes))}
    return flat, views


def _soa_source(t, out, index):
    """(tensor kept alive, R, ld, B, out) of one to_soa conversion."""
    inner = torch.Size(t.shape[1:])
    R = int(inner.numel())
    # no .contiguous() up front: a leading slice of longer rows (the whole
    # data set's ref[:, :H] in the indexed paths) must be read IN PLACE
    # through its row stride, not copied per minibatch
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    dense_rows = all(t.stride(i + 1) == s_ for i, s_ in
                     enumerate(torch.empty(inner, device="meta").stride()))
    if not dense_rows or (t.shape[0] > 1 and t.stride(0) < R):
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else R
    B = t.shape[0] if index is None else index.numel()
    if index is not None and (index.dtype != torch.int64 or not index.is_cuda
                              or not index.is_contiguous()):
        raise ValueError("to_soa: index must be a contiguous int64 device tensor")
    if out is None:
        out = torch.empty(*inner, B, dtype=torch.float32, device=t.device)
    if not (t.is_cuda and out.is_cuda and out.is_contiguous()):
        raise ValueError("to_soa: device tensors, contiguous destination "
                         "(there is no CPU fallback)")
    return t, R, ld, B, out


def _is_plane_view(t):
    """[B, R] fp32 device tensor that is the transposed view of contiguous
    [R][B] planes (LSTM_NEW.reset_hidden_state draws them that way)."""
    return (t.dim() == 2 and t.is_cuda and t.dtype == torch.float32
            and t.shape[0] > 1 and t.stride() == (1, t.shape[0]))


def to_soa(t, out=None, index=None):
    """[B, ...] fp32 -> [..., B] planes (apg_to_soa: both sides coalesced).  A
    leading slice of longer rows (e.g. ref[:, :H]) is read in place through its
    row stride.  `index` (int64 device tensor [B]): gather rows t[index] in the
    same pass (t is then the whole data set).  `out`: optional contiguous
    destination."""
    t, R, ld, B, out = _soa_source(t, out, index)
    check(lib().apg_to_soa(ptr(t), None if index is None else index.data_ptr(), B, R,
                           ld, ptr(out), stream_of(t)), "apg_to_soa")
    return out


def to_soa_multi(items, index=None):
    """Several to_soa conversions of the same batch in ONE launch
    (apg_to_soa_multi): `items` = [(tensor, out-or-None), ...]; returns the
    list of outputs."""
    prepared = [_soa_source(t, out, index) for t, out in items]
    B = prepared[0][3]
    if any(p[3] != B for p in prepared):
        raise ValueError("to_soa_multi: all tensors must have the same batch")
    outs = [p[4] for p in prepared]
    for lo in range(0, len(prepared), 6):        # APG_SOA_MAX_ITEMS
        part = prepared[lo:lo + 6]
        arr = (_capi.ApgSoaItem * len(part))()
        for q, (t, R, ld, _, out) in zip(arr, part):
            q.src, q.dst, q.R, q.ld = ptr(t), ptr(out), R, ld
            q.index = None if index is None else index.data_ptr()
        check(lib().apg_to_soa_multi(arr, len(part), B, stream_of(part[0][0])),
              "apg_to_soa_multi")
    return outs


class _SoaPlan:
    """A to_soa_multi call with everything but the index batch frozen: the
    argument structs are built once and only their index pointer changes -
    run_epoch's per-batch gather is then one ctypes call (~10 us of host time)
    instead of ~80 us of shape / stride bookkeeping per batch."""

    def __init__(self, key, chunks, B, outs, keep):
        self.key, self.chunks, self.B, self.outs, self.keep = key, chunks, B, outs, keep

    @staticmethod
    def key_of(items):
        return tuple((t.data_ptr(), tuple(t.shape), t.stride(), t.dtype,
                      None if o is None else o.data_ptr()) for t, o in items)

    def run(self, index):
        if index.numel() != self.B or index.dtype != torch.int64 or not index.is_cuda \
                or not index.is_contiguous():
            raise ValueError("to_soa: index must be a contiguous int64 device tensor "
                             "of the planned batch size")
        ip = index.data_ptr()
        st = stream_of(index)
        for arr, n in self.chunks:
            for q in arr:
                q.index = ip
            check(lib().apg_to_soa_multi(arr, n, self.B, st), "apg_to_soa_multi")
        return self.outs


def to_soa_multi_planned(items, index, plan=None):
    """to_soa_multi(items, index) for a gather that is repeated with the same
    sources and destinations: returns (outs, plan); hand `plan` back in to
    skip the per-call bookkeeping (it is rebuilt when the tensors differ)."""
    key = _SoaPlan.key_of(items)
    if plan is not None and plan.key == key and plan.B == index.numel():
        return plan.run(index), plan
    prepared = [_soa_source(t, out, index) for t, out in items]
    B = prepared[0][3]
    if any(p[3] != B for p in prepared):
        raise ValueError("to_soa_multi: all tensors must have the same batch")
    outs = [p[4] for p in prepared]
    chunks = []
    for lo in range(0, len(prepared), 6):        # APG_SOA_MAX_ITEMS
        part = prepared[lo:lo + 6]
        arr = (_capi.ApgSoaItem * len(part))()
        for q, (t, R, ld, _, out) in zip(arr, part):
            q.src, q.dst, q.R, q.ld = ptr(t), ptr(out), R, ld
        chunks.append((arr, len(part)))
    # the destinations are part of the key from now on
    plan = _SoaPlan(_SoaPlan.key_of([(t, o) for (t, _), o in zip(items, outs)]),
                    chunks, B, outs, [p[0] for p in prepared])
    outs = plan.run(index)
    # A plan freezes the SOURCE pointers.  Where _soa_source had to make a copy
    # (a data set that is not float32, rows that are not dense) that pointer is
    # the one-time copy: replaying it after an in-place refresh of the data set
    # (resample_data, self-play slots) would gather stale rows.  Such sources
    # are converted afresh on every call instead - no plan is handed back.
    if any(p[0].data_ptr() != t.data_ptr() or t.dtype != torch.float32
           for p, (t, _) in zip(prepared, items)):
        return outs, None
    return outs, plan


class _StaticPlanes:
    """Plane-layout copies of WHOLE input tensors that have not changed since
    the last step (a trainer stepping on its resident shard, bench.py): the
    layout change (apg_to_soa, 45-70 us at B = 65 536) is then done once, not
    per step.  Opt-in by the direct-gradient entry points only (no autograd
    tape holds the buffers), keyed on the identity of the caller's tensor
    OBJECTS (weak references) and their in-place version counters - a fresh
    minibatch tensor, or `t[:n] = ...` on the data set, can never hit."""

    def __init__(self, slots=4):
        self.slots, self.entries = slots, []

    def _match(self, e, tag, tensors):
        return (e[0] == tag and len(e[1]) == len(tensors) and all(
            r() is t and v == t._version for r, v, t in zip(e[1], e[2], tensors)))

    def lookup(self, tag, tensors):
        for e in self.entries:
            if self._match(e, tag, tensors):
                return e[3]
        return None

    def store(self, tag, tensors, value):
        import weakref
        self.entries = [e for e in self.entries
                        if not self._match(e, tag, tensors)][-(self.slots - 1):]
        self.entries.append((tag, [weakref.ref(t) for t in tensors],
                             [t._version for t in tensors], value))
        return value


_STATIC_PLANES = _StaticPlanes()


def static_plane_refs():
    """Strong references to every kept plane copy (a captured step graph reads
    them by address: its owner keeps them alive past an eviction here)."""
    return [e[3] for e in _STATIC_PLANES.entries]


def _ref_and_states(in_ref, state0, B, H, index=None, also=(), out=None):
    """One buffer for everything the conv-weight product reads as B operand:
    planes [0, 2H*9) = the reference tensor [2H][9][B], planes [2H*9, +(H+1)*12)
    = [state0; states of the rollout] ([H+1][12][B], the kernels write the H
    new states in place).  `also`: further [B, ...] tensors of the same
    (indexed) batch converted by the same launch.  `out`: the tuple a previous
    call returned, refilled in place (same batch size).  Returns (buffer,
    in_ref view, state0 view, states view, *planes of `also`)."""
    if out is None:
        buf = torch.empty(2 * H * 9 + (H + 1) * 12, B, dtype=torch.float32,
                          device=state0.device)
        extra = [None] * len(also)
    else:
        buf, extra = out[0], list(out[4:])
    if out is not None and index is not None:
        # repeated gather into the same buffers (run_epoch): planned call
        items = [(in_ref[:, :2 * H], out[1]), (state0, out[2])] + list(zip(also, extra))
        _, plan = to_soa_multi_planned(items, index, getattr(buf, "_apg_plan", None))
        buf._apg_plan = plan
        return out        # the same tensor objects, refilled
    inr = buf[:2 * H * 9].view(2 * H, 9, B)
    st_all = buf[2 * H * 9:].view(H + 1, 12, B)
    outs = to_soa_multi([(in_ref[:, :2 * H], inr), (state0, st_all[0])] +
                        list(zip(also, extra)), index=index)
    if out is not None:
        return out        # the same tensor objects, refilled
    return (buf, inr, st_all[0], st_all[1:], *outs[2:])


def quad_recurrent_prepare(state0, in_ref, ref, index=None, out=None, H=10):
    """The layout change (and, with `index`, the minibatch gather) of an
    autoregressive / LSTM step's inputs, as its own call: run_epoch issues it
    one batch ahead on a side stream (TrainBase._pipelined_epoch) and hands the
    result to quad_mlp_rollout_grads / quad_lstm_rollout_grads as `prepared`.
    `out`: a previous result to refill (same batch size)."""
    B = state0.shape[0] if index is None else index.numel()
    if in_ref.shape[1] < 2 * H or in_ref.shape[2] != 9 or ref.shape[1] < H:
        raise ValueError("in_ref [B,2H,9] and ref [B,>=H,9|6] with H = 10")
    _guard_policy_inputs("fused recurrent unroll", state0=state0, in_ref=in_ref)
    return _ref_and_states(_f32c(in_ref), _f32c(state0), B, H, index,
                           also=(ref[:, :H],), out=out)


def quad_concurrent_prepare(normed, state0, in_ref, ref, index=None, out=None, H=10):
    """The same for the concurrent step: (acts [431 + 9 H][B] with the feature
    planes 0..14 and the in_ref planes 431.. filled, state0 planes, ref
    planes)."""
    B = state0.shape[0] if index is None else index.numel()
    if in_ref.shape[1] < H or in_ref.shape[2] != 9 or ref.shape[1] < H \
            or normed.shape[1] != 15:
        raise ValueError("normed [B,15], in_ref [B,>=H,9], ref [B,>=H,9|6], H = 10")
    _guard_policy_inputs("fused concurrent step", normed=normed, in_ref=in_ref)
    if out is None:
        acts = torch.empty(431 + H * 9, B, dtype=torch.float32, device=state0.device)
        s0 = rf = None
    else:
        acts, s0, rf = out
        if index is not None:    # repeated gather (run_epoch): planned call
            if not hasattr(acts, "_apg_views"):
                acts._apg_views = (acts[:15], acts[431:].view(H, 9, B))
            feat, inr = acts._apg_views
            _, plan = to_soa_multi_planned(
                [(normed, feat), (in_ref[:, :H], inr), (state0, s0), (ref[:, :H], rf)],
                index, getattr(acts, "_apg_plan", None))
            acts._apg_plan = plan
            return out
    _, _, s0, rf = to_soa_multi(
        [(normed, acts[:15]), (in_ref[:, :H], acts[431:].view(H, 9, B)), (state0, s0),
         (ref[:, :H], rf)], index=index)
    return out if out is not None else (acts, s0, rf)


_CONV_DIAG_PLANES = 720      # 20 ch x 2 half-waves x 13 diagonals + 20 ch x H


def _conv_diag_problems(cv, refbuf, B, H, w_out, b_out):
    """The conv-weight gradient of the recurrent unrolls from the DIAGONAL sums
    the reverse sweep leaves (csrc/lstm.hip kConvP): cv [720][B] =
    G[ch][hi][tau] (tau = k + pos - 4 hi; the window of (step, position, tap) is
    reference row k + pos + tap) followed by P[ch][k] = sum_pos d[ch][pos][k].
      dW[ch][c][t] = sum G[ch][hi][tau] . ref[4 hi + tau + t][c]
                     - (c < 3) sum_k P[ch][k] . pos_k[c]
    as two segmented planes_gemm problems reading the reference windows and the
    position planes of `refbuf` in place; the first one writes straight into
    `w_out` (d conv_ref.weight [20,9,3]), finish() subtracts the second.
    Returns (problems, finish)."""
    assert H == 10
    dev = cv.device
    offs = [t * 9 + c for c in range(9) for t in range(3)]
    g_desc = make_bdesc(dev, offs, 36, 9, key=("conv_diag", H))
    p_desc = make_bdesc(dev, [2 * H * 9 + q for q in range(3)], 12, 0,
                        key=("conv_pos", H))
    cg = w_out.view(20, 27)
    cp = torch.empty(20, 3, dtype=torch.float32, device=dev)
    probs = [
        dict(A=cv[:520], M=20, S=26, Bp=refbuf, bdesc=g_desc, sdiv=13, N=B, out=cg,
             with_ones=False),
        dict(A=cv[520:], M=20, S=H, Bp=refbuf, bdesc=p_desc, sdiv=1, N=B, out=cp,
             bias_out=b_out)]

    def finish():
        cg[:, :9].view(20, 3, 3).sub_(cp[:, :, None])
    return probs, finish


def _lstm_batch_rows(state0, in_ref, ref, index, H):
    """Round 6: with resident tables and a row index the LSTM sweeps read the data
    set's tensors through the index themselves (apg_quad_lstm_rollout_fwd_rows /
    _bwd_rows) instead of a gather pass.  Returns the ApgBatchRows of an indexed
    LSTM minibatch, or None where the tensors are not what the kernels read in
    place (then the gather pass runs)."""
    ok = lambda t: (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                    and t.numel() * 4 < (1 << 32) - 64)
    n = state0.shape[0]
    if not (all(ok(t) for t in (state0, in_ref, ref)) and state0.dim() == 2
            and state0.shape[1] == 12 and in_ref.dim() == 3 and in_ref.shape[0] == n
            and in_ref.shape[1] >= 2 * H and in_ref.shape[2] == 9 and ref.dim() == 3
            and ref.shape[0] == n and ref.shape[1] >= H and ref.shape[2] in (6, 9)
            and index.is_cuda and index.dtype == torch.int64 and index.is_contiguous()):
        return None
    r = _capi.ApgBatchRows(index=index.data_ptr(), normed=None, state0=ptr(state0),
                           in_ref=ptr(in_ref), ref=ptr(ref), ld_normed=0,
                           ld_state0=state0.stride(0), ld_in_ref=in_ref.stride(0),
                           ld_ref=ref.stride(0), n_rows=n, running_loss=None)
    r._keep = (state0, in_ref, ref, index)
    return r


class _QuadLstmRolloutLoss(torch.autograd.Function):
    """loss of the LSTM-mode unroll with the policy inside the kernel.

    forward : apg_quad_lstm_rollout_fwd then apg_quad_lstm_rollout_bwd (the
              reverse sweep runs right away - the loss only exists after it).
    backward: turns the saved per-(step, trajectory) cotangent planes into the
              parameter gradients with two trajectory-major kernels
              (apg_quad_lstm_wgrads = apg_quad_lstm_gate_wgrad + apg_quad_lstm_conv_wgrad; round 6).
    Inputs are the reference's tensors: state0 [B,12], in_ref [B,2H,9],
    ref [B,>=H,9], h0 / c0 [B,8] and the LSTM_NEW parameters."""

    @staticmethod
    def forward(ctx, state0, in_ref, ref, h0, c0, conv_w, conv_b, w_ih, w_hh,
                b_ih, b_hh, w_out, b_out, dt, params, weights, index=None):
        H = 10
        if w_ih.shape != (32, 175) or conv_w.shape != (20, 9, 3):
            raise ValueError("fused path needs LSTM_NEW(15, 10, 9, 4, conv=1)")
        prepared = getattr(ctx, "prepared", None)
        if prepared is not None:     # quad_recurrent_prepare ran ahead
            refbuf, inr, s0, states, rf = prepared
            B = s0.shape[-1]
        else:
            B = state0.shape[0] if index is None else index.numel()
            if in_ref.shape[1] < 2 * H or in_ref.shape[2] != 9 or ref.shape[1] < H:
                raise ValueError("in_ref [B,2H,9] and ref [B,>=H,9|6] with H = 10")
            _guard_policy_inputs("fused LSTM unroll", state0=state0, in_ref=in_ref)
            src = getattr(ctx, "static_src", None) if index is None else None
            hit = _STATIC_PLANES.lookup("recurrent", src) if src else None
            rows = None
            if (index is not None and getattr(ctx, "lstm_tables", None) is not None
                    and getattr(ctx, "rows_in_kernel", True)):
                rows = _lstm_batch_rows(state0, in_ref, ref, index, H)
            if rows is not None:
                # round 6: the sweeps read the data set's rows through the index
                # themselves; the forward sweep writes the planes its followers read
                refbuf = torch.empty(2 * H * 9 + (H + 1) * 12, B, dtype=torch.float32,
                                     device=state0.device)
                inr = refbuf[:2 * H * 9].view(2 * H, 9, B)
                st_all = refbuf[2 * H * 9:].view(H + 1, 12, B)
                s0, states, rf = st_all[0], st_all[1:], None
            else:
                refbuf, inr, s0, states, rf = hit or _ref_and_states(
                    _f32c(in_ref), _f32c(state0), B, H, index, also=(ref[:, :H],))
            if src and hit is None:
                _STATIC_PLANES.store("recurrent", src, (refbuf, inr, s0, states, rf))
        if prepared is not None:
            rows = None
        dev = s0.device
        if all(_is_plane_view(t) for t in (h0, c0)):
            h0s, c0s = h0.detach().t(), c0.detach().t()   # already [8][B] planes
        else:
            h0s, c0s = to_soa_multi([(h0, None), (c0, None)])
        pw = dict(
            conv_w=_f32c(conv_w), conv_b=_f32c(conv_b), w_ih=_f32c(w_ih),
            w_hh=_f32c(w_hh), b_ih=_f32c(b_ih), b_hh=_f32c(b_hh),
            w_out=_f32c(w_out), b_out=_f32c(b_out))
        pw = {k: v.contiguous() for k, v in pw.items()}
        require_device(s0, inr, *([] if rf is None else [rf]), h0s, c0s, *pw.values())
        pol = _capi.ApgLstmPolicy(**{k: ptr(v) for k, v in pw.items()})
        N = H * B
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        actions = new(H, 4, B)
        # one buffer for what the weight-gradient kernel reads as B operand
        # (apg_quad_lstm_gate_wgrad): state features (15 planes), h_prev / c_prev
        # (16), h_new (8); round 6: the 160 relu(conv) inputs are recomputed there
        acts = new(39, N)
        x, hc, hnew = acts[:15], acts[15:31], acts[31:39]
        gates = new(32, N)
        relu_mask = torch.empty(5, N, dtype=torch.int32, device=dev)
        st = stream_of(s0)
        # resident operand tables (round 6, the trainers' step): packed once, kept
        # current by the step's tail - the sweeps launch nothing but themselves
        tables = getattr(ctx, "lstm_tables", None)
        if tables is not None:
            tables.ensure(list(pw.values()), pol, st)
        if rows is not None:
            check(lib().apg_quad_lstm_rollout_fwd_rows(
                ctypes.byref(rows), ptr(h0s), ptr(c0s), float(dt), ctypes.byref(params),
                ptr(tables.fwd), B, H, ptr(s0), ptr(inr), ptr(states), ptr(actions), ptr(x),
                ptr(gates), ptr(hc), ptr(hnew), relu_mask.data_ptr(), st),
                "apg_quad_lstm_rollout_fwd_rows")
        elif tables is not None:
            check(lib().apg_quad_lstm_rollout_fwd_packed(
                ptr(s0), ptr(inr), ptr(h0s), ptr(c0s), float(dt),
                ctypes.byref(params), ptr(tables.fwd), B, H, ptr(states),
                ptr(actions), ptr(x), ptr(gates), ptr(hc), ptr(hnew),
                relu_mask.data_ptr(), st), "apg_quad_lstm_rollout_fwd_packed")
        else:
            ws = new(lib().apg_quad_lstm_workspace_floats())
            check(lib().apg_quad_lstm_rollout_fwd(
                ptr(s0), ptr(inr), ptr(h0s), ptr(c0s), float(dt),
                ctypes.byref(params), ctypes.byref(pol), B, H, ptr(states),
                ptr(actions), ptr(x), ptr(gates), ptr(hc), ptr(hnew),
                relu_mask.data_ptr(), ptr(ws), st), "apg_quad_lstm_rollout_fwd")
        partials = new(max(1, lib().apg_quad_lstm_loss_partials_count(B)))
        loss = new(1)
        d_gates, d_zout, d_conv = new(32, N), new(4, N), new(_CONV_DIAG_PLANES, B)
        cot_amax = new(max(1, lib().apg_quad_lstm_cot_amax_floats(B)))
        # optional input gradients (state0, h0, c0)
        g_s0 = new(12, B) if ctx.needs_input_grad[0] else None
        g_h0 = new(8, B) if ctx.needs_input_grad[3] else None
        g_c0 = new(8, B) if ctx.needs_input_grad[4] else None
        if rows is not None:
            check(lib().apg_quad_lstm_rollout_bwd_rows(
                ctypes.byref(rows), ref.shape[2], ptr(s0), ptr(states), ptr(actions),
                relu_mask.data_ptr(), ptr(gates), ptr(hc), float(dt), ctypes.byref(params),
                ctypes.byref(weights), ptr(tables.bwd), B, H, ptr(partials), None,
                ptr(d_gates), ptr(d_zout), ptr(d_conv), ptr(g_s0), ptr(g_h0), ptr(g_c0),
                ptr(cot_amax), st), "apg_quad_lstm_rollout_bwd_rows")
            ctx.lstm_tail = (partials, loss, pw)
        elif tables is not None:
            # (loss = NULL: the step's tail sums the partials)
            check(lib().apg_quad_lstm_rollout_bwd_packed(
                ptr(s0), ptr(states), ptr(actions), ptr(rf), rf.shape[1],
                relu_mask.data_ptr(), ptr(gates), ptr(hc), float(dt),
                ctypes.byref(params), ctypes.byref(weights), ptr(tables.bwd), B, H,
                ptr(partials), None, ptr(d_gates), ptr(d_zout), ptr(d_conv), ptr(g_s0),
                ptr(g_h0), ptr(g_c0), ptr(cot_amax), st), "apg_quad_lstm_rollout_bwd_packed")
            ctx.lstm_tail = (partials, loss, pw)
        else:
            check(lib().apg_quad_lstm_rollout_bwd(
                ptr(s0), ptr(states), ptr(actions), ptr(rf), rf.shape[1],
                relu_mask.data_ptr(), ptr(gates), ptr(hc), float(dt),
                ctypes.byref(params),
                ctypes.byref(weights), ctypes.byref(pol), B, H, ptr(partials),
                ptr(loss), ptr(d_gates), ptr(d_zout), ptr(d_conv), ptr(g_s0),
                ptr(g_h0), ptr(g_c0), ptr(cot_amax), ptr(ws), st), "apg_quad_lstm_rollout_bwd")
        ctx.save_for_backward(refbuf, acts, d_gates, d_zout, d_conv, cot_amax, *pw.values())
        ctx.input_grads = (g_s0, g_h0, g_c0)
        ctx.mark_non_differentiable(states, actions)
        ctx.dims = (B, H)
        return loss.reshape(()), states, actions

    @staticmethod
    def backward(ctx, g, _gs, _ga):
        flat, gr = _lstm_param_grads(ctx.saved_tensors, ctx.dims)
        flat *= g
        grads = [gr[k] for k in ("conv_ref.weight", "conv_ref.bias", "lstm.weight_ih",
                                 "lstm.weight_hh", "lstm.bias_ih", "lstm.bias_hh",
                                 "fc_out.weight", "fc_out.bias")]
        g_s0, g_h0, g_c0 = (None if v is None else v.t() * g for v in ctx.input_grads)
        return (g_s0, None, None, g_h0, g_c0, *grads, None, None, None)


class LstmResidentTables:
    """The operand tables of the two LSTM sweeps (csrc/lstm.hip: fp16-split
    A-operand blocks + small fp32 tables, 33 + 29 KB) in buffers that outlive the
    step: packed by ONE launch when the parameters are not the ones the tables
    were made from (in-place version counter and storage address of the eight
    tensors), refreshed by apg_quad_lstm_step_tail after its update.  A write
    through `p.data` is not seen: `invalidate()` (TrainBase.run_epoch calls it at
    the start of every epoch), as for the concurrent step's plan."""

    def __init__(self, dev):
        n = lib().apg_quad_lstm_tables_floats
        self.fwd = torch.empty(n(0), dtype=torch.float32, device=dev)
        self.bwd = torch.empty(n(1), dtype=torch.float32, device=dev)
        self.key = None
        self.packs = 0

    @staticmethod
    def _key(tensors):
        return [(t._version, t.data_ptr()) for t in tensors]

    def ensure(self, tensors, pol, st):
        key = self._key(tensors)
        # (a replayed graph runs no Python: a captured step always packs)
        if key != self.key or torch.cuda.is_current_stream_capturing():
            check(lib().apg_quad_lstm_pack_tables(ctypes.byref(pol), ptr(self.fwd),
                                                  ptr(self.bwd), st),
                  "apg_quad_lstm_pack_tables")
            self.packs += 1
            self.key = key

    def refreshed(self, tensors):
        """The tail has packed the tables from `tensors` as they are now."""
        self.key = self._key(tensors)

    def invalidate(self):
        self.key = None


_LSTM_TABLES = None


def lstm_resident_tables(net, dev):
    """The LstmResidentTables of `net` (made on first use)."""
    global _LSTM_TABLES
    if _LSTM_TABLES is None:
        import weakref
        _LSTM_TABLES = weakref.WeakKeyDictionary()
    ent = _LSTM_TABLES.get(net)
    if ent is None or ent.fwd.device != dev:
        ent = _LSTM_TABLES[net] = LstmResidentTables(dev)
    return ent


def _lstm_param_grads(saved, dims, tail=None):
    """Weight gradients of the fused LSTM unroll from the saved planes: two
    trajectory-major kernels (gate / head weights with the conv inputs recomputed,
    conv weights from the diagonal sums); every gradient is a contiguous view of
    one flat buffer (returned first), keyed by LSTM_NEW parameter name.
    tail = (partials, loss, {C name: parameter}, tables, update): what follows
    the products is ONE launch (apg_quad_lstm_step_tail: gradients into place,
    momentum SGD if `update` = (lr, momentum, {parameter name: buffer}), the next
    step's tables, the loss) instead of three elementwise launches, the
    optimizer's, the loss reduction and two table packs."""
    refbuf, acts, d_gates, d_zout, d_conv, cot_amax = saved[:6]
    B, H = dims
    dev = acts.device
    flat, gr = _flat_grads(dev, {
        "lstm.weight_ih": (32, 175), "lstm.weight_hh": (32, 8),
        "lstm.bias_ih": (32,), "fc_out.weight": (4, 8), "fc_out.bias": (4,),
        "conv_ref.weight": (20, 9, 3), "conv_ref.bias": (20,)})
    ih_hh = torch.empty(32, 183, dtype=torch.float32, device=dev)
    # [dW_ih | dW_hh] = d_gates . [x ; h_prev]^T, db = row sums, dW_out = d_zout .
    # h_new^T: ONE kernel that recomputes the conv part of x (round 6); the conv
    # weights' own gradient: two segmented products over the window planes
    st_all = refbuf[2 * H * 9:]
    if tail is not None and tail[3] is not None:
        pol, tab = None, tail[3].fwd          # resident tables (current: ensure())
    else:
        pw8 = dict(zip(("conv_w", "conv_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_out",
                        "b_out"), saved[6:14]))
        pol = ctypes.byref(_capi.ApgLstmPolicy(**{k: ptr(v) for k, v in pw8.items()}))
        tab = torch.empty(lib().apg_quad_lstm_workspace_floats(), dtype=torch.float32,
                          device=dev)
    scratch = torch.empty(max(1, lib().apg_quad_lstm_gate_wgrad_partials_floats(B)),
                          dtype=torch.float32, device=dev)
    # the conv weights' gradient from the diagonal sums: one kernel over the planes
    # (round 6; rounds 3-5: two segmented planes_gemm products, _conv_diag_problems);
    # both products' partials are added up by one launch (apg_quad_lstm_wgrads)
    conv_pos = torch.empty(20, 3, dtype=torch.float32, device=dev)
    scratch_c = torch.empty(max(1, lib().apg_quad_lstm_conv_wgrad_partials_floats(B)),
                            dtype=torch.float32, device=dev)
    gr["lstm.bias_hh"] = gr["lstm.bias_ih"]
    t = None
    if tail is not None:
        # the step's tail: with B > 0 the threads that hold the gradients' final sums
        # put them in place and apply the update themselves (`finish`), and what is
        # left for apg_quad_lstm_step_tail is the tables and the loss
        partials, loss, pw, tables, update = tail
        G = _capi.ApgLstmPolicyGrads
        names = dict(zip(("conv_w", "conv_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_out",
                          "b_out"), _LSTM_PARAMS))
        t = _capi.ApgLstmStepTail(
            grad=G(**{c: ptr(gr[n]) for c, n in names.items()}),
            ih_hh=ptr(ih_hh), conv_pos=ptr(conv_pos), update=int(update is not None),
            param=G(**{c: ptr(pw[c]) for c in names}),
            tables_fwd=ptr(tables.fwd), tables_bwd=ptr(tables.bwd),
            loss_partials=ptr(partials), n_partials=partials.numel(), loss=ptr(loss),
            applied=int(B > 0))
        if update is not None:
            lr, momentum, bufs = update
            t.lr, t.momentum = float(lr), float(momentum)
            t.mom = G(**{c: ptr(bufs[n]) for c, n in names.items()})
    check(lib().apg_quad_lstm_wgrads(
        ptr(st_all[:12]), ptr(st_all[12:]), ptr(refbuf[:2 * H * 9]), ptr(acts),
        ptr(d_gates), ptr(d_zout), ptr(cot_amax), ptr(d_conv), ptr(st_all), pol, ptr(tab),
        B, H, ptr(scratch), ptr(scratch_c), ptr(ih_hh),
        ptr(gr["lstm.bias_ih"]), ptr(gr["fc_out.weight"]), ptr(gr["fc_out.bias"]),
        ptr(gr["conv_ref.weight"]), ptr(conv_pos), ptr(gr["conv_ref.bias"]),
        ctypes.byref(t) if t is not None and t.applied else None,
        stream_of(acts)), "apg_quad_lstm_wgrads")
    if tail is None:
        gr["conv_ref.weight"][:, :3].sub_(conv_pos[:, :, None])
        # contiguous per-parameter gradients (the fused optimizer path wants them)
        gr["lstm.weight_ih"].copy_(ih_hh[:, :175])
        gr["lstm.weight_hh"].copy_(ih_hh[:, 175:])
        return flat, gr
    check(lib().apg_quad_lstm_step_tail(ctypes.byref(t), stream_of(acts)),
          "apg_quad_lstm_step_tail")
    params8 = [pw[c] for c in names]
    if update is not None and not torch.cuda.is_current_stream_capturing():
        note_in_kernel_update(params8 + [bufs[n] for n in names.values()])
    tables.refreshed(params8)
    return flat, gr


def quad_lstm_rollout_loss(net, state0, in_ref, ref, dt, params, h0, c0,
                           weights=None):
    """Fused LSTM-mode unroll for an `LSTM_NEW(15, 10, 9, 4, conv=1)` policy.
    Returns (loss, states [H,12,B], actions [H,4,B]); `loss.backward()` fills
    the gradients of net.conv_ref / net.lstm / net.fc_out."""
    return _QuadLstmRolloutLoss.apply(
        state0, in_ref, ref, h0, c0, net.conv_ref.weight, net.conv_ref.bias,
        net.lstm.weight_ih, net.lstm.weight_hh, net.lstm.bias_ih,
        net.lstm.bias_hh, net.fc_out.weight, net.fc_out.bias, dt, params,
        weights or quad_loss_weights())


# ------------------------------- fused autoregressive MLP-policy unroll (K8)
class _QuadMlpRolloutLoss(torch.autograd.Function):
    """loss of the autoregressive unroll with the MLP policy inside the
    kernel (apg_quad_mlp_rollout_train_step, matrix cores).  Same contract as
    _QuadLstmRolloutLoss; the network is hutter_model.Net(15, 10, 9, 4,
    conv=1)."""

    @staticmethod
    def forward(ctx, state0, in_ref, ref, w_s, b_s, conv_w, conv_b, w_1, b_1,
                w_2, b_2, w_3, b_3, w_out, b_out, dt, params, weights, index=None):
        H = 10
        if (w_s.shape != (64, 15) or conv_w.shape != (20, 9, 3)
                or w_1.shape != (64, 224) or w_out.shape != (4, 64)):
            raise ValueError("fused path needs Net(15, 10, 9, 4, conv=1)")
        prepared = getattr(ctx, "prepared", None)
        if prepared is not None:     # quad_recurrent_prepare ran ahead
            B = prepared[2].shape[-1]
        else:
            B = state0.shape[0] if index is None else index.numel()
            if in_ref.shape[1] < 2 * H or in_ref.shape[2] != 9 or ref.shape[1] < H:
                raise ValueError("in_ref [B,2H,9] and ref [B,>=H,9|6] with H = 10")
        N = H * B
        if 256 * N * 4 >= 2 ** 32:
            raise ValueError("batch too large for one fused launch "
                             "(B <= 400 000); split it")
        if prepared is not None:
            refbuf, inr, s0, states, rf = prepared
        else:
            _guard_policy_inputs("fused autoregressive unroll", state0=state0,
                                 in_ref=in_ref)
            src = getattr(ctx, "static_src", None) if index is None else None
            hit = _STATIC_PLANES.lookup("recurrent", src) if src else None
            refbuf, inr, s0, states, rf = hit or _ref_and_states(
                _f32c(in_ref), _f32c(state0), B, H, index, also=(ref[:, :H],))
            if src and hit is None:
                _STATIC_PLANES.store("recurrent", src, (refbuf, inr, s0, states, rf))
        dev = s0.device
        names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2",
                 "w_3", "b_3", "w_out", "b_out")
        pw = dict(zip(names, (_f32c(v).contiguous() for v in (
            w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3, b_3, w_out,
            b_out))))
        require_device(s0, inr, rf, *pw.values())
        pol = _capi.ApgMlpPolicy(**{k: ptr(v) for k, v in pw.items()})
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        actions = new(H, 4, B)
        # everything the weight-gradient GEMMs read as B operand:
        # feat (15 planes) | x1 (224) | h1, h2, h3 (192)
        acts = new(431, N)
        feat, x1, h = acts[:15], acts[15:239], acts[239:431]
        relu_mask = torch.empty(5, N, dtype=torch.int32, device=dev)
        st = stream_of(s0)
        # (tests only: tests/plane_path.py hangs the rounds 1-4 sequence - cotangent
        # planes + planes_gemm products - in here as an independent implementation
        # of the same gradients; nothing in the package sets it)
        plane_tail = getattr(ctx, "plane_tail", None)
        if plane_tail is None:
            # round 5: every weight gradient is accumulated inside the reverse
            # sweep (apg_quad_mlp_rollout_train_step) - no cotangent planes, no
            # second pass of products; every gradient is a view of `flat`
            flat, gr = _flat_grads(dev, {
                "states_in.weight": (64, 15), "states_in.bias": (64,),
                "conv_ref.weight": (20, 9, 3), "conv_ref.bias": (20,),
                "fc1.weight": (64, 224), "fc1.bias": (64,), "fc2.weight": (64, 64),
                "fc2.bias": (64,), "fc3.weight": (64, 64), "fc3.bias": (64,),
                "fc_out.weight": (4, 64), "fc_out.bias": (4,)})
            gs = _capi.ApgMlpPolicyGrads(**{
                k: ptr(gr[n]) for k, n in zip(names, _MLP_PARAMS)})
            ws = new(lib().apg_quad_mlp_rollout_step_workspace_floats())
            part = new(max(1, lib().apg_quad_mlp_rollout_step_partials_floats(B)))
            partials = new(max(1, lib().apg_quad_mlp_loss_partials_count(B)))
            loss = new(1)
            g_s0 = new(12, B) if ctx.needs_input_grad[0] else None
            upd = None
            update = getattr(ctx, "update", None)
            if update is not None:
                lr, momentum, bufs = update
                if B == 0 or any(pw[k].data_ptr() != v.data_ptr() for k, v in zip(names, (
                        w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3, b_3,
                        w_out, b_out))):
                    raise ValueError("in-kernel update needs a non-empty batch and "
                                     "contiguous float32 parameters")
                require_device(*bufs.values())
                upd = ctypes.byref(_capi.ApgMlpSgdUpdate(
                    lr=float(lr), momentum=float(momentum),
                    param=_capi.ApgMlpPolicyGrads(**{k: ptr(v) for k, v in pw.items()}),
                    momentum_buf=_capi.ApgMlpPolicyGrads(**{
                        k: ptr(bufs[n]) for k, n in zip(names, _MLP_PARAMS)})))
            check(lib().apg_quad_mlp_rollout_train_step(
                ptr(s0), ptr(inr), ptr(rf), rf.shape[1], float(dt),
                ctypes.byref(params), ctypes.byref(weights), ctypes.byref(pol), B, H,
                ptr(states), ptr(actions), ptr(acts), relu_mask.data_ptr(),
                ptr(partials), ptr(loss), ctypes.byref(gs), ptr(g_s0), ptr(ws),
                ptr(part), upd, st), "apg_quad_mlp_rollout_train_step")
            if update is not None:
                note_in_kernel_update([w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3,
                                       b_3, w_out, b_out] + list(update[2].values()))
            ctx.flat_grads = (flat, gr)
            ctx.save_for_backward(acts)
            ctx.input_grads = (g_s0,)
            ctx.mark_non_differentiable(states, actions)
            ctx.dims = (B, H)
            return loss.reshape(()), states, actions
        return plane_tail(ctx, dict(
            s0=s0, inr=inr, rf=rf, refbuf=refbuf, states=states, actions=actions,
            acts=acts, feat=feat, x1=x1, h=h, relu_mask=relu_mask, pol=pol, params=params,
            weights=weights, dt=dt, B=B, H=H, N=N, st=st, new=new))

    @staticmethod
    def backward(ctx, g, _gs, _ga):
        gr = {k: v * g for k, v in ctx.flat_grads[1].items()}
        g_s0 = None if ctx.input_grads[0] is None else ctx.input_grads[0].t() * g
        return (g_s0, None, None, *[gr[k] for k in _MLP_PARAMS], None, None, None)


_MLP_PARAMS = ("states_in.weight", "states_in.bias", "conv_ref.weight",
               "conv_ref.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias",
               "fc3.weight", "fc3.bias", "fc_out.weight", "fc_out.bias")


def quad_recurrent_forward_inplace_ref(net, state0, in_ref, dt, params, h0=None, c0=None):
    """SURVEY.md §8a A4 `legacy_inplace_ref`, forward only: the recurrent unroll
    of scripts/train_drone.py:134-157 AS SHIPPED - the reference window is a view
    of the batch and the relative-position subtraction (:138-142) writes through
    it, so every step shifts the rows its window holds again - with the policy
    inside the kernel (`Net(15, 10, 9, 4, conv=1)`, or `LSTM_NEW(...)` with its
    hidden / cell state h0, c0 [B, 8]).  state0 [B,12], in_ref [B,>=2H,9] (not
    modified).  Returns (states [B,H,12], actions [B,H,4]); the loss is
    `quad_mpc_loss(states, ref[:, :H], actions)`.  No gradient exists: the
    reference cannot back-propagate through the in-place write, the training
    paths use the copied window."""
    H = 10
    lstm = hasattr(net, "lstm")
    B = state0.shape[0]
    if in_ref.shape[1] < 2 * H or in_ref.shape[2] != 9:
        raise ValueError("in_ref [B,2H,9] with H = 10")
    _guard_policy_inputs("fused recurrent unroll (as shipped)", state0=state0, in_ref=in_ref)
    with torch.no_grad():
        inr, s0 = to_soa_multi([(_f32c(in_ref)[:, :2 * H], None), (_f32c(state0), None)])
        dev = s0.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        N = H * B
        states, actions = new(H, 12, B), new(H, 4, B)
        mask = torch.empty(5, N, dtype=torch.int32, device=dev)
        st = stream_of(s0)
        if lstm:
            if h0 is None or c0 is None:
                raise ValueError("the LSTM unroll needs h0, c0 [B, 8]")
            h0s, c0s = to_soa_multi([(h0, None), (c0, None)])
            pw = [_f32c(v).contiguous() for v in _net_params(net, _LSTM_PARAMS)]
            pol = _capi.ApgLstmPolicy(**dict(zip(
                ("conv_w", "conv_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out"),
                map(ptr, pw))))
            require_device(s0, inr, h0s, c0s, *pw)
            x, gates, hc, hnew = new(15, N), new(32, N), new(16, N), new(8, N)
            ws = new(lib().apg_quad_lstm_workspace_floats())
            check(lib().apg_quad_lstm_rollout_fwd_inplace_ref(
                ptr(s0), ptr(inr), ptr(h0s), ptr(c0s), float(dt), ctypes.byref(params),
                ctypes.byref(pol), B, H, ptr(states), ptr(actions), ptr(x), ptr(gates),
                ptr(hc), ptr(hnew), mask.data_ptr(), ptr(ws), st),
                "apg_quad_lstm_rollout_fwd_inplace_ref")
        else:
            names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3",
                     "b_3", "w_out", "b_out")
            pw = [_f32c(v).contiguous() for v in _net_params(net, _MLP_PARAMS)]
            if pw[0].shape != (64, 15) or pw[4].shape != (64, 224) or pw[10].shape != (4, 64):
                raise ValueError("fused path needs Net(15, 10, 9, 4, conv=1)")
            pol = _capi.ApgMlpPolicy(**dict(zip(names, map(ptr, pw))))
            require_device(s0, inr, *pw)
            feat, x1, h = new(15, N), new(224, N), new(192, N)
            ws = new(lib().apg_quad_mlp_workspace_floats())
            check(lib().apg_quad_mlp_rollout_fwd_inplace_ref(
                ptr(s0), ptr(inr), float(dt), ctypes.byref(params), ctypes.byref(pol), B, H,
                ptr(states), ptr(actions), ptr(feat), ptr(x1), ptr(h), mask.data_ptr(),
                ptr(ws), st), "apg_quad_mlp_rollout_fwd_inplace_ref")
    return states.permute(2, 0, 1).contiguous(), actions.permute(2, 0, 1).contiguous()


def quad_mlp_rollout_loss(net, state0, in_ref, ref, dt, params, weights=None):
    """Fused autoregressive unroll for a `Net(15, 10, 9, 4, conv=1)` policy
    (train_mode "autoregressive", scripts/train_drone.py:113-173).  Returns
    (loss, states [H,12,B], actions [H,4,B]); `loss.backward()` fills the
    gradients of every used parameter of `net`."""
    return _QuadMlpRolloutLoss.apply(
        state0, in_ref, ref, net.states_in.weight, net.states_in.bias,
        net.conv_ref.weight, net.conv_ref.bias, net.fc1.weight, net.fc1.bias,
        net.fc2.weight, net.fc2.bias, net.fc3.weight, net.fc3.bias,
        net.fc_out.weight, net.fc_out.bias, dt, params,
        weights or quad_loss_weights())


# ------------------------------------------ batched closed-loop evaluation (N2)
def _closed_loop_env(learnt):
    """(struct or None, params override) for the closed-loop kernels: `learnt`
    is a LearntDynamics module (the simulator train_dynamics() fits and
    evaluate_model then flies, scripts/train_drone.py:44-45) or None."""
    if learnt is None:
        return None, None
    model = _learnt_model(learnt)
    return ctypes.byref(model), model


def quad_mlp_closed_loop(net, traj, dt, params, max_steps=251, thresh_div=1.0,
                         thresh_stable=1.0, test_time=0, want_trajectory=False,
                         learnt=None):
    """`QuadEvaluator.follow_trajectory("rand")` (scripts/evaluate_drone.py:
    81-194) for a batch of reference trajectories in one launch
    (apg_quad_mlp_closed_loop).  net: hutter_model.Net(15, 10, 9, 4 or 40,
    conv=1) - a concurrent-mode net uses its first action, as the reference
    does.  traj [B, L, 9] = (position, euler, velocity) rows, used as given
    (the reference's Random adds 3 to z: do that before the call).
    Returns dict(div [T,B], steps [B] int32, and with want_trajectory: drone
    [T+1,12,B], actions [T,4,B], start_states [T,12,B]).
    learnt: a LearntDynamics module - the environment steps through it (action
    transform, analytic step on `params`, residual network) instead of
    FlightmareDynamics."""
    _guard_policy_inputs("closed-loop evaluation", traj=traj)
    B, L, _ = traj.shape
    H = 10
    dev = traj.device
    tr = _f32c(traj).permute(1, 2, 0).contiguous()
    names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2",
             "w_3", "b_3", "w_out", "b_out")
    vals = (net.states_in.weight, net.states_in.bias, net.conv_ref.weight,
            net.conv_ref.bias, net.fc1.weight, net.fc1.bias, net.fc2.weight,
            net.fc2.bias, net.fc3.weight, net.fc3.bias, net.fc_out.weight,
            net.fc_out.bias)
    pw = {k: _f32c(v.detach()).contiguous() for k, v in zip(names, vals)}
    if (pw["w_s"].shape != (64, 15) or pw["conv_w"].shape != (20, 9, 3)
            or pw["w_1"].shape != (64, 224) or pw["w_out"].shape[1] != 64
            or pw["w_out"].shape[0] < 4):
        raise ValueError("closed loop needs Net(15, 10, 9, 4*k, conv=1)")
    require_device(tr, *pw.values())
    pol = _capi.ApgMlpPolicy(**{k: ptr(v) for k, v in pw.items()})
    T = min(int(max_steps), L + 1)
    new = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    div = new(T, B)
    steps = torch.zeros(B, dtype=torch.int32, device=dev)
    drone = new(T + 1, 12, B) if want_trajectory else None
    actions = new(T, 4, B) if want_trajectory else None
    start = new(T, 12, B) if want_trajectory else None
    ws = new(lib().apg_quad_mlp_workspace_floats())
    env, _keep = _closed_loop_env(learnt)
    check(lib().apg_quad_mlp_closed_loop_env(
        ptr(tr), L, float(dt), ctypes.byref(params), env, ctypes.byref(pol), B, H,
        int(max_steps), float(thresh_div), float(thresh_stable), int(test_time),
        ptr(div), steps.data_ptr(), ptr(drone), ptr(actions), ptr(start),
        ptr(ws), stream_of(tr)), "apg_quad_mlp_closed_loop_env")
    out = dict(div=div, steps=steps)
    if want_trajectory:
        out.update(drone=drone, actions=actions, start_states=start)
    return out


def quad_lstm_closed_loop(net, traj, dt, params, h0, c0, max_steps=251,
                          thresh_div=1.0, thresh_stable=1.0, test_time=0,
                          want_trajectory=False, learnt=None):
    """quad_mlp_closed_loop for an `LSTM_NEW(15, 10, 9, 4, conv=1)` controller;
    h0 / c0 [B, 8]: the hidden / cell state at the start of every run (the
    reference resets it once per evaluator and carries it through the run)."""
    _guard_policy_inputs("closed-loop evaluation", traj=traj)
    B, L, _ = traj.shape
    H = 10
    dev = traj.device
    tr = _f32c(traj).permute(1, 2, 0).contiguous()
    h0s, c0s = to_soa(h0), to_soa(c0)
    pw = dict(conv_w=net.conv_ref.weight, conv_b=net.conv_ref.bias,
              w_ih=net.lstm.weight_ih, w_hh=net.lstm.weight_hh,
              b_ih=net.lstm.bias_ih, b_hh=net.lstm.bias_hh,
              w_out=net.fc_out.weight, b_out=net.fc_out.bias)
    pw = {k: _f32c(v.detach()).contiguous() for k, v in pw.items()}
    if pw["w_ih"].shape != (32, 175) or pw["conv_w"].shape != (20, 9, 3) \
            or pw["w_out"].shape != (4, 8) or h0s.shape != (8, B):
        raise ValueError("closed loop needs LSTM_NEW(15, 10, 9, 4, conv=1), h0/c0 [B,8]")
    require_device(tr, h0s, c0s, *pw.values())
    pol = _capi.ApgLstmPolicy(**{k: ptr(v) for k, v in pw.items()})
    T = min(int(max_steps), L + 1)
    new = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    div = new(T, B)
    steps = torch.zeros(B, dtype=torch.int32, device=dev)
    drone = new(T + 1, 12, B) if want_trajectory else None
    actions = new(T, 4, B) if want_trajectory else None
    start = new(T, 12, B) if want_trajectory else None
    ws = new(lib().apg_quad_lstm_workspace_floats())
    env, _keep = _closed_loop_env(learnt)
    check(lib().apg_quad_lstm_closed_loop_env(
        ptr(tr), L, ptr(h0s), ptr(c0s), float(dt), ctypes.byref(params), env,
        ctypes.byref(pol), B, H, int(max_steps), float(thresh_div),
        float(thresh_stable), int(test_time), ptr(div), steps.data_ptr(),
        ptr(drone), ptr(actions), ptr(start), ptr(ws), stream_of(tr)),
        "apg_quad_lstm_closed_loop_env")
    out = dict(div=div, steps=steps)
    if want_trajectory:
        out.update(drone=drone, actions=actions, start_states=start)
    return out


def wing_mlp_closed_loop(net, targets, dt, params, mean, std, data_dt=0.05,
                         data_horizon=10, state0=None, max_steps=1000,
                         thresh_div=10.0, thresh_stable=0.8, test_time=0,
                         want_trajectory=False, learnt=None):
    """`FixedWingEvaluator.fly_to_point` (scripts/evaluate_fixed_wing.py:45-131)
    for a batch of target lists in one launch (apg_wing_mlp_closed_loop).
    net: hutter_model.Net(9, 1, 3, 4*k, conv=False) - the first action of the
    plan is flown, as in the reference.  targets [B, n_targets, 3]; mean / std:
    the data set's 12 normalisation values; data_dt / data_horizon: its dt and
    horizon (WingDataset.prepare_data, neural_control/dataset.py:322-350);
    state0 [B, 12] or None for SimpleWingEnv.zero_reset.
    Returns dict(div_linear [T,B], div_pass [T,B], div_fail [T,B] (-1 where
    fly_to_point appends nothing to div_target), steps [B] int32, and with
    want_trajectory: drone [T,16,B] (state after the step + action), seen
    [T,15,B] (state the policy saw + its target)).
    learnt: a LearntFixedWingDynamics module - the environment steps through its
    forward (the physics on the CURRENT values of its parameters, `params` is
    not read, plus the residual network), one device-to-host read of its 50
    physical numbers per call."""
    _guard_policy_inputs("closed-loop evaluation", targets=targets)
    B, n_targets, _ = targets.shape
    dev = targets.device
    tg = _f32c(targets).permute(1, 2, 0).contiguous()
    s0 = None if state0 is None else to_soa(state0)
    names = ("w_s", "b_s", "w_r", "b_r", "w_1", "b_1", "w_2", "b_2", "w_3",
             "b_3", "w_out", "b_out")
    vals = (net.states_in.weight, net.states_in.bias, net.ref_in.weight,
            net.ref_in.bias, net.fc1.weight, net.fc1.bias, net.fc2.weight,
            net.fc2.bias, net.fc3.weight, net.fc3.bias, net.fc_out.weight,
            net.fc_out.bias)
    pw = {k: _f32c(v.detach()).contiguous() for k, v in zip(names, vals)}
    if (pw["w_s"].shape != (64, 9) or pw["w_r"].shape != (64, 3)
            or pw["w_1"].shape != (64, 128) or pw["w_out"].shape[1] != 64
            or pw["w_out"].shape[0] < 4):
        raise ValueError("closed loop needs Net(9, 1, 3, 4*k, conv=False)")
    require_device(tg, *pw.values())
    if s0 is not None:
        require_device(s0)
    pol = _capi.ApgWingPolicy(**{k: ptr(v) for k, v in pw.items()})
    f12 = ctypes.c_float * 12
    mean = f12(*[float(v) for v in mean])
    std = f12(*[float(v) for v in std])
    T = int(max_steps)
    new = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    div_linear, div_pass, div_fail = new(T, B), new(T, B), new(T, B)
    steps = torch.zeros(B, dtype=torch.int32, device=dev)
    drone = new(T, 16, B) if want_trajectory else None
    seen = new(T, 15, B) if want_trajectory else None
    ws = new(lib().apg_wing_policy_workspace_floats())
    I9 = model = None
    if learnt is not None:
        host = torch.cat((learnt._theta().detach().reshape(-1).float(),
                          learnt.I.detach().reshape(-1).float())).cpu().tolist()
        params = _capi.ApgWingParams(*host[:41])
        I9 = (ctypes.c_float * 9)(*host[41:])
        res = [learnt.linear_state_1.weight, learnt.linear_state_1.bias,
               learnt.linear_state_2.weight, learnt.linear_state_2.bias]
        for t in res:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError("LearntFixedWingDynamics tensors must be contiguous "
                                   "fp32 device tensors (module.to('cuda'))")
        if tuple(res[0].shape) != (64, 16) or tuple(res[2].shape) != (12, 64):
            raise ValueError("closed loop expects the 16 -> 64 -> 12 residual network")
        model = _capi.ApgLearntResidual(None, *[t.data_ptr() for t in res])
    check(lib().apg_wing_mlp_closed_loop_env(
        ptr(tg), n_targets, ptr(s0), float(dt), ctypes.byref(params), I9,
        None if model is None else ctypes.byref(model),
        ctypes.byref(pol), mean, std, float(data_dt), int(data_horizon), B, T,
        float(thresh_div), float(thresh_stable), int(test_time),
        ptr(div_linear), ptr(div_pass), ptr(div_fail), steps.data_ptr(),
        ptr(drone), ptr(seen), ptr(ws), stream_of(tg)),
        "apg_wing_mlp_closed_loop_env")
    out = dict(div_linear=div_linear, div_pass=div_pass, div_fail=div_fail,
               steps=steps)
    if want_trajectory:
        out.update(drone=drone, seen=seen)
    return out


# --------------------------- concurrent mode with the policy inside (config 2)
def note_in_kernel_update(tensors):
    """The kernels have written these tensors (parameters, momentum buffers)
    through raw pointers: bump their in-place version counters, as any in-place
    op would - autograd's saved-tensor checks and everything that caches what it
    derived from the parameters (the step plans' resident operand tables)
    compare them.  Not under stream capture (the bump would happen once, at
    capture; who replays a graph that updates parameters bumps after the
    replay: train_base)."""
    torch.autograd.graph.increment_version(list(tensors))


def _step_events(events):
    """{"inputs_ready" | "after_forward" | "after_reverse": torch.cuda.Event}
    -> ApgStepEvents* (or None)."""
    if not events:
        return None
    return ctypes.byref(_capi.ApgStepEvents(**{
        k: getattr(e, "cuda_event", None) for k, e in events.items() if e is not None}))


class _QuadConcurrentPolicyLoss(torch.autograd.Function):
    """loss of the concurrent training step with `Net(15, 10, 9, 40, conv=1)`
    inside the kernels (apg_quad_mlp_concurrent_train_step): policy once per
    trajectory, rollout + quad_mpc_loss + adjoint, policy reverse pass with every
    weight gradient inside; backward() scales them by the upstream cotangent."""

    @staticmethod
    def forward(ctx, normed, state0, in_ref, ref, w_s, b_s, conv_w, conv_b, w_1,
                b_1, w_2, b_2, w_3, b_3, w_out, b_out, dt, params, weights,
                index=None):
        H = 10
        if (w_s.shape != (64, 15) or conv_w.shape != (20, 9, 3)
                or w_1.shape != (64, 224) or w_out.shape != (40, 64)):
            raise ValueError("fused path needs Net(15, 10, 9, 40, conv=1)")
        prepared = getattr(ctx, "prepared", None)
        # one buffer for the B operands of the weight products:
        # feat (15) | x1 (224) | h1, h2, h3 (192) | in_ref rows (H*9)
        if prepared is not None:     # quad_concurrent_prepare ran ahead
            acts, s0, rf = prepared
            B = s0.shape[-1]
        else:
            B = state0.shape[0] if index is None else index.numel()
            src = getattr(ctx, "static_src", None) if index is None else None
            hit = _STATIC_PLANES.lookup("concurrent", src) if src else None
            # (a hit: feat / in_ref planes still valid, the kernels rewrite x1
            # and h every step)
            acts, s0, rf = hit or quad_concurrent_prepare(
                normed, state0, in_ref, ref, index)
            if src and hit is None:
                _STATIC_PLANES.store("concurrent", src, (acts, s0, rf))
        dev = s0.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        feat, x1, h, inr = acts[:15], acts[15:239], acts[239:431], acts[431:]
        names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2",
                 "w_3", "b_3", "w_out", "b_out")
        pw = dict(zip(names, (_f32c(v).contiguous() for v in (
            w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3, b_3, w_out,
            b_out))))
        require_device(acts, s0, rf, *pw.values())
        pol = _capi.ApgMlpPolicy(**{k: ptr(v) for k, v in pw.items()})
        relu_mask = torch.empty(5, B, dtype=torch.int32, device=dev)
        partials = new(max(1, lib().apg_quad_mlp_loss_partials_count(B)))
        loss = new(1)
        ctx.dims = (B, H)
        plane_tail = getattr(ctx, "plane_tail", None)    # (tests only, as above)
        if plane_tail is None:
            # round 4: the weight gradients are accumulated inside the reverse
            # pass (apg_quad_mlp_concurrent_step) - no cotangent planes, no
            # second pass of products; every gradient is a view of `flat`
            flat, gr = _flat_grads(dev, {
                "states_in.weight": (64, 15), "states_in.bias": (64,),
                "conv_ref.weight": (20, 9, 3), "conv_ref.bias": (20,),
                "fc1.weight": (64, 224), "fc1.bias": (64,), "fc2.weight": (64, 64),
                "fc2.bias": (64,), "fc3.weight": (64, 64), "fc3.bias": (64,),
                "fc_out.weight": (40, 64), "fc_out.bias": (40,)})
            gs = _capi.ApgMlpPolicyGrads(**{
                k: ptr(gr[n]) for k, n in zip(names, _MLP_PARAMS)})
            d_zout = new(40, B)
            ws = new(lib().apg_quad_mlp_step_workspace_floats())
            part = new(max(1, lib().apg_quad_mlp_step_partials_floats(B)))
            upd = None
            update = getattr(ctx, "update", None)
            if update is not None:
                # the optimizer inside the second stage: momentum SGD on the
                # very tensors the policy struct points at
                lr, momentum, bufs = update
                if B == 0 or any(pw[k].data_ptr() != v.data_ptr() for k, v in zip(names, (
                        w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3, b_3,
                        w_out, b_out))):
                    raise ValueError("in-kernel update needs a non-empty batch and "
                                     "contiguous float32 parameters")
                require_device(*bufs.values())
                upd = ctypes.byref(_capi.ApgMlpSgdUpdate(
                    lr=float(lr), momentum=float(momentum),
                    param=_capi.ApgMlpPolicyGrads(**{k: ptr(v) for k, v in pw.items()}),
                    momentum_buf=_capi.ApgMlpPolicyGrads(**{
                        k: ptr(bufs[n]) for k, n in zip(names, _MLP_PARAMS)})))
            check(lib().apg_quad_mlp_concurrent_train_step(
                ptr(s0), ptr(rf), rf.shape[1], float(dt), ctypes.byref(params),
                ctypes.byref(weights), ctypes.byref(pol), B, H, ptr(acts),
                relu_mask.data_ptr(), ptr(d_zout), ptr(partials), ptr(loss),
                ctypes.byref(gs), None, ptr(ws), ptr(part), upd,
                _step_events(getattr(ctx, "events", None)),
                stream_of(s0)), "apg_quad_mlp_concurrent_train_step")
            if update is not None:
                note_in_kernel_update([w_s, b_s, conv_w, conv_b, w_1, b_1, w_2, b_2, w_3,
                                       b_3, w_out, b_out] + list(update[2].values()))
            ctx.flat_grads = (flat, gr)
            ctx.save_for_backward(acts)
            return loss.reshape(())
        return plane_tail(ctx, dict(
            acts=acts, feat=feat, x1=x1, h=h, inr=inr, s0=s0, rf=rf, relu_mask=relu_mask,
            pol=pol, params=params, weights=weights, dt=dt, B=B, H=H, partials=partials,
            loss=loss, new=new))

    @staticmethod
    def backward(ctx, g):
        gr = {k: v * g for k, v in ctx.flat_grads[1].items()}
        return (None, None, None, None, *[gr[k] for k in _MLP_PARAMS], None, None, None)


class QuadConcurrentStepPlan:
    """The concurrent training step of `Net(15, 10, 9, 40, conv=1)` with
    everything around the launch made ONCE: scratch planes, the flat gradient
    buffer, partials, the argument structs.  `launch()` is then one call into
    apg_quad_mlp_concurrent_train_step (five kernel launches) and nothing else:
    ~20 us of host time per step, so a Python loop keeps the GPU busy without
    a captured graph - and without the ~6-14 us a graph replay costs between
    two steps on this platform (tools/ab_graph_alternation.py).
      prepared  (acts [521][B], state0 planes, ref planes): what
                quad_concurrent_prepare returned - refilled in place by the
                caller between launches when the batch changes;
      update    (lr, momentum, {name: momentum buffer}) or None, as for
                quad_concurrent_policy_grads.
    Valid while the network's parameter tensors, `prepared`'s buffers, the
    physics struct, dt and the optimizer settings are the ones given here
    (TrainBase rebuilds it when its step signature changes).  The gradients are
    views of `flat` (+ one slot for the loss), `named` by parameter name."""

    def __init__(self, net, prepared, dt, params, weights=None, update=None, rows=None):
        """rows = (normed [N,15], state0 [N,12], in_ref [N,>=10,9], ref [N,>=10,9|6],
        B): the DATA SET's tensors - `launch(index=...)` then names the batch by
        row numbers and the forward kernel reads the rows itself
        (apg_quad_mlp_concurrent_train_step_rows: no gather pass, `prepared` is
        None)."""
        if rows is not None:
            normed, st0, inr, rfs, B = rows
            for t in (normed, st0, inr, rfs):
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise ValueError("rows: contiguous float32 device tensors")
            N = st0.shape[0]
            if (normed.shape != (N, 15) or st0.shape != (N, 12) or inr.shape[0] != N
                    or inr.shape[1] < 10 or inr.shape[2] != 9 or rfs.shape[0] != N
                    or rfs.shape[1] < 10 or rfs.shape[2] not in (9, 6)):
                raise ValueError("rows: normed [N,15], state0 [N,12], in_ref [N,>=10,9], "
                                 "ref [N,>=10,9|6]")
            acts = torch.empty(431 + 90, B, dtype=torch.float32, device=st0.device)
            s0 = rf = None
            prepared = (acts, None, None)
            H = 10
        else:
            acts, s0, rf = prepared
            B, H = s0.shape[-1], 10
        tensors = _net_params(net, _MLP_PARAMS)
        shapes = {"states_in.weight": (64, 15), "conv_ref.weight": (20, 9, 3),
                  "fc1.weight": (64, 224), "fc_out.weight": (40, 64)}
        for n, t in zip(_MLP_PARAMS, tensors):
            if n in shapes and tuple(t.shape) != shapes[n]:
                raise ValueError("fused path needs Net(15, 10, 9, 40, conv=1)")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("the step plan needs contiguous float32 parameters")
        if B == 0:
            raise ValueError("the step plan needs a non-empty batch")
        require_device(acts, *tensors) if rows is not None else \
            require_device(acts, s0, rf, *tensors)
        dev = acts.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2",
                 "w_3", "b_3", "w_out", "b_out")
        self.flat, self.named = _flat_grads(dev, {
            n: tuple(t.shape) for n, t in zip(_MLP_PARAMS, tensors)})
        self.loss = new(1)
        self.loss0 = self.loss.reshape(())
        # everything the launch reads or writes stays referenced from here
        self._keep = dict(
            prepared=prepared, tensors=[t.detach() for t in tensors],
            relu_mask=torch.empty(5, B, dtype=torch.int32, device=dev),
            partials=new(max(1, lib().apg_quad_mlp_loss_partials_count(B))),
            d_zout=new(40, B), ws=new(lib().apg_quad_mlp_step_workspace_floats()),
            part=new(max(1, lib().apg_quad_mlp_step_partials_floats(B))),
            params=params, weights=weights or quad_loss_weights())
        k = self._keep
        k["pol"] = _capi.ApgMlpPolicy(**{a: ptr(t) for a, t in zip(names, k["tensors"])})
        k["gs"] = _capi.ApgMlpPolicyGrads(**{
            a: ptr(self.named[n]) for a, n in zip(names, _MLP_PARAMS)})
        upd = None
        if update is not None:
            lr, momentum, bufs = update
            require_device(*bufs.values())
            k["bufs"] = bufs
            self._written = list(k["tensors"]) + list(bufs.values())
            k["upd"] = _capi.ApgMlpSgdUpdate(
                lr=float(lr), momentum=float(momentum),
                param=_capi.ApgMlpPolicyGrads(**{a: ptr(t) for a, t in
                                                 zip(names, k["tensors"])}),
                momentum_buf=_capi.ApgMlpPolicyGrads(**{
                    a: ptr(bufs[n]) for a, n in zip(names, _MLP_PARAMS)}))
            upd = ctypes.byref(k["upd"])
        self.updates = update is not None
        # resident operand tables (ApgMlpSgdUpdate.resident): the plan owns the
        # workspace, so the second stage can keep the packed tables current and
        # the pack launch at the head of the step goes away - as long as nobody
        # else writes the parameters (their in-place version counters, which the
        # kernel's update does not touch, are compared before every launch)
        self.resident_tables = RESIDENT_TABLES and update is not None
        self._versions, self._map_built = None, False
        self._dev = dev
        self.B = B
        tail = (float(dt), ctypes.byref(params),
                ctypes.byref(k["weights"]), ctypes.byref(k["pol"]), B, H, ptr(acts),
                k["relu_mask"].data_ptr(), ptr(k["d_zout"]), ptr(k["partials"]),
                ptr(self.loss), ctypes.byref(k["gs"]), None, ptr(k["ws"]),
                ptr(k["part"]), upd)
        # rows plans: every launch adds its loss to `running` (an epoch loop
        # zeroes it, reads it once at the end)
        self.running = torch.zeros(1, dtype=torch.float32, device=dev)
        self.launches = 0
        if rows is not None:
            k["rows_src"] = (normed, st0, inr, rfs)
            self._rows = k["rows"] = _capi.ApgBatchRows(
                index=None, normed=ptr(normed), state0=ptr(st0), in_ref=ptr(inr),
                ref=ptr(rfs), ld_normed=15, ld_state0=12,
                ld_in_ref=inr.shape[1] * 9, ld_ref=rfs.shape[1] * rfs.shape[2], n_rows=N,
                running_loss=ptr(self.running))
            self._fn = lib().apg_quad_mlp_concurrent_train_step_rows
            self._args = [ctypes.byref(self._rows), rfs.shape[2], *tail]
        else:
            self._rows = None
            self._fn = lib().apg_quad_mlp_concurrent_train_step
            self._args = [ptr(s0), ptr(rf), rf.shape[1], *tail]
        # every launch writes its loss into a tensor of its own (handed to the
        # caller as it is: the clone a shared buffer needed was a 5 us copy
        # kernel per step, 4 % of it)
        self._loss_at = len(self._args) - len(tail) + 10

    def launch(self, events=None, index=None):
        """Enqueue the step on the current stream; returns the loss (0-dim, a
        tensor of this launch's own).  index (rows
        plans): int64 [B] device tensor, contiguous - this batch's rows; the
        caller keeps it alive until the step has run."""
        if self._rows is not None:
            if (index is None or index.dtype != torch.int64 or not index.is_cuda
                    or index.numel() != self.B or not index.is_contiguous()):
                raise ValueError("rows plan: index must be a contiguous int64 device "
                                 f"tensor of {self.B} row numbers")
            self._rows.index = index.data_ptr()
        if self.resident_tables:
            k = self._keep
            if torch.cuda.is_current_stream_capturing():
                # (a replayed graph runs no Python: it could not notice a parameter
                # written from outside - captured steps pack their tables)
                k["upd"].resident, self._versions = 0, None
            else:
                # (version counter AND storage address: `p.data = other` moves the
                # storage without touching the counter of the Parameter)
                now = [(t._version, t.data_ptr()) for t in k["tensors"]]
                # 1: first launch on this workspace (builds the table map); then
                # 2 while nobody else has written the parameters, else 3 (pack)
                k["upd"].resident = (2 if now == self._versions
                                     else 3 if self._map_built else 1)
                self._versions, self._map_built = now, True
        self.loss = torch.empty(1, dtype=torch.float32, device=self._dev)
        self.loss0 = self.loss.reshape(())
        self._args[self._loss_at] = ptr(self.loss)
        check(self._fn(*self._args, _step_events(events),
                       torch.cuda.current_stream(self._dev).cuda_stream),
              "apg_quad_mlp_concurrent_train_step")
        self.launches += 1
        if self.updates:
            note_in_kernel_update(self._written)
            if self._versions is not None:      # (ours: the tables follow them)
                self._versions = [(t._version, t.data_ptr()) for t in self._keep["tensors"]]
        return self.loss0

    def invalidate(self):
        """The parameters were written behind autograd's back - through `p.data`
        (`p.data.clamp_()`, `p.data.copy_()`: weight clipping, soft updates), which
        bumps a version counter this plan cannot see.  The next launch
        packs the operand tables from the parameters again (one pack launch, ~6 us).
        TrainBase calls this at the start of every epoch; a caller that writes
        `.data` between steps calls it after each write, or builds its trainer with
        `resident_tables = False` (pack at every step)."""
        self._versions = None


# True: a step plan with the in-kernel update keeps its packed operand tables
# current from step to step (no pack launch)
RESIDENT_TABLES = True

def quad_concurrent_policy_loss(net, normed, state0, in_ref, ref, dt, params,
                                weights=None):
    """The concurrent training step's loss for a `Net(15, 10, 9, 40, conv=1)`:
    quad_mpc_loss(unroll(dyn, state0, sigmoid(net(normed, in_ref))), ref) with
    the policy inside the kernels; `loss.backward()` fills the parameter
    gradients (the inputs carry none, as in scripts/train_base.py:198-204)."""
    return _QuadConcurrentPolicyLoss.apply(
        normed, state0, in_ref, ref, net.states_in.weight, net.states_in.bias,
        net.conv_ref.weight, net.conv_ref.bias, net.fc1.weight, net.fc1.bias,
        net.fc2.weight, net.fc2.bias, net.fc3.weight, net.fc3.bias,
        net.fc_out.weight, net.fc_out.bias, dt, params,
        weights or quad_loss_weights())


# ------------------------------------- fused policies without the autograd tape
class _DirectCtx:
    """Stand-in for the autograd context when a fused loss is evaluated for its
    parameter gradients only (trainers): no input gradients, nothing taped."""
    needs_input_grad = (False,) * 8

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass


_MLP_PARAM_SLOTS = None


def mlp_param_objects(net):
    """The 12 Parameter objects of `_MLP_PARAMS` as they are installed in `net`
    right now - through the modules' own dictionaries (12 attribute reads on
    nn.Module cost ~10 us, more than everything else a planned step does on
    the host); () when the network has other submodules."""
    global _MLP_PARAM_SLOTS
    if _MLP_PARAM_SLOTS is None:
        _MLP_PARAM_SLOTS = [tuple(n.rsplit(".", 1)) for n in _MLP_PARAMS]
    try:
        mods = net._modules
        return [mods[m]._parameters[a] for m, a in _MLP_PARAM_SLOTS]
    except (KeyError, AttributeError):
        return ()


def _net_params(net, names):
    out = []
    for n in names:
        mod, attr = n.rsplit(".", 1)
        out.append(getattr(net.get_submodule(mod), attr).detach())
    return out


def quad_concurrent_policy_grads(net, normed, state0, in_ref, ref, dt, params,
                                 weights=None, index=None, static_inputs=False,
                                 prepared=None, events=None, update=None):
    """quad_concurrent_policy_loss + its parameter gradients, without autograd:
    returns (loss, {parameter name: gradient}, flat); the gradients are
    contiguous views of the flat buffer (no per-parameter clone as
    `loss.backward()` does), whose last element is a free slot for the loss -
    so the buffer itself is the all-reduce message.  `index` (int64 device
    tensor): the four tensors are the whole data set and the minibatch
    rows are gathered while they are brought into the plane layout.
    `static_inputs` (index None): the caller steps repeatedly on these very
    tensor objects (its resident shard) - their plane-layout copies are kept
    while the tensors stay unchanged (_StaticPlanes).  `prepared`: the result
    of quad_concurrent_prepare for this batch (the four tensors are then
    unused and may be None).  `events`: torch.cuda.Events of a pipelined caller,
    {"inputs_ready": waited on before the forward kernel, "after_forward" /
    "after_reverse": recorded once that kernel is enqueued} (include/apg.h,
    ApgStepEvents; in-sweep path only; run_epoch issues the next batch's gather
    behind one of them).
    `update` = (lr, momentum, {parameter name: momentum buffer}): the step also
    APPLIES torch.optim.SGD's update (buf = momentum buf + grad, p -= lr buf)
    to the network's parameters and these buffers, inside the second stage
    (apg_quad_mlp_concurrent_train_step; in-sweep path, one process)."""
    ctx = _DirectCtx()
    ctx.events = events
    if update is not None:
        ctx.update = update
    if prepared is not None:
        ctx.prepared = prepared
    elif static_inputs and index is None:
        ctx.static_src = (normed, state0, in_ref, ref)
    with torch.no_grad():
        loss = _QuadConcurrentPolicyLoss.forward(
            ctx, normed, state0, in_ref, ref, *_net_params(net, _MLP_PARAMS), dt,
            params, weights or quad_loss_weights(), index)
        flat, gr = ctx.flat_grads
    return loss, gr, flat


# the sweeps address every plane tensor with unsigned 32-bit byte offsets: the
# largest (256 cotangent planes of H*B floats) stays below 4 GiB up to 419 430
_MAX_FUSED_AR_BATCH = 393216


def quad_mlp_rollout_grads(net, state0, in_ref, ref, dt, params, weights=None,
                           index=None, static_inputs=False, prepared=None,
                           update=None):
    """quad_mlp_rollout_loss (autoregressive unroll) + parameter gradients,
    without autograd; see quad_concurrent_policy_grads (`update` as there:
    momentum SGD inside the second stage, one process, one launch chunk).
    Batches beyond 393 216 trajectories are processed in chunks (losses and
    gradients are sums over trajectories, so the chunks simply add up)."""
    B = (prepared[2].shape[-1] if prepared is not None
         else state0.shape[0] if index is None else index.numel())
    if B > _MAX_FUSED_AR_BATCH:
        if prepared is not None or update is not None:
            raise ValueError("prepared batches / in-kernel updates must fit one "
                             "fused launch")
        n = -(-B // _MAX_FUSED_AR_BATCH)
        step = -(-B // n)
        if index is None:
            index = torch.arange(B, device=state0.device)
        loss, gr, flat = None, None, None
        for lo in range(0, B, step):
            l, g, f = quad_mlp_rollout_grads(net, state0, in_ref, ref, dt, params,
                                             weights, index[lo:lo + step].contiguous())
            if flat is None:
                loss, gr, flat = l, g, f
            else:
                flat += f
                loss = loss + l
        return loss, gr, flat
    ctx = _DirectCtx()
    if update is not None:
        ctx.update = update
    if prepared is not None:     # quad_recurrent_prepare's result
        ctx.prepared = prepared
    elif static_inputs and index is None:
        ctx.static_src = (state0, in_ref, ref)
    with torch.no_grad():
        loss, _, _ = _QuadMlpRolloutLoss.forward(
            ctx, state0, in_ref, ref, *_net_params(net, _MLP_PARAMS), dt, params,
            weights or quad_loss_weights(), index)
        flat, gr = ctx.flat_grads
    return loss, gr, flat


_LSTM_PARAMS = ("conv_ref.weight", "conv_ref.bias", "lstm.weight_ih",
                "lstm.weight_hh", "lstm.bias_ih", "lstm.bias_hh", "fc_out.weight",
                "fc_out.bias")


def quad_lstm_rollout_grads(net, state0, in_ref, ref, dt, params, h0, c0,
                            weights=None, index=None, static_inputs=False,
                            prepared=None, update=None, resident_tables=True,
                            rows_in_kernel=True):
    """quad_lstm_rollout_loss + parameter gradients, without autograd.
    resident_tables (round 6): the sweeps read operand tables that stay packed
    between steps (LstmResidentTables) and ONE launch behind the products puts
    the gradients into place, reduces the loss and packs the next step's tables;
    `update` = (lr, momentum, {parameter name: momentum buffer}): that launch
    also applies torch.optim.SGD's momentum update (one process: the trainer
    then skips optimizer.step()).  rows_in_kernel (with `index` and resident
    tables): the sweeps read the data set's rows through the index themselves;
    False: the gather pass (the same numbers - what the tests compare with)."""
    ctx = _DirectCtx()
    ctx.rows_in_kernel = rows_in_kernel
    if prepared is not None:     # quad_recurrent_prepare's result
        ctx.prepared = prepared
    elif static_inputs and index is None:
        ctx.static_src = (state0, in_ref, ref)
    if update is not None and not resident_tables:
        raise ValueError("the in-kernel update needs the resident tables")
    with torch.no_grad():
        if resident_tables:
            ctx.lstm_tables = lstm_resident_tables(net, net.fc_out.weight.device)
        loss, _, _ = _QuadLstmRolloutLoss.forward(
            ctx, state0, in_ref, ref, h0, c0, *_net_params(net, _LSTM_PARAMS), dt,
            params, weights or quad_loss_weights(), index)
        tail = None
        if resident_tables:
            partials, loss1, pw = ctx.lstm_tail
            tail = (partials, loss1, pw, ctx.lstm_tables, update)
        flat, gr = _lstm_param_grads(ctx.saved_tensors, ctx.dims, tail)
    return loss, gr, flat


# ------------------------------ fixed wing: concurrent step, policy in-kernel
_WING_PARAMS = ("states_in.weight", "states_in.bias", "ref_in.weight", "ref_in.bias",
                "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight",
                "fc3.bias", "fc_out.weight", "fc_out.bias")


def wing_concurrent_policy_grads(net, normed, in_ref, state0, ref, dt, params,
                                 weights=None, index=None):
    """The fixed-wing concurrent training step without the autograd tape:
    loss = fixed_wing_mpc_loss(unroll(dyn, state0, sigmoid(net(normed,
    in_ref))), ref) for `Net(9, 1, 3, 4 H, conv=False)`, H = 10 or 20, with the policy on the
    matrix cores (apg_wing_policy_fwd / _bwd) around the fused rollout
    (apg_wing_rollout_fwd_bwd).  normed [B,9], in_ref [B,3], state0 [B,12],
    ref [B,H,3].  Returns (loss, {parameter name: gradient}, flat) like
    quad_concurrent_policy_grads; `index` selects the batch rows out of whole
    data-set tensors in the same way."""
    B, H = (state0.shape[0] if index is None else index.numel()), ref.shape[1]
    NA = 4 * H
    pw = dict(zip(("w_s", "b_s", "w_r", "b_r", "w_1", "b_1", "w_2", "b_2", "w_3",
                   "b_3", "w_out", "b_out"),
                  (_f32c(v).contiguous() for v in _net_params(net, _WING_PARAMS))))
    if (pw["w_s"].shape != (64, 9) or pw["w_r"].shape != (64, 3)
            or pw["w_1"].shape != (64, 128) or pw["w_out"].shape != (NA, 64)
            or H not in (10, 20)
            or normed.shape[1] != 9 or in_ref.reshape(in_ref.shape[0], -1).shape[1] != 3
            or ref.shape[1:] != (H, 3)):
        raise ValueError("fused path needs Net(9, 1, 3, 4 H, conv=False), H = 10 or 20")
    _guard_policy_inputs("fused fixed-wing step", normed=normed, in_ref=in_ref)
    dev = state0.device
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    with torch.no_grad():
        # B operands of the weight products: feat 0..8 | ref 9..11 | x1 12..139 |
        # h1 140.. | h2 204.. | h3 268..
        acts = new(332, B)
        feat, rin, x1, h = acts[:9], acts[9:12], acts[12:140], acts[140:]
        _, _, s0_planes, ref_planes = to_soa_multi(
            [(normed, feat), (in_ref.reshape(-1, 3), rin), (state0, None),
             (ref, None)], index=index)
        require_device(acts, *pw.values())
        pol = _capi.ApgWingPolicy(**{k: ptr(v) for k, v in pw.items()})
        actions = new(H, 4, B)
        ws = new(lib().apg_wing_policy_workspace_floats())
        st = stream_of(acts)
        check(lib().apg_wing_policy_fwd(ptr(feat), ptr(rin), ctypes.byref(pol), B, H,
                                        ptr(actions), ptr(x1), ptr(h), ptr(ws), st),
              "apg_wing_policy_fwd")
        res = wing_rollout_fwd_bwd(s0_planes, actions, ref_planes, dt, params,
                                   weights, layout="soa", want_grad_state0=False)
        cot = new(NA + 320, B)
        d_zout, d_pre = cot[:NA], cot[NA:]
        check(lib().apg_wing_policy_bwd(
            ptr(actions), ptr(res["grad_actions"]), ptr(x1), ptr(h),
            ctypes.byref(pol), B, H, ptr(d_zout), ptr(d_pre), ptr(ws), st),
            "apg_wing_policy_bwd")
        flat, gr = _flat_grads(dev, {
            "states_in.weight": (64, 9), "states_in.bias": (64,),
            "ref_in.weight": (64, 3), "ref_in.bias": (64,),
            "fc1.weight": (64, 128), "fc1.bias": (64,), "fc2.weight": (64, 64),
            "fc2.bias": (64,), "fc3.weight": (64, 64), "fc3.bias": (64,),
            "fc_out.weight": (NA, 64), "fc_out.bias": (NA,)})
        R = lambda lo, hi_: make_bdesc(dev, range(lo, hi_), key=("wing", lo, hi_))
        head = [dict(A=d_zout[:min(NA, 64)], M=min(NA, 64), S=1, Bp=acts,
                     bdesc=R(268, 332), out=gr["fc_out.weight"][:64],
                     bias_out=gr["fc_out.bias"][:64])]
        if NA > 64:       # M <= 64 per product: the 80-row head in two parts
            head.append(dict(A=d_zout[64:], M=NA - 64, S=1, Bp=acts, bdesc=R(268, 332),
                             out=gr["fc_out.weight"][64:], bias_out=gr["fc_out.bias"][64:]))
        _run_products(head + [
            dict(A=d_pre[128:192], M=64, S=1, Bp=acts, bdesc=R(204, 268),
                 out=gr["fc3.weight"], bias_out=gr["fc3.bias"]),
            dict(A=d_pre[64:128], M=64, S=1, Bp=acts, bdesc=R(140, 204),
                 out=gr["fc2.weight"], bias_out=gr["fc2.bias"]),
            dict(A=d_pre[0:64], M=64, S=1, Bp=acts, bdesc=R(12, 76), with_ones=False,
                 out=gr["fc1.weight"]),
            dict(A=d_pre[0:64], M=64, S=1, Bp=acts, bdesc=R(76, 140),
                 out=gr["fc1.weight"][:, 64:], bias_out=gr["fc1.bias"]),
            dict(A=d_pre[192:256], M=64, S=1, Bp=acts, bdesc=R(0, 9),
                 out=gr["states_in.weight"], bias_out=gr["states_in.bias"]),
            dict(A=d_pre[256:320], M=64, S=1, Bp=acts, bdesc=R(9, 12),
                 out=gr["ref_in.weight"], bias_out=gr["ref_in.bias"])])
    return res["loss"].reshape(()), gr, flat
