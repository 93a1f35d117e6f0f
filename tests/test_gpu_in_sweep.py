"""Round 4: the concurrent training step with its weight gradients accumulated
INSIDE the reverse kernel (apg_quad_mlp_concurrent_step; csrc/mlp_concurrent.hip,
mlp_concurrent_bwd_tm_kernel) - no cotangent planes, no second pass of
products.  One loss.backward() of the reference yields every parameter
gradient (scripts/train_drone.py:175-203); this path must too, to the same
1e-4 as the plane + product path it replaces (which stays available behind
tests/plane_path.py since round 6 - the package has one path - and is the
comparison here)."""
import copy
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
H, DT = 10, 0.1


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


import plane_path as PP


def N(t):
    return t.detach().double().cpu().numpy()


def _case(B, seed, dev):
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=seed)
    s0 = d["state0"].to(dev)
    with torch.no_grad():
        normed = state_preprocessing(s0)
    return d, (normed, s0, d["in_ref"].to(dev), d["ref"].to(dev))


def _fp64_grads(net, d):
    from oracle import torch_port as tp
    net64 = copy.deepcopy(net).double().cpu()
    s64 = d["state0"].double()
    acts = torch.sigmoid(net64(tp.quad_state_features(s64),
                               d["in_ref"].double())).reshape(-1, H, 4)
    loss = tp.quad_mpc_loss(tp.unroll(tp.QuadOracle(dtype=torch.float64), s64, acts, DT),
                            d["ref"].double(), acts)
    loss.backward()
    return loss.item(), {k: p.grad.numpy() for k, p in net64.named_parameters()
                         if p.grad is not None}


# 1: one lane; 31 / 77: part of a wave; 256: exactly one workgroup; 257: a
# second workgroup with one trajectory; 300, 4113: ragged last workgroups;
# 8192 + 3: more than one chunk of the second stage (32 workgroups)
@pytest.mark.parametrize("B", [1, 31, 77, 256, 257, 300, 4113, 8195])
def test_in_sweep_gradients_vs_fp64_oracle_and_plane_products(dev, B):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    torch.manual_seed(8)
    net = Net(15, H, 9, 4 * H, conv=1)
    gnet = copy.deepcopy(net).to(dev)
    d, inputs = _case(B, 100 + B, dev)
    dyn = FlightmareDynamics()
    res = []
    for fn in (F.quad_concurrent_policy_grads, PP.quad_concurrent_policy_grads_planes):
        loss, grads, flat = fn(gnet, *inputs, DT, dyn.params)
        assert flat.numel() == sum(g.numel() for g in grads.values()) + 1   # + loss slot
        res.append((loss.item(), {k: N(v) for k, v in grads.items()}))
    loss64, want = _fp64_grads(net, d)
    (l1, g1), (l0, g0) = res
    assert l1 == l0                               # the same forward kernel
    assert abs(l1 - loss64) / abs(loss64) < 1e-5
    assert set(g1) == set(want)
    for k, w in want.items():
        assert rel_err(g1[k], w) < 1e-4, (k, rel_err(g1[k], w))
        assert rel_err(g1[k], g0[k]) < 2e-5, (k, rel_err(g1[k], g0[k]))


def test_in_sweep_is_deterministic_and_feeds_autograd(dev):
    """Fixed-order sums (staged kernel) / fixed-point accumulators (trajectory-
    major kernel) and a fixed-order second stage: equal inputs give equal bits,
    at a batch of several workgroups and chunks as well; loss.backward()
    of the autograd entry point delivers the same gradients (scaled by the
    upstream cotangent)."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    torch.manual_seed(3)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    _, inputs = _case(9000, 5, dev)
    dyn = FlightmareDynamics()
    l0, g0, f0 = F.quad_concurrent_policy_grads(net, *inputs, DT, dyn.params)
    f0 = f0.clone()
    for _ in range(3):
        l1, g1, f1 = F.quad_concurrent_policy_grads(net, *inputs, DT, dyn.params)
        assert torch.equal(f1[:-1], f0[:-1]) and torch.equal(l1, l0)
    loss = F.quad_concurrent_policy_loss(net, *inputs, DT, dyn.params)
    (2.5 * loss).backward()
    for k, p in net.named_parameters():
        if k in g0:       # (Net's linear reference branch is unused with conv=1)
            assert torch.allclose(p.grad, 2.5 * g0[k], rtol=1e-6, atol=0), k
        else:
            assert p.grad is None, k


def test_concurrent_step_through_the_c_abi(dev):
    """apg_quad_mlp_concurrent_step called directly (plain pointers): the
    result of the Python entry point; B = 0 zeroes the gradients and the loss;
    argument errors come back as APG_ERR_ARG."""
    from apg_trajectory_tracking_amd import _capi, functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    lib = _capi.lib()
    B = 700
    torch.manual_seed(11)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    _, inputs = _case(B, 17, dev)
    dyn = FlightmareDynamics()
    want_loss, want, _ = F.quad_concurrent_policy_grads(net, *inputs, DT, dyn.params)
    acts, s0, rf = F.quad_concurrent_prepare(*inputs)
    new = lambda *s: torch.full(s, 7.0, device=dev)
    names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3",
             "w_out", "b_out")
    params = [p.detach().contiguous() for p in F._net_params(net, F._MLP_PARAMS)]
    pol = _capi.ApgMlpPolicy(**{k: v.data_ptr() for k, v in zip(names, params)})
    grads = [new(*p.shape) for p in params]
    gs = _capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, grads)})
    mask = torch.empty(5, B, dtype=torch.int32, device=dev)
    dz, lp = new(40, B), new(lib.apg_quad_mlp_loss_partials_count(B))
    loss = new(1)
    ws = new(lib.apg_quad_mlp_step_workspace_floats())
    part = new(lib.apg_quad_mlp_step_partials_floats(B))
    w = F.quad_loss_weights()

    def call(batch, grads_struct=gs):
        return lib.apg_quad_mlp_concurrent_step(
            s0.data_ptr(), rf.data_ptr(), rf.shape[1], DT, ctypes.byref(dyn.params),
            ctypes.byref(w), ctypes.byref(pol), batch, H, acts.data_ptr(),
            mask.data_ptr(), dz.data_ptr(), lp.data_ptr(), loss.data_ptr(),
            ctypes.byref(grads_struct), None, ws.data_ptr(), part.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert call(B) == 0
    torch.cuda.synchronize()
    assert loss.item() == want_loss.item()
    for g, k in zip(grads, F._MLP_PARAMS):
        assert torch.equal(g, want[k].view_as(g)), k
    assert call(0) == 0
    torch.cuda.synchronize()
    assert loss.item() == 0.0 and all(float(g.abs().max()) == 0.0 for g in grads)
    bad = _capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, grads)})
    bad.w_2 = None
    assert call(B, bad) == -1 and b"gradient pointer" in lib.apg_last_error_string()
    assert lib.apg_quad_mlp_concurrent_step(
        s0.data_ptr(), rf.data_ptr(), 7, DT, ctypes.byref(dyn.params), ctypes.byref(w),
        ctypes.byref(pol), B, H, acts.data_ptr(), mask.data_ptr(), dz.data_ptr(),
        lp.data_ptr(), loss.data_ptr(), ctypes.byref(gs), None, ws.data_ptr(),
        part.data_ptr(), None, None) == -1


def test_train_step_through_the_c_abi(dev):
    """apg_quad_mlp_concurrent_train_step: the gradients of the plain step,
    then buf = momentum buf + grad, p -= lr buf on the parameter and momentum
    tensors handed in - twice, against the same arithmetic in torch; NULL
    pointers inside the update and B = 0 are argument errors."""
    from apg_trajectory_tracking_amd import _capi, functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    lib = _capi.lib()
    B, lr, mom = 900, 2e-6, 0.9
    torch.manual_seed(12)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    _, inputs = _case(B, 19, dev)
    dyn = FlightmareDynamics()
    acts, s0, rf = F.quad_concurrent_prepare(*inputs)
    new = lambda *s: torch.zeros(s, device=dev)
    names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3",
             "w_out", "b_out")
    params = [p.detach() for p in F._net_params(net, F._MLP_PARAMS)]
    struct = lambda ts: _capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, ts)})
    pol = _capi.ApgMlpPolicy(**{k: v.data_ptr() for k, v in zip(names, params)})
    grads, bufs = [new(*p.shape) for p in params], [new(*p.shape) for p in params]
    gs = struct(grads)
    upd = _capi.ApgMlpSgdUpdate(lr=lr, momentum=mom, param=struct(params),
                                momentum_buf=struct(bufs))
    mask = torch.empty(5, B, dtype=torch.int32, device=dev)
    dz, lp = new(40, B), new(lib.apg_quad_mlp_loss_partials_count(B))
    loss = new(1)
    ws = new(lib.apg_quad_mlp_step_workspace_floats())
    part = new(lib.apg_quad_mlp_step_partials_floats(B))
    w = F.quad_loss_weights()

    def call(batch, update):
        return lib.apg_quad_mlp_concurrent_train_step(
            s0.data_ptr(), rf.data_ptr(), rf.shape[1], DT, ctypes.byref(dyn.params),
            ctypes.byref(w), ctypes.byref(pol), batch, H, acts.data_ptr(),
            mask.data_ptr(), dz.data_ptr(), lp.data_ptr(), loss.data_ptr(),
            ctypes.byref(gs), None, ws.data_ptr(), part.data_ptr(),
            ctypes.byref(update) if update is not None else None, None, None)
    want_p = [p.clone() for p in params]
    want_m = [torch.zeros_like(p) for p in params]
    for step in range(2):
        # the gradients the step will see: the plain call on the current weights
        torch.cuda.synchronize()
        assert call(B, None) == 0
        torch.cuda.synchronize()
        g_now = [g.clone() for g in grads]
        l_now = loss.item()
        for p_, m_, g_ in zip(want_p, want_m, g_now):
            m_.mul_(mom).add_(g_)
            p_.sub_(lr * m_)
        assert call(B, upd) == 0
        torch.cuda.synchronize()
        assert loss.item() == l_now
        for k, g, g0, p, m, wp, wm in zip(F._MLP_PARAMS, grads, g_now, params, bufs, want_p,
                                          want_m):
            assert torch.equal(g, g0), k
            assert torch.allclose(m, wm, rtol=1e-6, atol=1e-6 * float(wm.abs().max())), k
            assert torch.allclose(p, wp, rtol=1e-6, atol=1e-9), k
            assert not torch.equal(p, wp + lr * wm), k    # it moved
    broken = _capi.ApgMlpSgdUpdate(lr=lr, momentum=mom, param=struct(params),
                                   momentum_buf=struct(bufs))
    broken.momentum_buf.b_3 = None
    assert call(B, broken) == -1 and b"update" in lib.apg_last_error_string()
    assert call(0, upd) == -1 and b"B = 0" in lib.apg_last_error_string()


def test_in_sweep_kernel_agrees_with_plane_path_at_full_size_and_flags_non_finite(dev):
    """B = 65 536: the trajectory-major kernel against the plane + product path
    on every parameter gradient (two independent implementations of the same
    sums), bit-reproducible; a NaN / inf planted in one trajectory's features
    (behind the host's range check) gives a non-finite loss and NaN in EVERY
    gradient - never a finite number (the fixed-point conversion would turn a
    NaN into 0 if nothing looked)."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B = 65536
    torch.manual_seed(21)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    _, inputs = _case(B, 5, dev)
    dyn = FlightmareDynamics()
    prepared = F.quad_concurrent_prepare(*inputs)
    plan = F.QuadConcurrentStepPlan(net, prepared, DT, dyn.params)
    plan.launch()
    first = plan.flat.clone()
    plan.launch()
    assert torch.equal(plan.flat[:-1], first[:-1])                    # reproducible
    got = {k: N(v) for k, v in plan.named.items()}
    for poison in (float("nan"), float("inf")):      # one trajectory's feature
        prepared[0][3, 40000] = poison
        loss = plan.launch()
        torch.cuda.synchronize()
        assert not torch.isfinite(loss).item(), poison
        for k, v in plan.named.items():
            assert not torch.isfinite(v).any(), (poison, k)
        prepared[0][3, 40000] = 0.25
    _, planes, _ = PP.quad_concurrent_policy_grads_planes(net, *inputs, DT, dyn.params)
    for k in got:
        assert rel_err(got[k], N(planes[k])) < 5e-6, (k, rel_err(got[k], N(planes[k])))


def test_step_gradients_add_up_over_batch_halves_beyond_32_chunks(dev):
    """B = 300 011 is 1 172 workgroups = 37 chunk rows of the second stage (its
    last level sums 32 rows per round): loss and every gradient must equal
    the sum over two halves of the batch (each below 32 chunks) - sums over
    trajectories are linear.  (Until late in round 4 the last level read 32
    chunk rows and no more: batches beyond 262 144 lost the rest silently.)"""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B, cut = 300011, 150016
    torch.manual_seed(31)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    _, inputs = _case(B, 9, dev)
    dyn = FlightmareDynamics()
    whole = F.quad_concurrent_policy_grads(net, *inputs, DT, dyn.params)
    parts = [F.quad_concurrent_policy_grads(
        net, *(t[sl].contiguous() for t in inputs), DT, dyn.params)
        for sl in (slice(0, cut), slice(cut, B))]
    loss = parts[0][0].double() + parts[1][0].double()
    assert abs(whole[0].double() - loss).item() <= 2e-6 * abs(loss.item())
    for k, g in whole[1].items():
        want = N(parts[0][1][k]) + N(parts[1][1][k])
        assert rel_err(N(g), want) < 1e-5, (k, rel_err(N(g), want))
