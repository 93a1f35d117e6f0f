"""Round 5: the autoregressive training step with its weight gradients
accumulated INSIDE the reverse sweep (apg_quad_mlp_rollout_train_step;
csrc/mlp_rollout.hip, mlp_rollout_bwd_tm_kernel) - no cotangent planes, no
planes_gemm launches.  One loss.backward() of the reference yields every
parameter gradient (scripts/train_drone.py:113-173); this path must too, to the
same 1e-4 as the plane + product path it replaced (rounds 1-4; an independent
implementation of the same sums that lives in tests/plane_path.py since round 6 -
the package has one path - and is the comparison here)."""
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
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=seed, ref_length=2 * H)
    return d, tuple(d[k].to(dev) for k in ("state0", "in_ref", "ref"))


def _oracle_grads(net, d, dtype):
    from oracle import torch_port as tp
    n = copy.deepcopy(net).to(dtype).cpu()
    _, _, loss = tp.quad_recurrent_unroll(
        n, tp.QuadOracle(dtype=dtype), d["state0"].to(dtype), d["in_ref"].to(dtype),
        d["ref"].to(dtype), H, DT)
    loss.backward()
    return loss.item(), {k: p.grad.double().numpy() for k, p in n.named_parameters()
                         if p.grad is not None}


# 1: one lane; 31 / 77: part of a wave; 256: exactly one workgroup; 257: a
# second workgroup with one trajectory; 300, 4113: ragged last workgroups;
# 8192 + 3: more than one chunk of the second stage (32 workgroups)
@pytest.mark.parametrize("B", [1, 31, 77, 256, 257, 300, 4113, 8195])
def test_ar_in_sweep_gradients_vs_fp64_oracle_and_plane_products(dev, B):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    torch.manual_seed(8)
    net = Net(15, H, 9, 4, conv=1)
    gnet = copy.deepcopy(net).to(dev)
    d, inputs = _case(B, 200 + B, dev)
    dyn = FlightmareDynamics()
    res = []
    for fn in (F.quad_mlp_rollout_grads, PP.quad_mlp_rollout_grads_planes):
        loss, grads, flat = fn(gnet, *inputs, DT, dyn.params)
        assert flat.numel() == sum(g.numel() for g in grads.values()) + 1   # + loss slot
        res.append((loss.item(), {k: N(v) for k, v in grads.items()}))
    loss64, want = _oracle_grads(net, d, torch.float64)
    (l1, g1), (l0, g0) = res
    assert l1 == l0                               # the same arithmetic up to the loss
    assert abs(l1 - loss64) / abs(loss64) < 1e-5
    assert set(g1) == set(want)
    for k, w in want.items():
        assert rel_err(g1[k], w) < 1e-4, (k, rel_err(g1[k], w))
        assert rel_err(g1[k], g0[k]) < 2e-5, (k, rel_err(g1[k], g0[k]))


def test_ar_in_sweep_is_deterministic_and_feeds_autograd(dev):
    """Fixed-point LDS accumulators, a global accumulator element owned by ONE
    thread that adds the steps in order, a fixed-order second stage: equal
    inputs give equal bits, at a batch of several workgroups and chunks as
    well; loss.backward() of the autograd entry point delivers the same
    gradients (scaled by the upstream cotangent) and the state cotangent of the
    plane path."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    torch.manual_seed(3)
    net = Net(15, H, 9, 4, conv=1).to(dev)
    _, inputs = _case(9000, 5, dev)
    dyn = FlightmareDynamics()
    l0, g0, f0 = F.quad_mlp_rollout_grads(net, *inputs, DT, dyn.params)
    f0 = f0.clone()
    for _ in range(3):
        l1, g1, f1 = F.quad_mlp_rollout_grads(net, *inputs, DT, dyn.params)
        assert torch.equal(f1[:-1], f0[:-1]) and torch.equal(l1, l0)
    net.zero_grad()
    s0 = inputs[0].clone().requires_grad_(True)
    loss, states, actions = F.quad_mlp_rollout_loss(net, s0, *inputs[1:], DT, dyn.params)
    (2.5 * loss).backward()
    p1 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    _, gp, _, gs0 = PP.quad_mlp_rollout_grads_planes(net, *inputs, DT, dyn.params,
                                                     want_state_grad=True)
    assert rel_err(N(s0.grad), 2.5 * N(gs0)) < 1e-6
    assert set(p1) == set(gp) == set(g0)
    for k in p1:
        assert torch.allclose(p1[k], 2.5 * g0[k], rtol=1e-6, atol=0), k


def test_ar_in_sweep_full_size_bits_and_rows(dev):
    """BASELINE configs[2] per rank, 65 536 trajectories: bit-reproducible run
    to run, and every OUTPUT ROW of every parameter gradient as good as float32
    autograd's, the float64 oracle arbitrating (VERDICT r4 next #2a)."""
    from conftest import assert_param_rows_no_worse_than_fp32
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B = 65536
    torch.manual_seed(5)
    net = Net(15, H, 9, 4, conv=1)
    gnet = copy.deepcopy(net).to(dev)
    d, inputs = _case(B, 23, dev)
    dyn = FlightmareDynamics()
    l0, g0, f0 = F.quad_mlp_rollout_grads(gnet, *inputs, DT, dyn.params)
    f0 = f0.clone()
    l1, g1, f1 = F.quad_mlp_rollout_grads(gnet, *inputs, DT, dyn.params)
    assert torch.equal(f1[:-1], f0[:-1]) and torch.equal(l1, l0)
    _, want = _oracle_grads(net, d, torch.float64)
    _, f32 = _oracle_grads(net, d, torch.float32)
    got = {k: N(v) for k, v in g1.items()}
    for k, w in want.items():
        assert rel_err(got[k], w) < 1e-4, (k, rel_err(got[k], w))
    assert_param_rows_no_worse_than_fp32(got, f32, want, "autoregressive, 65 536")


def test_ar_step_through_the_c_abi(dev):
    """apg_quad_mlp_rollout_train_step called directly (plain pointers): the
    result of the Python entry point; B = 0 zeroes the gradients and the loss;
    argument errors come back as APG_ERR_ARG; the in-kernel SGD update is
    torch.optim.SGD's step on the same gradients."""
    from apg_trajectory_tracking_amd import _capi, functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    lib = _capi.lib()
    B = 700
    torch.manual_seed(11)
    net = Net(15, H, 9, 4, conv=1).to(dev)
    _, inputs = _case(B, 17, dev)
    dyn = FlightmareDynamics()
    want_loss, want, _ = F.quad_mlp_rollout_grads(net, *inputs, DT, dyn.params)
    refbuf, inr, s0, states, rf = F.quad_recurrent_prepare(*inputs)
    new = lambda *s: torch.full(s, 7.0, device=dev)
    names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3",
             "w_out", "b_out")
    params = [p.detach().clone().contiguous() for p in F._net_params(net, F._MLP_PARAMS)]
    pol = _capi.ApgMlpPolicy(**{k: v.data_ptr() for k, v in zip(names, params)})
    grads = [new(*p.shape) for p in params]
    gs = _capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, grads)})
    mask = torch.empty(5, H * B, dtype=torch.int32, device=dev)
    acts, actions = new(431, H * B), new(H, 4, B)
    lp, loss = new(lib.apg_quad_mlp_loss_partials_count(B)), new(1)
    ws = new(lib.apg_quad_mlp_rollout_step_workspace_floats())
    part = new(lib.apg_quad_mlp_rollout_step_partials_floats(B))
    w = F.quad_loss_weights()
    st = torch.cuda.current_stream(dev).cuda_stream

    def call(B_, update=None, grads_=gs):
        return lib.apg_quad_mlp_rollout_train_step(
            s0.data_ptr(), inr.data_ptr(), rf.data_ptr(), rf.shape[1], DT,
            ctypes.byref(dyn.params), ctypes.byref(w), ctypes.byref(pol), B_, H,
            states.data_ptr(), actions.data_ptr(), acts.data_ptr(), mask.data_ptr(),
            lp.data_ptr(), loss.data_ptr(),
            None if grads_ is None else ctypes.byref(grads_), None, ws.data_ptr(),
            part.data_ptr(), update, st)

    assert call(B) == 0, lib.apg_last_error_string()
    torch.cuda.synchronize()
    assert torch.equal(loss.reshape(()), want_loss)
    for g, n in zip(grads, F._MLP_PARAMS):
        assert torch.equal(g, want[n]), n
    # the update inside the second stage = optimizer.step() on these gradients
    opt_p = [p.clone().requires_grad_(True) for p in params]
    opt = torch.optim.SGD(opt_p, lr=1e-7, momentum=0.9, fused=True)   # (double, one rounding)
    bufs = [torch.zeros_like(p) for p in params]
    for step in range(2):
        for p, g in zip(opt_p, grads):
            p.grad = g.clone()
        upd = _capi.ApgMlpSgdUpdate(
            lr=1e-7, momentum=0.9,
            param=_capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, params)}),
            momentum_buf=_capi.ApgMlpPolicyGrads(**{k: v.data_ptr()
                                                    for k, v in zip(names, bufs)}))
        before = [g.clone() for g in grads]
        assert call(B, ctypes.byref(upd)) == 0, lib.apg_last_error_string()
        torch.cuda.synchronize()
        if step == 0:
            for g, b_ in zip(grads, before):
                assert torch.equal(g, b_)          # same parameters -> same gradients
        for p, g in zip(opt_p, grads):
            p.grad = g.clone()
        opt.step()
        for p, q, n in zip(params, opt_p, F._MLP_PARAMS):
            assert torch.equal(p, q.detach()), (step, n)
    # B = 0: zero gradients and loss
    assert call(0) == 0
    torch.cuda.synchronize()
    assert loss.item() == 0.0 and all(float(g.abs().max()) == 0.0 for g in grads)
    assert call(B, grads_=None) != 0 and b"gradient" in lib.apg_last_error_string()
    assert call(-1) != 0
