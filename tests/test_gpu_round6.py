"""Round 6, VERDICT r5 "next" #4 (the three parity holes) and #1 on the GPU:
  * the per-row float64 arbiter on the LSTM step's parameter gradients
    (configs[4]) and on the fixed-wing policy's (configs[3]);
  * the fused fixed-wing TRAINING step at B = 131 072, H = 20 - the launch
    `secondary.wing_train_step` times - against float64 autograd over
    oracle.torch_port (scripts/train_fixed_wing.py:90-116,
    neural_control/drone_loss.py:72-82);
  * `python bench.py --gpus N` as a plain command on a box with fewer GPUs:
    an explicit error, not a hang."""
import ctypes
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_param_rows_no_worse_than_fp32, rel_err

pytestmark = pytest.mark.gpu
H, DT = 10, 0.1


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def N(t):
    return t.detach().double().cpu().numpy()


@pytest.mark.parametrize("case", ["random_init", "outlier_per_workgroup"])
def test_lstm_parameter_gradient_rows_vs_fp64_at_full_size(dev, case):
    """configs[4] (LSTM, B = 65 536, H = 10): every ROW of every parameter
    gradient of the fused step (sweeps + planes_gemm products) against float64
    autograd, float32 autograd as the yardstick - the arbiter that judges the
    concurrent and autoregressive steps (tests/test_gpu_round5.py).  The hidden
    state of rnn.py:30-33 is drawn once and fed to kernels and oracles alike."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from oracle import torch_port as tp
    B = 65536
    torch.manual_seed(31)
    net = LSTM_NEW(15, H, 9, 4, conv=1)
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=77, ref_length=2 * H)
    if case == "outlier_per_workgroup":
        d["ref"] = d["ref"].clone()
        d["ref"][7::256] *= 1.0e3
    gen = torch.Generator().manual_seed(6)
    h0 = torch.randn(B, 8, generator=gen)
    c0 = torch.randn(B, 8, generator=gen)

    def oracle(dtype):
        n = copy.deepcopy(net).to(dtype)
        n.hidden_state, n.cell_state = h0.to(dtype), c0.to(dtype)
        _, _, loss = tp.quad_recurrent_unroll(
            n, tp.QuadOracle(dtype=dtype), d["state0"].to(dtype), d["in_ref"].to(dtype),
            d["ref"].to(dtype), H, DT)
        loss.backward()
        return {k: p.grad.double().numpy() for k, p in n.named_parameters()
                if p.grad is not None}
    want, f32 = oracle(torch.float64), oracle(torch.float32)
    gnet = copy.deepcopy(net).to(dev)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    loss, _, _ = F.quad_lstm_rollout_loss(gnet, s0, in_ref, ref, DT,
                                          FlightmareDynamics().params, h0.to(dev), c0.to(dev))
    loss.backward()
    got = {k: N(p.grad) for k, p in gnet.named_parameters() if p.grad is not None}
    assert set(got) == set(want)
    for k, w in want.items():
        assert rel_err(got[k], w) < 1e-4, (k, rel_err(got[k], w))
    assert_param_rows_no_worse_than_fp32(got, f32, want, f"LSTM, {case}")


def test_wing_fused_step_full_size_vs_fp64_oracle(dev):
    """configs[3] as a TRAINING step: TrainFixedWing.train_concurrent_fused at
    B = 131 072, H = 20 (mlp_wing.hip policy forward / reverse on the matrix
    cores around wing_rollout_pk_kernel, weight products, SGD) - loss and every
    parameter gradient against float64 autograd over the oracle's op sequence,
    per-tensor at north_star's 1e-4 and per row by the arbiter."""
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import FixedWingDynamics
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    from oracle import torch_port as tp
    B, Hw, dt = 131072, 20, 0.05
    cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B,
               state_size=12, horizon=Hw, ref_dim=3, action_dim=4,
               learning_rate_controller=1e-12, system="fixed_wing", modified_params={})
    data = SyntheticWingDataset(B, Hw, dt, seed=17, device=dev)
    torch.manual_seed(6)
    net = Net(9, 1, 3, 4 * Hw, conv=False)
    t = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), dict(cfg))
    t.net = copy.deepcopy(net).to(dev)
    t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-12, momentum=0.9)
    loss = t.train_concurrent_fused(data.normed_states, data.states, data.in_ref_states,
                                    data.ref_states)
    assert loss is not None, "the fused fixed-wing step did not apply"
    got = {k: N(p.grad) for k, p in t.net.named_parameters() if p.grad is not None}

    def oracle(dtype):
        n = copy.deepcopy(net).to(dtype)
        acts = torch.sigmoid(n(data.normed_states.cpu().to(dtype),
                               data.in_ref_states.cpu().to(dtype))).reshape(-1, Hw, 4)
        states = tp.unroll(tp.WingOracle(dtype=dtype), data.states.cpu().to(dtype), acts, dt)
        l = tp.fixed_wing_mpc_loss(states, data.ref_states.cpu().to(dtype), acts)
        l.backward()
        return float(l), {k: p.grad.double().numpy() for k, p in n.named_parameters()
                          if p.grad is not None}
    l64, want = oracle(torch.float64)
    _, f32 = oracle(torch.float32)
    assert abs(loss.item() - l64) / l64 < 1e-5, (loss.item(), l64)
    assert set(got) == set(want)
    for k, w in want.items():
        assert rel_err(got[k], w) < 1e-4, (k, rel_err(got[k], w))
    assert_param_rows_no_worse_than_fp32(got, f32, want, "fixed wing fused step, B = 131 072")


def test_bench_gpus_beyond_the_node_is_an_explicit_error(dev):
    """`python bench.py --gpus N` with N above
    torch.cuda.device_count() says so and exits at once (no rendezvous, no RCCL
    init that would wait for ranks that cannot exist)."""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n),
                        "--steps", "20", "--warmup", "5"], capture_output=True, text=True,
                       timeout=240, env=env)
    assert r.returncode != 0
    assert f"--gpus {n}" in r.stderr and "GPU(s)" in r.stderr, r.stderr[-500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_resident_tables_after_a_write_through_data_need_invalidate(dev):
    """a parameter written through `p.data` bumps a version counter
    the step plan cannot see - its resident operand tables would stay stale.
    `plan.invalidate()` (TrainBase.run_epoch calls it for every plan at the start
    of an epoch) makes the next launch pack again.  Against a plan that packs at
    every step: bit for bit.  (A REPLACED storage, `p.data = other`, is outside a
    plan's contract - its argument structs hold the parameters' addresses;
    TrainBase's step signature carries them and rebuilds the plan.)"""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B = 1024
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=6, ref_length=H)
    st, inr, rf = (d[k].to(dev).contiguous() for k in ("state0", "in_ref", "ref"))
    normed = state_preprocessing(st).contiguous()
    params = FlightmareDynamics().params
    prepared = F.quad_concurrent_prepare(normed, st, inr, rf)
    outs = []
    for resident in (True, False):
        torch.manual_seed(1)
        net = Net(15, H, 9, 4 * H, conv=1).to(dev)
        bufs = {n: torch.zeros_like(p) for n, p in net.named_parameters() if n in F._MLP_PARAMS}
        plan = F.QuadConcurrentStepPlan(net, prepared, DT, params,
                                        update=(2e-4 / B, 0.9, bufs))
        plan.resident_tables = resident
        flags, losses = [], []
        for i in range(7):
            if i == 2:          # weight clipping behind autograd's back
                net.fc2.weight.data.clamp_(-0.05, 0.05)
                plan.invalidate()
            if i == 4:          # ... and a soft update, again through .data
                net.fc3.bias.data.mul_(0.5)
                plan.invalidate()
            losses.append(float(plan.launch()))
            flags.append(plan._keep["upd"].resident)
        if resident:
            assert flags == [1, 2, 3, 2, 3, 2, 2], flags
        outs.append((losses, [p.detach().clone() for p in net.parameters()]))
    (la, pa), (lb, pb) = outs
    assert la == lb and np.isfinite(la).all()
    assert all(torch.equal(x, y) for x, y in zip(pa, pb))


def _lstm_trainer(dev, B, in_kernel, seed=4):
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(delta_t=0.1, delta_t_train=0.1, epoch_size=1000, self_play=1, batch_size=B,
               state_size=12, horizon=10, train_mode="LSTM", ref_dim=9, action_dim=4,
               learning_rate_controller=1e-7, system="quad", modified_params={})
    t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
    torch.manual_seed(seed)
    t.net = LSTM_NEW(15, H, 9, 4, conv=1).to(dev)
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=9, ref_length=t.ref_length)

    class Shard:
        states, in_ref_states, ref_states = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    with torch.no_grad():
        Shard.normed_states = state_preprocessing(Shard.states)
    t.state_data, t.static_shard = Shard, True
    gen = torch.Generator().manual_seed(3)
    hc = torch.randn(2, 8, B, generator=gen).to(dev)

    def fixed_reset(batch_size=1, generator=None, net=t.net):
        net.hidden_state, net.cell_state = hc[0].t(), hc[1].t()
    t.net.reset_hidden_state = fixed_reset
    t.graph_steps = False
    t.in_kernel_update = in_kernel
    t.resident_tables = in_kernel
    t.init_optimizer()
    step = lambda: t.train_recurrent_model(None, Shard.states, Shard.in_ref_states,
                                           Shard.ref_states)
    return t, step


@pytest.mark.parametrize("B", [1000, 65536])
def test_lstm_step_tail_equals_the_separate_launches(dev, B):
    """the LSTM step's tail - gradients into place, momentum
    SGD, the next step's operand tables, the loss - as ONE launch
    (apg_quad_lstm_step_tail) against what it replaces: three elementwise
    launches, torch's fused SGD, the loss reduction and the two table packs of
    the next step's sweeps.  Five steps: losses, gradients, parameters and
    momentum buffers bit for bit; the resident tables are packed once."""
    from apg_trajectory_tracking_amd import functional as F
    outs = []
    for in_kernel in (True, False):
        t, step = _lstm_trainer(dev, B, in_kernel)
        losses = [float(step()) for _ in range(5)]
        grads = {k: p.grad.detach().clone() for k, p in t.net.named_parameters()
                 if p.grad is not None}
        state = t.optimizer_controller.state
        bufs = {k: state[p]["momentum_buffer"].clone() for k, p in t.net.named_parameters()
                if p in state and state[p].get("momentum_buffer") is not None}
        outs.append((losses, grads, {k: p.detach().clone() for k, p in
                                     t.net.named_parameters()}, bufs))
        tab = F._LSTM_TABLES.get(t.net) if F._LSTM_TABLES is not None else None
        if in_kernel:
            assert tab is not None and tab.packs == 1, tab and tab.packs
            assert t._in_kernel_update(True, names=F._LSTM_PARAMS, tensors=tuple(
                dict(t.net.named_parameters())[n] for n in F._LSTM_PARAMS)) is not None
        else:
            assert tab is None
    (la, ga, pa, ba), (lb, gb, pb, bb) = outs
    assert la == lb and np.isfinite(la).all() and la[-1] != la[0]
    assert set(ga) == set(gb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    assert set(ba) == set(bb) == set(F._LSTM_PARAMS)
    for xs, ys in ((pa, pb), (ba, bb)):
        for k in xs:
            assert torch.equal(xs[k], ys[k]), k


def test_lstm_resident_tables_notice_foreign_writes(dev):
    """A parameter written from outside (in-place version counter) makes the next
    step pack again; a write through `.data` needs `invalidate()` - run_epoch's
    epoch start does it."""
    from apg_trajectory_tracking_amd import functional as F
    t, step = _lstm_trainer(dev, 512, True)
    ref, rstep = _lstm_trainer(dev, 512, False)
    tab = lambda: F._LSTM_TABLES.get(t.net)
    for i in range(5):
        if i == 2:
            for n in (t.net, ref.net):
                with torch.no_grad():
                    n.lstm.bias_ih.mul_(1.5)
        if i == 4:
            for n in (t.net, ref.net):
                n.fc_out.weight.data.mul_(0.5)
            tab().invalidate()
        assert float(step()) == float(rstep()), i
    assert tab().packs == 3
    for a, b in zip(t.net.parameters(), ref.net.parameters()):
        assert torch.equal(a, b)


def test_config2_eight_shards_summed_equal_the_whole_batch(dev):
    """BASELINE configs[2] / SURVEY §8(e) "verification", on ONE GPU: the
    autoregressive step's loss and every parameter gradient of a 524 288-trajectory
    batch (a) as eight contiguous shards of 65 536 - what the eight ranks compute -
    summed in rank order (what the all-reduce(sum) of the flat message does: the
    reference's losses are sums over the batch, so no rescaling) and (b) as ONE call
    on the whole batch (beyond the per-launch limit: processed in chunks).  Same
    policy, same data: rel. error <= 1e-5 (float32 summation order only)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.parallel import shard_range
    B, world = 524288, 8
    torch.manual_seed(12)
    net = Net(15, H, 9, 4, conv=1).to(dev)
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=40, ref_length=2 * H)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    dyn = FlightmareDynamics()
    flat_sum = None
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        assert hi - lo == 65536
        loss, gr, flat = F.quad_mlp_rollout_grads(
            net, s0[lo:hi].contiguous(), in_ref[lo:hi].contiguous(), ref[lo:hi].contiguous(),
            DT, dyn.params)
        flat = flat.clone()
        flat[-1] = loss.reshape(())          # the message: gradients + loss slot
        flat_sum = flat.double() if flat_sum is None else flat_sum + flat.double()
    names = [(k, v.numel()) for k, v in gr.items()]
    loss_w, gr_w, flat_w = F.quad_mlp_rollout_grads(net, s0, in_ref, ref, DT, dyn.params)
    assert abs(flat_sum[-1].item() - loss_w.item()) <= 1e-5 * abs(loss_w.item())
    off = 0
    for k, n in names:
        got = flat_sum[off:off + n].cpu().numpy()
        want = gr_w[k].double().reshape(-1).cpu().numpy()
        assert rel_err(got, want) < 1e-5, (k, rel_err(got, want))
        off += n
    assert off + 1 == flat_sum.numel()


@pytest.mark.parametrize("B", [37, 1000, 4100])
def test_lstm_gate_weight_gradients_recompute_the_conv_inputs(dev, B):
    """apg_quad_lstm_gate_wgrad (round 6): [dW_ih | dW_hh], db, dW_out, db_out of
    LSTM_NEW (neural_control/models/rnn.py:35-51; `loss.backward()` of
    scripts/train_base.py:200-204) from the reverse sweep's cotangent planes with
    the 160 relu(conv) inputs RECOMPUTED from the reference window - against the
    same sums in float64 over x rebuilt on the host from the planes the sweeps
    left (features, h_prev, h_new) and torch's own conv1d on the windows; ragged
    batches (tail masks, one and several workgroups).  The reverse sweep's
    cot_amax is the per-group maximum the kernel scales by, and two runs agree
    to the bit (fixed summation order)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=B, ref_length=20)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    g = torch.Generator().manual_seed(B)
    h0, c0 = (torch.randn(B, 8, generator=g).to(dev) for _ in range(2))
    torch.manual_seed(B)
    net = LSTM_NEW(15, H, 9, 4, conv=1).to(dev)
    dyn = FlightmareDynamics()
    runs = []
    for _ in range(2):
        ctx = F._DirectCtx()
        with torch.no_grad():
            F._QuadLstmRolloutLoss.forward(
                ctx, s0, in_ref, ref, h0, c0, *F._net_params(net, F._LSTM_PARAMS), DT,
                dyn.params, F.quad_loss_weights())
            flat, gr = F._lstm_param_grads(ctx.saved_tensors, ctx.dims)
        runs.append((flat.clone(), {k: v.clone() for k, v in gr.items()}, ctx))
    assert torch.equal(runs[0][0][:-1], runs[1][0][:-1])       # bit-reproducible
    _, gr, ctx = runs[0]
    refbuf, acts, d_gates, d_zout, _, cot_amax = ctx.saved_tensors[:6]
    n = H * B
    # the maxima the kernel scales by: per group of 32 trajectories over all steps
    groups = (B + 31) // 32
    dg = d_gates.view(32, H, B).abs().amax(dim=(0, 1))
    dz = d_zout.view(4, H, B).abs().amax(dim=(0, 1))
    pad = groups * 32 - B
    dg, dz = (torch.nn.functional.pad(v, (0, pad)).view(groups, 32).amax(dim=1) for v in (dg, dz))
    assert torch.equal(cot_amax.view(groups, 2)[:, 0], dg)
    assert torch.equal(cot_amax.view(groups, 2)[:, 1], dz)
    # x of every (step, trajectory) in float64: features | relu(conv1d(window)) | h_prev
    inr = refbuf[:2 * H * 9].view(2 * H, 9, B).double()
    st_all = refbuf[2 * H * 9:].view(H + 1, 12, B).double()
    conv = torch.nn.Conv1d(9, 20, 3).double().to(dev)
    conv.load_state_dict({k: v.double() for k, v in net.conv_ref.state_dict().items()})
    xs = []
    for k in range(H):
        win = inr[k:k + H].clone()                      # [H, 9, B]
        win[:, :3] -= st_all[k, :3]                     # relative to the current position
        c = torch.relu(conv(win.permute(2, 1, 0)))      # [B, 20, 8]
        xs.append(c.reshape(B, 160).t())
    xc = torch.stack(xs, dim=1).reshape(160, n)         # column = step * B + trajectory
    a = acts.double()
    x = torch.cat([a[:15], xc, a[15:23]])               # 183 rows
    want = {"lstm.weight_ih": (d_gates.double() @ x.t())[:, :175],
            "lstm.weight_hh": (d_gates.double() @ x.t())[:, 175:],
            "lstm.bias_ih": d_gates.double().sum(dim=1),
            "fc_out.weight": d_zout.double() @ a[31:39].t(),
            "fc_out.bias": d_zout.double().sum(dim=1)}
    # the conv weights' gradient (apg_quad_lstm_conv_wgrad) from the diagonal sums the
    # reverse sweep left: G[ch][hi][tau] against window row 4 hi + tau + t, the
    # position sums P[ch][k] against the current position (and ones: the bias)
    d_conv = ctx.saved_tensors[4].double()
    G = d_conv[:520].view(20, 2, 13, B)
    P = d_conv[520:].view(20, H, B)
    dw = torch.zeros(20, 9, 3, dtype=torch.float64, device=dev)
    for hi_ in range(2):
        for tau in range(13):
            for t in range(3):
                dw[:, :, t] += G[:, hi_, tau] @ inr[4 * hi_ + tau + t].t()
    dw[:, :3] -= torch.einsum("ckb,kjb->cj", P, st_all[:H, :3])[:, :, None]
    want["conv_ref.weight"] = dw
    want["conv_ref.bias"] = P.sum(dim=(1, 2))
    for k, w in want.items():
        scale = w.abs().max().item() + 1e-300
        assert (gr[k].double() - w).abs().max().item() / scale < 2e-6, k
    # apg_quad_lstm_wgrads (what _lstm_param_grads calls: both products, one sum
    # launch) against the two entry points with a sum launch each: bit for bit
    from apg_trajectory_tracking_amd._capi import ApgLstmPolicy, check, lib, ptr, stream_of
    pw8 = dict(zip(("conv_w", "conv_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out"),
                   ctx.saved_tensors[6:14]))
    pol = ctypes.byref(ApgLstmPolicy(**{k: ptr(v) for k, v in pw8.items()}))
    new = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
    tab = new(lib().apg_quad_lstm_workspace_floats())
    st = refbuf[2 * H * 9:]
    ih_hh, b_ih, w_out, b_out = new(32, 183), new(32), new(4, 8), new(4)
    conv_w, conv_pos, conv_b = new(20, 27), new(20, 3), new(20)
    check(lib().apg_quad_lstm_gate_wgrad(
        ptr(st[:12]), ptr(st[12:]), ptr(refbuf[:2 * H * 9]), ptr(acts), ptr(d_gates),
        ptr(d_zout), ptr(cot_amax), pol, ptr(tab), B, H,
        ptr(new(max(1, lib().apg_quad_lstm_gate_wgrad_partials_floats(B)))),
        ptr(ih_hh), ptr(b_ih), ptr(w_out), ptr(b_out), stream_of(acts)), "gate_wgrad")
    check(lib().apg_quad_lstm_conv_wgrad(
        ptr(ctx.saved_tensors[4]), ptr(refbuf[:2 * H * 9]), ptr(st), B, H,
        ptr(new(max(1, lib().apg_quad_lstm_conv_wgrad_partials_floats(B)))),
        ptr(conv_w), ptr(conv_pos), ptr(conv_b), stream_of(acts)), "conv_wgrad")
    assert torch.equal(ih_hh[:, :175], gr["lstm.weight_ih"])
    assert torch.equal(ih_hh[:, 175:], gr["lstm.weight_hh"])
    assert torch.equal(b_ih, gr["lstm.bias_ih"]) and torch.equal(b_out, gr["fc_out.bias"])
    assert torch.equal(w_out, gr["fc_out.weight"]) and torch.equal(conv_b, gr["conv_ref.bias"])
    cw = conv_w.view(20, 9, 3).clone()
    cw[:, :3] -= conv_pos[:, :, None]
    assert torch.equal(cw, gr["conv_ref.weight"])


def test_lstm_wgrads_finish_argument_checks(dev):
    """apg_quad_lstm_wgrads(finish = ...): the tensors the sum launch would write
    are checked before anything is launched - gradients always, parameters and
    momentum buffers when the update is asked for; a finish with B = 0 is an
    error (nothing would be summed, the caller's tail would never run)."""
    from apg_trajectory_tracking_amd import _capi
    from apg_trajectory_tracking_amd._capi import lib, ptr
    G = _capi.ApgLstmPolicyGrads
    buf = torch.zeros(8192, device=dev)
    full = G(**{n: ptr(buf) for n, _ in G._fields_})
    part = G(**{n: ptr(buf) for n, _ in G._fields_ if n != "w_hh"})
    call = lambda t, B=64: lib().apg_quad_lstm_wgrads(
        *([None] * 9), None, None, B, 10, *([None] * 9), ctypes.byref(t), None)
    err = lambda: lib().apg_last_error_string().decode()
    assert call(_capi.ApgLstmStepTail(grad=part)) != 0 and "gradient" in err()
    assert call(_capi.ApgLstmStepTail(grad=full, update=1, param=full)) != 0
    assert "momentum" in err()
    # complete finish, B = 0: the products zero the gradients, nothing to finish
    t = _capi.ApgLstmStepTail(grad=full, update=1, param=full, mom=full)
    out = [torch.zeros(32 * 183, device=dev) for _ in range(7)]
    rc = lib().apg_quad_lstm_wgrads(
        *([None] * 9), None, None, 0, 10, None, None, *[ptr(o) for o in out],
        ctypes.byref(t), None)
    assert rc != 0 and "B > 0" in err()


@pytest.mark.parametrize("B,ref_cols", [(1000, 9), (129, 6), (4096, 9)])
def test_lstm_sweeps_read_the_batch_rows_through_the_index(dev, B, ref_cols):
    """TrainBase.run_epoch's batch selection (scripts/train_base.py:191-194:
    `batch = data[index]`) inside the LSTM sweeps (round 6:
    apg_quad_lstm_rollout_fwd_rows / _bwd_rows): loss, every parameter gradient
    and the rollout equal the gather pass + plane sweeps TO THE BIT - the same
    numbers reach the same arithmetic - for a shuffled index with repeated rows
    out of a larger data set, ragged batches, both reference layouts."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    n = 3 * B + 17
    d = synthetic.quad_polynomial_batch(n, H, DT, seed=B, ref_length=24)
    s0, in_ref = d["state0"].to(dev), d["in_ref"].to(dev)
    ref = d["ref"][:, :, :ref_cols].contiguous().to(dev)
    g = torch.Generator().manual_seed(B)
    index = torch.randint(0, n, (B,), generator=g).to(dev)
    index[1] = index[0]                                       # a repeated row
    h0, c0 = (torch.randn(B, 8, generator=g).to(dev) for _ in range(2))
    torch.manual_seed(B)
    net = LSTM_NEW(15, H, 9, 4, conv=1).to(dev)
    dyn = FlightmareDynamics()
    out = []
    for rows in (False, True):
        loss, gr, flat = F.quad_lstm_rollout_grads(net, s0, in_ref, ref, DT, dyn.params,
                                                   h0, c0, index=index, rows_in_kernel=rows)
        out.append((loss.clone(), flat.clone()))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1][:-1], out[1][1][:-1])
    assert torch.isfinite(out[1][1][:-1]).all() and out[1][1][:-1].abs().max() > 0


@pytest.mark.parametrize("mode", ["ar", "lstm"])
def test_recurrent_forward_as_shipped_in_place_window(dev, mode):
    """SURVEY §8a A4 `legacy_inplace_ref` (VERDICT r5 missing #3): the recurrent
    unroll with the reference window shifted IN PLACE as scripts/train_drone.py:
    138-142 ships it - forward only - inside the fused kernels
    (apg_quad_mlp_rollout_fwd_inplace_ref / apg_quad_lstm_rollout_fwd_inplace_ref)
    against what the reference itself computed (G4b: tests/golden/
    quad_recurrent_inplace.npz, made by the reference's own loop), and at a ragged
    1 000 against the oracle's restatement in float64; the caller's in_ref stays
    as it was, and the pinned (copied-window) forward differs."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from conftest import load_golden
    from oracle import torch_port as tp
    g, gi = load_golden("quad_recurrent.npz"), load_golden("quad_recurrent_inplace.npz")
    net = (LSTM_NEW if mode == "lstm" else Net)(15, H, 9, 4, conv=1)
    net.load_state_dict({k[len(mode) + 3:]: torch.from_numpy(g[k])
                         for k in g.files if k.startswith(mode + ".w.")})
    dyn = FlightmareDynamics()
    s0, in_ref = (torch.from_numpy(g[k]).to(dev) for k in ("state0", "in_ref"))
    hc = ((torch.from_numpy(g["lstm_h0"]).to(dev), torch.from_numpy(g["lstm_c0"]).to(dev))
          if mode == "lstm" else (None, None))
    before = in_ref.clone()
    states, actions = F.quad_recurrent_forward_inplace_ref(
        net.to(dev), s0, in_ref, float(g["dt"]), dyn.params, *hc)
    assert torch.equal(in_ref, before)
    assert rel_err(N(states), gi[f"{mode}.states"]) < 2e-5
    assert rel_err(N(actions), gi[f"{mode}.actions"]) < 2e-5
    assert rel_err(N(states), g[f"{mode}.states"]) > 1e-3      # not the pinned semantics
    # the trainer's forward-only entry: the reference's loss of that loop
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(delta_t=float(g["dt"]), delta_t_train=float(g["dt"]), epoch_size=32, self_play=0,
               batch_size=32, state_size=12, horizon=H, ref_dim=9, action_dim=4,
               train_mode="LSTM" if mode == "lstm" else "autoregressive",
               learning_rate_controller=1e-9, system="quad", modified_params={})
    t = TrainDrone(dyn, dyn, cfg)
    t.net = net.to(dev)
    if mode == "lstm":
        def fixed_reset(batch_size=1, generator=None):
            net.hidden_state, net.cell_state = hc[0].clone(), hc[1].clone()
        net.reset_hidden_state = fixed_reset
    loss, _, _ = t.recurrent_forward_as_shipped(s0, in_ref, torch.from_numpy(g["ref"]).to(dev))
    assert abs(loss.item() - gi[f"{mode}.loss"]) / gi[f"{mode}.loss"] < 2e-5
    assert torch.equal(in_ref, before)
    # a ragged batch against the oracle in float64
    B = 1000
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=77, ref_length=20)
    gen = torch.Generator().manual_seed(5)
    h0, c0 = torch.randn(B, 8, generator=gen), torch.randn(B, 8, generator=gen)
    net64 = copy.deepcopy(net).cpu().double()
    if mode == "lstm":
        net64.hidden_state, net64.cell_state = h0.double(), c0.double()
    with torch.no_grad():
        inter, acts, _ = tp.quad_recurrent_unroll(
            net64, tp.QuadOracle(dtype=torch.float64), d["state0"].double(),
            d["in_ref"].double(), d["ref"].double(), H, DT, legacy_inplace_ref=True)
    states, actions = F.quad_recurrent_forward_inplace_ref(
        net.to(dev), d["state0"].to(dev), d["in_ref"].to(dev), DT, dyn.params,
        *((h0.to(dev), c0.to(dev)) if mode == "lstm" else (None, None)))
    assert rel_err(N(states), inter.numpy()) < 2e-5
    assert rel_err(N(actions), acts.numpy()) < 2e-5
