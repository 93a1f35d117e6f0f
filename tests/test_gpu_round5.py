"""Round 5, VERDICT r4 "next" items 2, 6, 9 and ADVICE r4 on the GPU:
  * parameter gradients arbitrated PER OUTPUT ROW by the float64 oracle, with a
    x1e3 cotangent outlier in every workgroup and with the shipped controller's
    weights (the fixed-point accumulators' unit is per workgroup: a max norm
    over a tensor cannot see what that does to a small row);
  * the N > 1 step's scheduling (graph A -> RCCL all-reduce -> graph B) on ONE
    GPU with a live nccl process group of world size 1;
  * graph replay or stream order: measured once per train mode, not a constant;
  * a graph capture's warm-up steps leave no trace in ANY optimizer's state;
    Adam is made capturable; a gather plan never replays a stale copy."""
import copy
import os
import tempfile
import warnings

import numpy as np
import pytest
import torch

from conftest import assert_param_rows_no_worse_than_fp32, load_golden, rel_err

pytestmark = pytest.mark.gpu
H, DT = 10, 0.1
QUAD_CFG = dict(
    delta_t=0.1, delta_t_train=0.1, epoch_size=1000, self_play=1,
    batch_size=64, state_size=12, horizon=10, train_mode="concurrent",
    ref_dim=9, action_dim=4, learning_rate_controller=1e-7, system="quad",
    modified_params={},
)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def N(t):
    return t.detach().double().cpu().numpy()


def _shipped_quad_net(n_out):
    """The controller the reference ships (trained_models/quad, G9), its head cut
    to the first `n_out` rows (the autoregressive policy uses the first action,
    as the reference's evaluator does with a concurrent net)."""
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    ck = load_golden("checkpoints.npz")
    sd = {k[len("quad.w."):]: torch.from_numpy(ck[k]) for k in ck.files
          if k.startswith("quad.w.")}
    net = Net(15, H, 9, n_out, conv=1)
    sd["fc_out.weight"], sd["fc_out.bias"] = sd["fc_out.weight"][:n_out], sd["fc_out.bias"][:n_out]
    net.load_state_dict(sd)
    return net


def _oracle(net, d, mode, dtype):
    from oracle import torch_port as tp
    n = copy.deepcopy(net).to(dtype).cpu()
    s = d["state0"].to(dtype)
    if mode == "concurrent":
        acts = torch.sigmoid(n(tp.quad_state_features(s),
                               d["in_ref"][:, :H].to(dtype))).reshape(-1, H, 4)
        loss = tp.quad_mpc_loss(tp.unroll(tp.QuadOracle(dtype=dtype), s, acts, DT),
                                d["ref"][:, :H].to(dtype), acts)
    else:
        _, _, loss = tp.quad_recurrent_unroll(n, tp.QuadOracle(dtype=dtype), s,
                                              d["in_ref"].to(dtype), d["ref"].to(dtype), H, DT)
    loss.backward()
    return {k: p.grad.double().numpy() for k, p in n.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("mode", ["concurrent", "autoregressive"])
@pytest.mark.parametrize("case", ["random_init", "outlier_per_workgroup", "shipped_controller"])
def test_parameter_gradient_rows_vs_fp64_at_full_size(dev, mode, case):
    """B = 65 536.  `outlier_per_workgroup`: every 256th trajectory (one per
    workgroup of the reverse kernels) has its loss reference 1e3 x further away
    - its cotangents are 1e3-1e4 x the others' and set the workgroup's
    fixed-point unit for all 256.  `shipped_controller`: the trained weights
    (saturated tanh units, heavy-tailed cotangents) instead of a random init.
    Kernel and float32 autograd against float64 autograd, per output row of
    every parameter: err(kernel) <= 4 err(fp32) + eps (conftest)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B = 65536
    n_out = 4 * H if mode == "concurrent" else 4
    torch.manual_seed(31)
    net = _shipped_quad_net(n_out) if case == "shipped_controller" else Net(15, H, 9, n_out, conv=1)
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=77, ref_length=2 * H)
    if case == "outlier_per_workgroup":
        d["ref"] = d["ref"].clone()
        d["ref"][7::256] *= 1.0e3
    gnet = copy.deepcopy(net).to(dev)
    dyn = FlightmareDynamics()
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    import plane_path as PP
    if mode == "concurrent":
        with torch.no_grad():
            normed = state_preprocessing(s0)
        args = (gnet, normed, s0, in_ref[:, :H].contiguous(), ref[:, :H].contiguous(), DT,
                dyn.params)
        _, grads, _ = F.quad_concurrent_policy_grads(*args)
    else:
        args = (gnet, s0, in_ref, ref, DT, dyn.params)
        _, grads, _ = F.quad_mlp_rollout_grads(*args)
    got = {k: N(v) for k, v in grads.items()}
    # the same step through the plane + product sequence of rounds 1-4
    # (tests/plane_path.py): the same cotangents, exact float accumulation instead
    # of the fixed-point blocks
    _, gp, _ = (PP.quad_concurrent_policy_grads_planes if mode == "concurrent"
                else PP.quad_mlp_rollout_grads_planes)(*args)
    planes = {k: N(v) for k, v in gp.items()}
    want = _oracle(net, d, mode, torch.float64)
    f32 = _oracle(net, d, mode, torch.float32)
    assert set(want) <= set(got)
    for k, w in want.items():
        assert rel_err(got[k], w) < 1e-4, (k, rel_err(got[k], w))
    assert_param_rows_no_worse_than_fp32(planes, f32, want, f"{mode}, {case}, PLANE path",
                                         report_only=True)
    assert_param_rows_no_worse_than_fp32(got, f32, want, f"{mode}, {case}")


def test_gather_plan_never_replays_a_stale_copy(dev):
    """a planned gather (to_soa_multi_planned) froze the source
    POINTERS; where the source had to be converted first (a float64 data set)
    that pointer was a one-time copy and an in-place refresh of the data set
    was not seen.  Such sources are converted on every call now."""
    from apg_trajectory_tracking_amd import functional as F
    src64 = torch.randn(512, 12, dtype=torch.float64, device=dev)
    src32 = torch.randn(512, 15, device=dev)
    idx = torch.randperm(512, device=dev)[:128].contiguous()
    plan = None
    for rnd in range(3):
        outs, plan = F.to_soa_multi_planned([(src64, None), (src32, None)], idx, plan)
        assert torch.equal(outs[0], src64[idx].float().t())
        assert torch.equal(outs[1], src32[idx].t())
        src64.mul_(-1.5), src32.add_(1.0)       # resample_data() writes in place
    assert plan is None                          # (a converted source: never planned)
    outs, plan = F.to_soa_multi_planned([(src32, None)], idx, None)
    assert plan is not None
    src32.mul_(2.0)
    outs2, plan2 = F.to_soa_multi_planned([(src32, outs[0])], idx, plan)
    assert torch.equal(outs2[0], src32[idx].t())


def _trainer(mode, B, dev, net=None, seed=4):
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(),
                   dict(QUAD_CFG, train_mode=mode, batch_size=B))
    torch.manual_seed(seed)
    if net is not None:
        t.net = copy.deepcopy(net).to(dev)
    elif mode == "LSTM":
        t.net = LSTM_NEW(15, H, 9, 4, conv=1).to(dev)
    else:
        t.net = Net(15, H, 9, 4 * H if mode == "concurrent" else 4, conv=1).to(dev)
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=9, ref_length=t.ref_length)

    class Shard:
        states, in_ref_states, ref_states = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    with torch.no_grad():
        Shard.normed_states = state_preprocessing(Shard.states)
    t.state_data, t.static_shard = Shard, True
    t.hidden_generator = None
    if mode == "LSTM":      # the same (h0, c0) in every step of every trainer
        gen = torch.Generator().manual_seed(3)
        hc = torch.randn(2, 8, B, generator=gen).to(dev)

        def fixed_reset(batch_size=1, generator=None, net=t.net):
            net.hidden_state, net.cell_state = hc[0].t(), hc[1].t()
        t.net.reset_hidden_state = fixed_reset
    t.init_optimizer()
    if mode == "concurrent":
        step = lambda: t.train_concurrent_fused(
            Shard.normed_states, Shard.states, Shard.in_ref_states, Shard.ref_states)
    else:
        step = lambda: t.train_recurrent_model(
            None, Shard.states, Shard.in_ref_states, Shard.ref_states)
    return t, step


def _params(t):
    return [p.detach().clone() for p in t.net.parameters()]


def test_capture_warm_up_leaves_no_trace_in_adam_and_adam_is_captured(dev):
    """ADVICE r4 + VERDICT r4 missing #4: a user-supplied Adam keeps its graphs
    (capturable is switched on) and starts from a clean state - the two warm-up
    steps of the capture are undone in exp_avg / exp_avg_sq / step as well."""
    from apg_trajectory_tracking_amd import functional as F
    res = {}
    for graph in (True, False):
        F._STATIC_PLANES.entries.clear()
        t, step = _trainer("autoregressive", 1024, dev)
        t.measure_launch_form = False
        # (the eager reference is built capturable: torch's capturable Adam keeps
        # its step counter on the device and rounds the bias correction there -
        # like for like with what the graphed trainer is switched to)
        t.optimizer_controller = torch.optim.Adam(t.net.parameters(), lr=1e-4,
                                                  capturable=not graph)
        t.graph_steps = graph
        with warnings.catch_warnings():
            warnings.simplefilter("error")          # "graph_steps switched off" would raise
            losses = [float(step()) for _ in range(3)]
        st = t.optimizer_controller.state
        steps = {float(s["step"]) for s in st.values() if "step" in s}
        assert steps == {3.0}, steps
        if graph:
            assert t._graphs and t.graph_steps is True
        res[graph] = (losses, _params(t))
    assert res[True][0] == res[False][0]
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    F._STATIC_PLANES.entries.clear()


@pytest.mark.parametrize("mode", ["concurrent", "autoregressive", "LSTM"])
def test_launch_form_is_measured_and_its_steps_are_undone(dev, mode):
    """the first capture of a mode times graph replays
    against stream-order launches, records the choice, and the training those
    timing steps did is undone: parameters after n real steps equal a trainer
    that never measured."""
    from apg_trajectory_tracking_amd import functional as F
    outs = {}
    for measure in (True, False):
        F._STATIC_PLANES.entries.clear()
        t, step = _trainer(mode, 2048, dev)
        t.measure_launch_form, t.launch_form_steps = measure, 4
        t.plan_steps = False                       # (the concurrent step: a graph, not a plan)
        losses = [float(step()) for _ in range(4)]
        outs[measure] = (losses, _params(t))
        if measure:
            rec = t.results_dict["launch_form"]
            assert len(rec) == 1 and rec[0]["train_mode"] == mode
            assert rec[0]["chosen"] in ("graph", "eager")
            assert t.launch_form[mode] == rec[0]["chosen"]
            assert (rec[0]["chosen"] == "graph") == (
                rec[0]["ms_graph"] <= rec[0]["ms_stream_order"])
            assert bool(t._graphs) == (rec[0]["chosen"] == "graph")
    assert outs[True][0] == outs[False][0]
    for a, b in zip(outs[True][1], outs[False][1]):
        assert torch.equal(a, b)
    F._STATIC_PLANES.entries.clear()


@pytest.mark.timeout(300)
def test_split_graph_step_through_a_live_rccl_group_of_one(dev):
    """what one GPU can show of the N > 1 step.  A nccl
    process group of world size 1 is initialised in-process; the forced split
    step then runs graph A -> dist.all_reduce (a real RCCL call: communicator,
    RCCL's stream, the event ordering against both graphs, capture under the
    live watchdog with capture_error_mode thread_local) -> graph B.  Four steps
    of every mode equal the unsplit step bit for bit; the collective's latency
    for the three message sizes is printed (bench.py records it as well)."""
    import torch.distributed as dist
    from apg_trajectory_tracking_amd import functional as F, parallel
    assert not dist.is_initialized()
    store = dist.FileStore(os.path.join(tempfile.mkdtemp(), "rccl_world1"), 1)
    dist.init_process_group("nccl", store=store, rank=0, world_size=1)
    try:
        assert parallel.group_live() and parallel.world_size() == 1
        for mode in ("concurrent", "autoregressive", "LSTM"):
            outs = {}
            for split in (True, False):
                F._STATIC_PLANES.entries.clear()
                t, step = _trainer(mode, 4096, dev)
                t.measure_launch_form = False
                t.split_graph = True if split else None
                losses = [float(step()) for _ in range(4)]
                if split:
                    assert t._graphs and all(g.split for g in t._graphs.values()), mode
                outs[split] = (losses, _params(t))
            assert outs[True][0] == outs[False][0], mode
            for a, b in zip(outs[True][1], outs[False][1]):
                assert torch.equal(a, b), mode
        lat = {}
        for n in (32729, 30389, 12341):            # concurrent / AR / LSTM messages (+ loss slot)
            buf = torch.zeros(n, device=dev)
            for _ in range(5):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            lat[n] = e0.elapsed_time(e1) / 50 * 1e3
        print("allreduce_us_world1:", {k: round(v, 1) for k, v in lat.items()})
    finally:
        dist.destroy_process_group()
        F._STATIC_PLANES.entries.clear()


# ---------------------------------------------------------------- item 5
def _g17(dev):
    """(golden G17, shipped controller, LSTM controller + its start state, the
    package's LearntDynamics carrying the fixture's fitted simulator)."""
    import ast
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import LearntDynamics
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    g, g11, ck = (load_golden("closed_loop_learnt.npz"), load_golden("closed_loop.npz"),
                  load_golden("checkpoints.npz"))
    net = build_policy("quad", {k[len("quad.w."):]: torch.from_numpy(ck[k])
                                for k in ck.files if k.startswith("quad.w.")}).to(dev)
    lstm = LSTM_NEW(15, 10, 9, 4, conv=1)
    lstm.load_state_dict({k[len("lstm.w."):]: torch.from_numpy(g11[k])
                          for k in g11.files if k.startswith("lstm.w.")})
    init = {kv.split("=")[0]: ast.literal_eval(kv.split("=")[1]) for kv in g["init"]}
    dyn = LearntDynamics(initial_params=init)
    dyn.load_state_dict({k[len("dyn."):]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith("dyn.")})
    hidden = (torch.from_numpy(g11["lstm.h0"]).to(dev), torch.from_numpy(g11["lstm.c0"]).to(dev))
    return g, net, lstm.to(dev), hidden, dyn.to(dev)


@pytest.mark.parametrize("case", ["train", "test", "tight", "lstm_train", "lstm_test"])
def test_closed_loop_through_the_learnt_simulator_matches_reference_evaluator(dev, case):
    """G17 / N2 x N3 (VERDICT r4 missing #2): the closed-loop kernels stepping
    through LearntDynamics - action transform, Flightmare step with the
    construction-time kinv / inertia, residual network - against the REAL
    QuadEvaluator.follow_trajectory over QuadRotorEnvBase(LearntDynamics)
    (scripts/train_drone.py:44-45, 205-238), shipped controller and LSTM;
    through QuadEvaluator, as TrainDrone.evaluate_model calls it."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.evaluate_drone import QuadEvaluator
    g, net, lstm, hidden, dyn = _g17(dev)
    traj = torch.from_numpy(g["trajs"]).to(dev).clone()
    traj[:, :, 2] += 3          # Random.__init__, random_traj.py:34
    kw = dict(max_steps=int(g["max_steps"]), thresh_div=float(g[f"{case}.thresh_div"]),
              thresh_stable=float(g[f"{case}.thresh_stable"]),
              test_time=int(g[f"{case}.test_time"]), want_trajectory=True)
    if case.startswith("lstm"):
        out = F.quad_lstm_closed_loop(lstm, traj, float(g["dt"]), dyn.params, *hidden,
                                      learnt=dyn, **kw)
    else:
        out = F.quad_mlp_closed_loop(net, traj, float(g["dt"]), dyn.params, learnt=dyn, **kw)
    for i in range(traj.shape[0]):
        n = len(g[f"{case}.{i}.div"])
        assert int(out["steps"][i]) == n, (case, i)
        assert rel_err(N(out["drone"][:n + 1, :, i]), g[f"{case}.{i}.drone"]) < 1e-4
        assert np.abs(N(out["div"][:n, i]) - g[f"{case}.{i}.div"]).max() < 2e-4
        want = g[f"{case}.{i}.actions"]
        want = want[:, 0] if want.ndim == 3 else want
        assert rel_err(N(out["actions"][:n, :, i]), want) < 1e-4
    # the evaluator object hands the learnt simulator on by itself
    ctrl = lstm if case.startswith("lstm") else net
    ev = QuadEvaluator(ctrl, dyn, ref_length=10, dt=float(g["dt"]),
                       test_time=kw["test_time"])
    assert ev.learnt is dyn
    if case.startswith("lstm"):
        ev.hidden = hidden
    _, drone, divs, _ = ev.follow_trajectory(
        "rand", max_nr_steps=kw["max_steps"], thresh_stable=kw["thresh_stable"],
        thresh_div=kw["thresh_div"], trajectories=traj)
    for i in range(traj.shape[0]):
        assert len(divs[i]) == len(g[f"{case}.{i}.div"])
        assert np.abs(N(divs[i]) - g[f"{case}.{i}.div"]).max() < 2e-4
    # the residual and the action transform matter in this fixture
    if case == "train":
        plain = F.quad_mlp_closed_loop(net, traj, float(g["dt"]), dyn.params, **kw)
        n = len(g["train.0.div"])
        assert rel_err(N(plain["drone"][:n + 1, :, 0]), g["train.0.drone"]) > 1e-2


def test_closed_loop_learnt_simulator_large_batch_vs_oracle(dev):
    """The learnt-simulator closed loop over several workgroups with a ragged
    tail, MLP and LSTM controller, against the batched oracle loop."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from oracle import torch_port as tp
    g, net, lstm, _, dyn = _g17(dev)
    weights = {k[len("dyn."):]: g[k] for k in g.files if k.startswith("dyn.")}
    import ast
    init = {kv.split("=")[0]: ast.literal_eval(kv.split("=")[1]) for kv in g["init"]}
    B, L, steps = 300, 40, 30
    traj = synthetic.quad_eval_trajectories(B, L, 0.1, seed=9)
    lifted = traj.clone()
    lifted[:, :, 2] += 3
    gen = torch.Generator().manual_seed(4)
    h0, c0 = torch.randn(B, 8, generator=gen), torch.randn(B, 8, generator=gen)
    for ctrl, name in ((net, "mlp"), (lstm, "lstm")):
        cpu = __import__("copy").deepcopy(ctrl).cpu()
        if name == "lstm":
            cpu.hidden_state, cpu.cell_state = h0.clone(), c0.clone()
            out = F.quad_lstm_closed_loop(ctrl, lifted.to(dev), 0.1, dyn.params, h0.to(dev),
                                          c0.to(dev), max_steps=steps, thresh_div=0.4,
                                          learnt=dyn)
        else:
            out = F.quad_mlp_closed_loop(ctrl, lifted.to(dev), 0.1, dyn.params,
                                         max_steps=steps, thresh_div=0.4, learnt=dyn)
        ref = tp.quad_closed_loop(cpu, tp.LearntQuadOracle(weights, init), traj, 0.1, 10,
                                  steps, 0.4, 1.0, 0)
        same = [i for i in range(B) if int(out["steps"][i]) == int(ref["steps"][i])]
        assert len(same) > 0.97 * B, name
        bad = [i for i in same
               if np.abs(N(out["div"][:int(ref["steps"][i]), i])
                         - ref["div"][i, :int(ref["steps"][i])].numpy()).max() > 2e-3]
        assert len(bad) <= 0.03 * B, (name, len(bad))


def test_evaluate_model_flies_the_learnt_simulator_without_a_substitution(dev):
    """TrainDrone.evaluate_model with sample_in = "train_env" and a learnt
    training simulator (the reference's train_dynamics() flow): no warning, and
    the statistics are those of the learnt simulator's closed loop - not those
    of the analytic evaluation dynamics rounds 3 / 4 substituted."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.evaluate_drone import QuadEvaluator
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g, net, _, _, dyn = _g17(dev)
    cfg = dict(QUAD_CFG, batch_size=32, epoch_size=64, self_play=0, nr_test=64,
               sample_in="train_env", max_steps=40, thresh_div_start=0.3,
               thresh_div_end=2.0, thresh_stable_start=1.0)
    t = TrainDrone(dyn, FlightmareDynamics(), cfg)
    t.initialize_model(device=dev, seed=2)
    t.net = net
    t.save_path = "/tmp/apg_eval_test_learnt"
    torch.manual_seed(12)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)    # (the substitution's category)
        res = t.evaluate_model(0)
    assert res is not None
    stats = {}
    for name, env in (("learnt", dyn), ("analytic", FlightmareDynamics())):
        torch.manual_seed(12)
        stats[name] = QuadEvaluator(net, env, ref_length=10, dt=0.1).run_eval(
            "rand", nr_test=64, max_steps=40, thresh_div=0.3, thresh_stable=1.0)
    assert t.results_dict["mean_divergence"][-1] == pytest.approx(stats["learnt"][4])
    assert t.results_dict["mean_success"][-1] == pytest.approx(stats["learnt"][0])
    assert abs(stats["analytic"][4] - stats["learnt"][4]) > 1e-3 * abs(stats["learnt"][4])
    assert "evaluation_env" not in t.results_dict or not t.results_dict["evaluation_env"]


# ------------------------------------------------------------- ADVICE r4, lows
def test_step_hooks_keep_the_optimizer_in_charge_and_schedulers_see_steps(dev):
    """With the in-kernel update optimizer.step() is never called: a user's
    step hooks would be skipped.  With hooks registered the update is left to
    the optimizer (the hook runs once per step, the result is the same); without
    hooks an LR scheduler does not warn about a step it did not see."""
    t, step = _trainer("concurrent", 1024, dev)
    assert t._in_kernel_update(True) is not None
    sched = torch.optim.lr_scheduler.StepLR(t.optimizer_controller, step_size=100)
    step()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sched.step()                   # "scheduler.step() before optimizer.step()"
    plain = [float(step()) for _ in range(2)]
    h, hstep = _trainer("concurrent", 1024, dev)
    calls = []
    h.optimizer_controller.register_step_post_hook(lambda opt, a, k: calls.append(1))
    assert h._in_kernel_update(True) is None
    hooked = [float(hstep()) for _ in range(3)]
    assert len(calls) == 3
    assert hooked[1:] == pytest.approx(plain, rel=1e-6)


def test_data_set_written_in_place_is_checked_again_before_replays(dev):
    """The operand-range contract of the in-kernel policies (finite, below 2^14)
    on paths that replay: an epoch trains from plans / graphs, then the data
    set is written in place (as resample_data / add_eval_data do) - the next
    run_epoch raises instead of training on inf."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    for mode in ("concurrent", "autoregressive"):
        cfg = dict(QUAD_CFG, train_mode=mode, batch_size=96, epoch_size=300, self_play=0)
        t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
        t.initialize_model(device=dev, seed=3)
        assert np.isfinite(t.run_epoch("controller", 0))
        assert np.isfinite(t.run_epoch("controller", 1))
        with torch.no_grad():
            t.state_data.in_ref_states[7, 3, 2] = float("inf")
        with pytest.raises(ValueError, match="in_ref_states"):
            t.run_epoch("controller", 2)
        with torch.no_grad():
            t.state_data.in_ref_states[7, 3, 2] = 0.5
            t.state_data.states[11, 0] = 3e4
        with pytest.raises(ValueError, match="states"):
            t.run_epoch("controller", 3)


def test_linear_weight_gradient_with_mixed_dtypes_under_autocast(dev):
    """init_optimizer switches a user policy's torch.nn.Linear layers to the
    library's weight gradient; under autocast grad_out is half and x float:
    the fallback multiplies in the wider type, as stock nn.Linear does."""
    from apg_trajectory_tracking_amd import nn as apg_nn
    torch.manual_seed(0)
    lin, ref = apg_nn.Linear(24, 16).to(dev), torch.nn.Linear(24, 16).to(dev)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(64, 24, device=dev)
    outs = []
    for m in (lin, ref):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.float().pow(2).sum().backward()
        outs.append((m.weight.grad, m.bias.grad, xi.grad))
    for a, b in zip(*outs):
        assert a.dtype == b.dtype and rel_err(N(a), N(b)) < 1e-2


# ------------------------------------------------- item 4: the gather, folded
@pytest.mark.parametrize("B,rows_per_ref", [(65536, 10), (1000, 20), (31, 10)])
def test_rows_read_through_the_index_equal_the_gathered_step(dev, B, rows_per_ref):
    """apg_quad_mlp_concurrent_train_step_rows - the forward
    kernel reads its trajectories' rows of the data set's tensors through the
    index (direct-to-LDS loads) - against the same step on planes gathered by
    apg_to_soa_multi: loss, every gradient and the parameters after three
    in-kernel SGD steps, bit for bit (the arithmetic is the same; only where the
    inputs come from differs).  Ragged batch, repeated rows, 20-row windows."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    N_ = 3 * B + 17
    d = synthetic.quad_polynomial_batch(N_, H, DT, seed=5, ref_length=rows_per_ref)
    st, inr, rf = (d[k].to(dev).contiguous() for k in ("state0", "in_ref", "ref"))
    with torch.no_grad():
        normed = state_preprocessing(st).contiguous()
    gen = torch.Generator().manual_seed(B)
    index = torch.randint(0, N_, (B,), generator=gen).to(dev)
    index[B // 2] = index[0]                                  # a row used twice
    params = FlightmareDynamics().params
    results = []
    for rows in (False, True):
        torch.manual_seed(1)
        net = Net(15, H, 9, 4 * H, conv=1).to(dev)
        names = F._MLP_PARAMS
        bufs = {n: torch.zeros_like(p) for n, p in net.named_parameters() if n in names}
        update = (1e-6, 0.9, bufs)
        if rows:
            plan = F.QuadConcurrentStepPlan(net, None, DT, params, update=update,
                                            rows=(normed, st, inr, rf, B))
            # (nobody writes the feature / window planes in this mode, and a ragged
            # workgroup's dead lanes read past the last activation plane: whatever
            # the buffer held must not reach a gradient)
            plan._keep["prepared"][0].fill_(float("nan"))
            step = lambda: plan.launch(index=index)
        else:
            prepared = F.quad_concurrent_prepare(normed, st, inr, rf, index=index)
            plan = F.QuadConcurrentStepPlan(net, prepared, DT, params, update=update)
            step = lambda: plan.launch()
        losses = [float(step()) for _ in range(3)]
        results.append((losses, plan.flat.clone(), [p.detach().clone() for p in net.parameters()]))
    (l0, g0, p0), (l1, g1, p1) = results
    assert l0 == l1 and np.isfinite(l0).all()
    assert torch.equal(g0[:-1], g1[:-1])     # (the last slot is the all-reduce's loss)
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)


def test_rows_step_argument_checks(dev):
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    d = synthetic.quad_polynomial_batch(64, H, DT, seed=5, ref_length=10)
    st, inr, rf = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    normed = state_preprocessing(st)
    net = Net(15, H, 9, 4 * H, conv=1).to(dev)
    params = FlightmareDynamics().params
    plan = F.QuadConcurrentStepPlan(net, None, DT, params, rows=(normed, st, inr, rf, 32))
    with pytest.raises(ValueError):
        plan.launch()                                             # no index
    with pytest.raises(ValueError):
        plan.launch(index=torch.arange(32, device=dev, dtype=torch.int32))
    with pytest.raises(ValueError):
        plan.launch(index=torch.arange(31, device=dev))
    with pytest.raises(ValueError):
        F.QuadConcurrentStepPlan(net, None, DT, params,
                                 rows=(normed.double(), st, inr, rf, 32))
    assert np.isfinite(float(plan.launch(index=torch.arange(32, device=dev))))


def test_run_epoch_concurrent_names_its_batches_by_rows(dev):
    """run_epoch's concurrent loop takes the rows path (no to_soa gather) and
    trains exactly as the loop over gathered batches does."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    outs = []
    for rows in (True, False):
        cfg = dict(QUAD_CFG, batch_size=96, epoch_size=300, self_play=0)
        t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
        t.rows_in_kernel = rows
        t.shuffle = False
        t.initialize_model(device=dev, seed=3)
        if outs:
            t.net.load_state_dict(first)
        else:
            first = copy.deepcopy(t.net.state_dict())
        losses = [t.run_epoch("controller", e) for e in range(3)]
        assert ("rows" in t.last_epoch_loop) == rows
        outs.append((losses, _params(t)))
    assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-6)   # (sums in another order)
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("graph", [False, True])
def test_autoregressive_step_applies_the_update_inside_its_second_stage(dev, graph):
    """The autoregressive trainer step with momentum SGD applied by the reverse
    sweep's second stage (one process) == the same step followed by
    optimizer.step() (torch's fused SGD: the same arithmetic): losses, gradients,
    parameters and momentum buffers of four steps, bit for bit."""
    from apg_trajectory_tracking_amd import functional as F
    outs = []
    for in_kernel in (True, False):
        F._STATIC_PLANES.entries.clear()
        t, step = _trainer("autoregressive", 1500, dev)
        t.graph_steps, t.in_kernel_update = graph, in_kernel
        t.measure_launch_form = False
        losses = [float(step()) for _ in range(4)]
        assert (t._in_kernel_update(True) is not None) == in_kernel
        bufs = [t.optimizer_controller.state[p]["momentum_buffer"].clone()
                for p in t.net.parameters() if p.grad is not None]
        outs.append((losses, _params(t), bufs,
                     [p.grad.clone() for p in t.net.parameters() if p.grad is not None]))
    (la, pa, ba, ga), (lb, pb, bb, gb) = outs
    assert la == lb and la[3] < la[0]
    for xs, ys in ((pa, pb), (ba, bb), (ga, gb)):
        assert len(xs) == len(ys) and all(torch.equal(x, y) for x, y in zip(xs, ys))
    F._STATIC_PLANES.entries.clear()


# ------------------------------------------- resident operand tables (item 3)
@pytest.mark.parametrize("B", [65536, 700])
def test_resident_tables_equal_repacked_tables(dev, B):
    """The step plan keeps its packed operand tables current through the second
    stage (ApgMlpSgdUpdate.resident: no pack launch from the second step on).
    Against the plan that packs at the head of every step: losses, gradients,
    parameters, momentum buffers of six steps bit for bit; the resident tables
    themselves equal a fresh pack of the final parameters; a parameter written
    from outside (in-place version counter) makes the next step pack again."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=6, ref_length=H)
    st, inr, rf = (d[k].to(dev).contiguous() for k in ("state0", "in_ref", "ref"))
    normed = state_preprocessing(st).contiguous()
    params = FlightmareDynamics().params
    prepared = F.quad_concurrent_prepare(normed, st, inr, rf)
    n_tab = None
    outs = []
    for resident in (True, False):
        torch.manual_seed(1)
        net = Net(15, H, 9, 4 * H, conv=1).to(dev)
        bufs = {n: torch.zeros_like(p) for n, p in net.named_parameters() if n in F._MLP_PARAMS}
        plan = F.QuadConcurrentStepPlan(net, prepared, DT, params,
                                        update=(2e-4 / B, 0.9, bufs))
        plan.resident_tables = resident
        flags, losses = [], []
        for i in range(6):
            if i == 4:          # somebody else writes a parameter
                with torch.no_grad():
                    net.fc2.bias.mul_(1.5)
            losses.append(float(plan.launch()))
            flags.append(plan._keep["upd"].resident)
        assert flags == ([1, 2, 2, 2, 3, 2] if resident else [0] * 6)
        outs.append((losses, plan.flat[:-1].clone(), [p.detach().clone() for p in net.parameters()],
                     [b.clone() for b in bufs.values()], plan, net))
    (la, ga, pa, ba, plan_a, net_a), (lb, gb, pb, bb, _, _) = outs
    assert la == lb and np.isfinite(la).all() and la[3] != la[0]
    assert torch.equal(ga, gb)
    for xs, ys in ((pa, pb), (ba, bb)):
        assert all(torch.equal(x, y) for x, y in zip(xs, ys))
    # the tables the second stage left == the tables a pack of these parameters gives
    zero = {n: torch.zeros_like(p) for n, p in net_a.named_parameters() if n in F._MLP_PARAMS}
    fresh = F.QuadConcurrentStepPlan(net_a, prepared, DT, params, update=(0.0, 0.0, zero))
    fresh.resident_tables = False
    fresh.launch()             # (lr = 0: packs, leaves the parameters alone)
    n_tab = (plan_a._keep["ws"].numel() - 4 * 31 * 1024 - 4)
    ta, tb = (p._keep["ws"][:n_tab].view(torch.int32).clone() for p in (plan_a, fresh))
    for lo, hi in ((352, 384), (644, 1024)):    # gaps of the float tables: never written
        ta[lo:hi] = tb[lo:hi] = 0
    assert torch.equal(ta, tb)


def test_rows_epochs_with_shuffle_are_reproducible_and_draw_ahead(dev):
    """The rows epoch loop draws the NEXT epoch's permutation on a side stream
    while the current one runs: same seeds -> the same epochs, bit for bit; the
    order drawn ahead is the one the next epoch uses; a changed data-set size
    drops it."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    runs = []
    for _ in range(2):
        cfg = dict(QUAD_CFG, batch_size=128, epoch_size=1000, self_play=0)
        t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
        t.initialize_model(device=dev, seed=3)
        if runs:
            t.net.load_state_dict(first)
        else:
            first = copy.deepcopy(t.net.state_dict())
        torch.manual_seed(21)
        torch.cuda.manual_seed(22)
        losses = []
        for e in range(3):
            ahead = getattr(t.trainloader, "_order_ahead", None)
            losses.append(t.run_epoch("controller", e))
            assert "rows" in t.last_epoch_loop
            if e:
                assert ahead is not None and sorted(ahead[0].tolist()) == list(range(1000))
        runs.append((losses, _params(t)))
        assert t.trainloader._order_ahead[0].numel() == 1000
    assert runs[0][0] == runs[1][0] and len(set(runs[0][0])) == 3
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    # an order drawn for another data-set size is not used
    ld = t.trainloader
    ld._order_ahead = (torch.arange(7, device=dev), torch.cuda.Event())
    ld._order_ahead[1].record()
    assert ld.epoch_order().numel() == 1000 and not hasattr(ld, "_order_ahead")
