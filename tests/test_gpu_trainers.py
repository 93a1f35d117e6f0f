"""Trainer-level parity on the GPU: the drop-in TrainDrone / TrainFixedWing /
TrainCartpole against the golden train steps recorded from the reference
trainer (tests/golden/make_golden.py G3, G4, G6) and against the oracle."""
import os
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

QUAD_CFG = dict(
    delta_t=0.1, delta_t_train=0.1, epoch_size=1000, self_play=1,
    batch_size=64, state_size=12, horizon=10, train_mode="concurrent",
    ref_dim=9, action_dim=4, learning_rate_controller=1e-5, system="quad",
    modified_params={},
)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def D(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def load_weights(net, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files
          if k.startswith(prefix)}
    net.load_state_dict(sd)


def make_trainer(cls, dyn, cfg):
    cfg = dict(cfg)
    return cls(dyn, dyn, cfg)


def test_quad_train_controller_two_sgd_steps(dev):
    """G3: scripts/train_drone.py:175-203 through run_epoch's body, twice
    (second step exercises the SGD momentum buffer)."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g = load_golden("quad_train.npz")
    trainer = make_trainer(TrainDrone, FlightmareDynamics(), QUAD_CFG)
    net = Net(15, 10, 9, 40, conv=1)
    load_weights(net, g, "w0.")
    trainer.net = net.to(dev)
    trainer.optimizer_controller = torch.optim.SGD(
        trainer.net.parameters(), lr=float(g["lr"]), momentum=float(g["momentum"]))
    in_state, in_ref = D(g["in_state"], dev), D(g["in_ref"], dev)
    state0, ref = D(g["state0"], dev), D(g["ref"], dev)
    for step in (1, 2):
        actions = torch.sigmoid(trainer.net(in_state, in_ref))
        action_seq = torch.reshape(actions, (-1, 10, 4))
        loss = trainer.train_controller_model(state0, action_seq, in_ref, ref)
        assert abs(loss.item() - g[f"loss{step}"]) / g[f"loss{step}"] < 1e-5
        if step == 1:
            assert rel_err(N(action_seq), g["actions1"]) < 1e-5
            for k, p in trainer.net.named_parameters():
                if "g1." + k in g.files:
                    assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k
        for k, v in trainer.net.state_dict().items():
            assert rel_err(N(v), g[f"w{step}.{k}"]) < 1e-5, (step, k)


def test_quad_concurrent_fused_policy_two_sgd_steps(dev):
    """G3 again, with policy forward, rollout, loss, adjoint and policy
    backward inside the fused kernels (apg_quad_mlp_concurrent_fwd_bwd):
    loss, every parameter gradient, post-SGD weights of two steps."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g = load_golden("quad_train.npz")
    trainer = make_trainer(TrainDrone, FlightmareDynamics(), QUAD_CFG)
    net = Net(15, 10, 9, 40, conv=1)
    load_weights(net, g, "w0.")
    trainer.net = net.to(dev)
    trainer.optimizer_controller = torch.optim.SGD(
        trainer.net.parameters(), lr=float(g["lr"]), momentum=float(g["momentum"]))
    in_state, in_ref = D(g["in_state"], dev), D(g["in_ref"], dev)
    state0, ref = D(g["state0"], dev), D(g["ref"], dev)
    for step in (1, 2):
        loss = trainer.train_concurrent_fused(in_state, state0, in_ref, ref)
        assert loss is not None
        assert abs(loss.item() - g[f"loss{step}"]) / g[f"loss{step}"] < 1e-5
        if step == 1:
            for k, p in trainer.net.named_parameters():
                if "g1." + k in g.files:
                    assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k
        for k, v in trainer.net.state_dict().items():
            assert rel_err(N(v), g[f"w{step}.{k}"]) < 1e-5, (step, k)


@pytest.mark.parametrize("B", [1, 77, 600])
def test_quad_concurrent_fused_matches_unfused(dev, B):
    """Ragged batches: fused concurrent step == torch policy + fused rollout."""
    import copy
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=90 + B)
    state0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    with torch.no_grad():
        normed = state_preprocessing(state0)
    torch.manual_seed(4)
    base = Net(15, 10, 9, 40, conv=1)
    res = []
    for fused in (False, True):
        trainer = TrainDrone(FlightmareDynamics(), FlightmareDynamics(),
                             dict(QUAD_CFG, batch_size=B))
        trainer.net = copy.deepcopy(base).to(dev)
        trainer.optimizer_controller = torch.optim.SGD(trainer.net.parameters(), lr=0.0)
        if fused:
            loss = trainer.train_concurrent_fused(normed, state0, in_ref, ref)
        else:
            acts = torch.sigmoid(trainer.net(normed, in_ref)).reshape(-1, 10, 4)
            loss = trainer.train_controller_model(state0, acts, in_ref, ref)
        res.append((loss.item(), {k: N(p.grad) for k, p in
                                  trainer.net.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) / abs(l0) < 1e-5
    assert set(g0) == set(g1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-4, k


def test_quad_soa_head_matches_aos_path(dev):
    """Net.forward_soa + layout='soa' rollout == the AoS trainer path."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    g = load_golden("quad_train.npz")
    dyn = FlightmareDynamics()
    net = Net(15, 10, 9, 40, conv=1)
    load_weights(net, g, "w0.")
    net.to(dev)
    in_state, in_ref = D(g["in_state"], dev), D(g["in_ref"], dev)
    state0, ref = D(g["state0"], dev), D(g["ref"], dev)
    acts = torch.sigmoid(net.forward_soa(in_state, in_ref)).reshape(10, 4, -1)
    loss = F.quad_rollout_loss(
        state0.t().contiguous(), acts, ref.permute(1, 2, 0).contiguous(),
        0.1, dyn.params, layout="soa")
    loss.backward()
    assert abs(loss.item() - g["loss1"]) / g["loss1"] < 1e-5
    for k, p in net.named_parameters():
        if "g1." + k in g.files:
            assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k


@pytest.mark.parametrize("mode", ["ar", "lstm"])
def test_quad_recurrent_unroll(dev, mode):
    """G4: autoregressive / LSTM unroll with the pinned window semantics."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g = load_golden("quad_recurrent.npz")
    cfg = dict(QUAD_CFG, batch_size=32,
               train_mode="LSTM" if mode == "lstm" else "autoregressive")
    trainer = make_trainer(TrainDrone, FlightmareDynamics(), cfg)
    assert trainer.ref_length == 20 and trainer.actions_out_dim == 4
    net = (LSTM_NEW if mode == "lstm" else Net)(15, 10, 9, 4, conv=1)
    load_weights(net, g, f"{mode}.w.")
    trainer.net = net.to(dev)
    trainer.optimizer_controller = torch.optim.SGD(
        trainer.net.parameters(), lr=0.0, momentum=0.9)
    state0, in_ref, ref = (D(g["state0"], dev), D(g["in_ref"], dev),
                           D(g["ref"], dev))
    if mode == "lstm":
        h0, c0 = D(g["lstm_h0"], dev), D(g["lstm_c0"], dev)

        def fixed_reset(batch_size=1, generator=None):
            net.hidden_state, net.cell_state = h0.clone(), c0.clone()
        net.reset_hidden_state = fixed_reset
    in_ref_before = in_ref.clone()
    for fused in (False, True):
        trainer.fused_policy = fused   # K7/K8 (policy in-kernel) vs per-step kernels
        loss = trainer.train_recurrent_model(None, state0, in_ref, ref)
        assert torch.equal(in_ref, in_ref_before)   # the window is copied
        assert abs(loss.item() - g[f"{mode}.loss"]) / g[f"{mode}.loss"] < 2e-5
        for k, p in trainer.net.named_parameters():
            key = f"{mode}.g.{k}"
            if key in g.files:
                assert rel_err(N(p.grad), g[key]) < 1e-4, (fused, k)


def test_cartpole_train_step(dev):
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import (
        CartpoleDynamics)
    from apg_trajectory_tracking_amd.models.simple_model import Net
    from apg_trajectory_tracking_amd.train_cartpole import TrainCartpole
    g = load_golden("cartpole.npz")
    cfg = dict(delta_t=float(g["dt"]), batch_size=64, state_size=4, horizon=5,
               action_dim=1, ref_dim=4, train_mode="concurrent",
               learning_rate_controller=1e-4, system="cartpole")
    trainer = TrainCartpole(CartpoleDynamics(), CartpoleDynamics(), cfg)
    net = Net(4, 5)
    load_weights(net, g, "w0.")

    class OneBatch:
        states = D(g["state0"], dev)
        labels = D(g["state0"], dev)
        num_sampled_states = 64
    trainer.initialize_model(base_model=net, state_data=OneBatch, device=dev)
    trainer.shuffle = False
    trainer.init_optimizer()
    # the reference makes the same reference trajectory
    ref = trainer.make_reference(OneBatch.labels)
    assert rel_err(N(ref), g["ref"]) < 1e-7
    with pytest.raises(ZeroDivisionError):   # single batch: running_loss / 0
        trainer.run_epoch("controller")
    assert torch.equal(OneBatch.states, OneBatch.labels)  # dataset untouched
    for k, p in trainer.net.named_parameters():
        assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k
    for k, v in trainer.net.state_dict().items():
        assert rel_err(N(v), g["w1." + k]) < 1e-5, k


def test_wing_train_step_vs_oracle(dev):
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    from oracle import torch_port as tp
    import copy
    cfg = dict(delta_t=0.05, delta_t_train=0.05, epoch_size=256, self_play=0,
               batch_size=256, state_size=12, horizon=20, ref_dim=3,
               action_dim=4, train_mode="concurrent",
               learning_rate_controller=1e-6, system="wing")
    trainer = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), cfg)
    torch.manual_seed(3)
    trainer.initialize_model(device=dev, seed=7)
    ref_net = copy.deepcopy(trainer.net).cpu()
    d = trainer.state_data
    assert isinstance(d, SyntheticWingDataset) and d.ref_states.shape == (256, 20, 3)
    actions = torch.sigmoid(trainer.net(d.normed_states, d.in_ref_states))
    loss = trainer.train_controller_model(
        d.states, actions.reshape(-1, 20, 4), d.in_ref_states, d.ref_states)
    # oracle: same step with CPU autograd
    a = torch.sigmoid(ref_net(d.normed_states.cpu(), d.in_ref_states.cpu()))
    inter = tp.unroll(tp.WingOracle(), d.states.cpu(), a.reshape(-1, 20, 4), 0.05)
    ref_loss = tp.fixed_wing_mpc_loss(inter, d.ref_states.cpu(), a.reshape(-1, 20, 4))
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) / ref_loss.item() < 1e-4
    for (k, p), (_, q) in zip(trainer.net.named_parameters(),
                              ref_net.named_parameters()):
        if q.grad is not None:
            assert rel_err(N(p.grad), q.grad.numpy()) < 1e-4, k


@pytest.mark.parametrize("mode", ["concurrent", "autoregressive", "LSTM"])
def test_quad_run_epoch_learns(dev, mode):
    """A few epochs of the real run_epoch on the synthetic set: finite losses
    that go down (sanity of the whole loop incl. loader, SGD, resampling)."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(QUAD_CFG, epoch_size=512, self_play=0, batch_size=128,
               train_mode=mode, learning_rate_controller=2e-6,
               resample_every=100)
    torch.manual_seed(0)
    trainer = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
    trainer.initialize_model(device=dev, seed=1)
    trainer.hidden_generator = torch.Generator().manual_seed(5)
    losses = [trainer.run_epoch("controller", epoch=e) for e in range(4)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    assert trainer.results_dict["loss"][1:] == losses


def test_fused_lstm_rollout_matches_reference_unroll(dev):
    """K7: policy-in-kernel LSTM unroll vs the golden LSTM unroll (G4):
    states, actions, loss and every parameter gradient."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    g = load_golden("quad_recurrent.npz")
    net = LSTM_NEW(15, 10, 9, 4, conv=1)
    load_weights(net, g, "lstm.w.")
    net.to(dev)
    dyn = FlightmareDynamics()
    state0, in_ref, ref = (D(g["state0"], dev), D(g["in_ref"], dev),
                           D(g["ref"], dev))
    h0, c0 = D(g["lstm_h0"], dev), D(g["lstm_c0"], dev)
    loss, states, actions = F.quad_lstm_rollout_loss(
        net, state0, in_ref, ref, float(g["dt"]), dyn.params, h0, c0)
    loss.backward()
    assert rel_err(N(states.permute(2, 0, 1)), g["lstm.states"]) < 2e-5
    assert rel_err(N(actions.permute(2, 0, 1)), g["lstm.actions"]) < 2e-5
    assert abs(loss.item() - g["lstm.loss"]) / g["lstm.loss"] < 2e-5
    for k, p in net.named_parameters():
        key = f"lstm.g.{k}"
        if key in g.files:
            assert p.grad is not None, k
            assert rel_err(N(p.grad), g[key]) < 1e-4, k


def test_fused_mlp_rollout_matches_reference_unroll(dev):
    """K8: policy-in-kernel autoregressive unroll (MFMA) vs the golden
    autoregressive unroll (G4): states, actions, loss, parameter gradients."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    g = load_golden("quad_recurrent.npz")
    net = Net(15, 10, 9, 4, conv=1)
    load_weights(net, g, "ar.w.")
    net.to(dev)
    dyn = FlightmareDynamics()
    state0, in_ref, ref = (D(g["state0"], dev), D(g["in_ref"], dev),
                           D(g["ref"], dev))
    loss, states, actions = F.quad_mlp_rollout_loss(
        net, state0, in_ref, ref, float(g["dt"]), dyn.params)
    loss.backward()
    assert rel_err(N(actions.permute(2, 0, 1)), g["ar.actions"]) < 2e-5
    assert rel_err(N(states.permute(2, 0, 1)), g["ar.states"]) < 2e-5
    assert abs(loss.item() - g["ar.loss"]) / g["ar.loss"] < 2e-5
    for k, p in net.named_parameters():
        key = f"ar.g.{k}"
        if key in g.files:
            assert p.grad is not None, k
            assert rel_err(N(p.grad), g[key]) < 1e-4, k


@pytest.mark.parametrize("B", [1, 100, 300])
def test_fused_mlp_ragged_batches_match_unfused(dev, B):
    """K8 on batch sizes that do not fill a wave / workgroup."""
    import copy
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(QUAD_CFG, batch_size=B, train_mode="autoregressive")
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=70 + B, ref_length=20)
    state0, in_ref, ref = (d["state0"].to(dev), d["in_ref"].to(dev),
                           d["ref"].to(dev))
    torch.manual_seed(9)
    base = Net(15, 10, 9, 4, conv=1)
    results = []
    for fused in (False, True):
        trainer = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
        net = copy.deepcopy(base).to(dev)
        trainer.net = net
        trainer.fused_policy = fused
        trainer.optimizer_controller = torch.optim.SGD(net.parameters(), lr=0.0)
        loss = trainer.train_recurrent_model(None, state0, in_ref, ref)
        results.append((loss.item(), {k: N(p.grad) for k, p in
                                      net.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = results
    assert abs(l0 - l1) / abs(l0) < 1e-5
    assert set(g0) == set(g1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-4, k


@pytest.mark.parametrize("mode", ["lstm", "mlp"])
def test_fused_policy_input_gradients(dev, mode):
    """dL/dstate0 (and dL/dh0, dL/dc0 for the LSTM) of the in-kernel policies
    against autograd through the per-step path - checks the carried adjoints
    (lambda, dh, dc) of the reverse sweeps end to end."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.drone_loss import quad_mpc_loss
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    B, H = 77, 10
    d = synthetic.quad_polynomial_batch(B, H, 0.1, seed=11, ref_length=20)
    in_ref, ref = d["in_ref"].to(dev), d["ref"].to(dev)
    torch.manual_seed(3)
    net = (LSTM_NEW if mode == "lstm" else Net)(15, 10, 9, 4, conv=1).to(dev)
    dyn = FlightmareDynamics()
    gen = torch.Generator().manual_seed(8)
    h0, c0 = (torch.randn(B, 8, generator=gen).to(dev) for _ in range(2))

    def leaves():
        return [t.clone().requires_grad_(True) for t in
                ((d["state0"].to(dev), h0, c0) if mode == "lstm"
                 else (d["state0"].to(dev),))]

    # per-step path
    ins = leaves()
    cur = ins[0]
    if mode == "lstm":
        net.hidden_state, net.cell_state = ins[1], ins[2]
    sts, acs = [], []
    for k in range(H):
        rel = in_ref[:, k:k + H].clone()
        rel[:, :, :3] = rel[:, :, :3] - cur[:, None, :3]
        a = torch.sigmoid(net(state_preprocessing(cur), rel))
        cur = dyn(cur, a, dt=0.1)
        sts.append(cur), acs.append(a)
    quad_mpc_loss(torch.stack(sts, 1), ref[:, :H], torch.stack(acs, 1)).backward()
    want = [t.grad for t in ins]
    # fused
    ins = leaves()
    if mode == "lstm":
        loss, _, _ = F.quad_lstm_rollout_loss(net, ins[0], in_ref, ref, 0.1,
                                              dyn.params, ins[1], ins[2])
    else:
        loss, _, _ = F.quad_mlp_rollout_loss(net, ins[0], in_ref, ref, 0.1, dyn.params)
    (2.0 * loss).backward()
    for t, w in zip(ins, want):
        assert t.grad is not None
        assert rel_err(N(t.grad), 2.0 * N(w)) < 1e-4


def test_wing_and_cartpole_run_epoch(dev):
    """Multi-batch epochs of the other two systems through the real loader /
    SGD path: finite, decreasing losses; `epoch_loss = running_loss / i`."""
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import (
        CartpoleDynamics)
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.train_cartpole import TrainCartpole
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    torch.manual_seed(0)
    wcfg = dict(delta_t=0.05, delta_t_train=0.05, epoch_size=512, self_play=0,
                batch_size=128, state_size=12, horizon=10, ref_dim=3,
                action_dim=4, train_mode="concurrent",
                learning_rate_controller=1e-6, system="wing",
                resample_every=100)
    wt = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), wcfg)
    wt.initialize_model(device=dev, seed=2)
    wl = [wt.run_epoch("controller", epoch=e) for e in range(4)]
    assert all(np.isfinite(wl)) and wl[-1] < wl[0]
    ccfg = dict(delta_t=0.05, batch_size=64, state_size=4, horizon=5,
                action_dim=1, ref_dim=4, train_mode="concurrent",
                learning_rate_controller=1e-4, system="cartpole",
                sample_data=256)
    ct = TrainCartpole(CartpoleDynamics(), CartpoleDynamics(), ccfg)
    ct.initialize_model(device=dev, seed=3)
    before = ct.state_data.states.clone()
    cl = [ct.run_epoch("controller") for _ in range(4)]
    assert all(np.isfinite(cl)) and cl[-1] < cl[0]
    assert torch.equal(ct.state_data.states, before)   # policy input was copied
    assert ct.results_dict["loss_controller"] == cl


def test_learnt_dynamics_matches_reference(dev):
    """N3: LearntDynamics forward + every parameter gradient against the
    reference's autograd (golden G10), then a few `train_dynamics_model`
    steps that pull the learnable simulator towards a mismatched one."""
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import (
        LearntDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g = load_golden("learnt_dynamics.npz")
    dyn = LearntDynamics(initial_params={"rotational_drag": [.01, .02, .03]})
    dyn.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith("w.")})
    dyn.to(dev)
    target = FlightmareDynamics(modified_params=dict(
        translational_drag=[.1, .2, .3], rotational_drag=[.01, .02, .03],
        mass=1.0))
    state, action = D(g["state"], dev), D(g["action"], dev)
    d1 = dyn(state, action, float(g["dt"]))
    assert rel_err(N(d1), g["next"]) < 1e-6
    d2 = target(state, action, float(g["dt"]))
    assert rel_err(N(d2), g["target_next"]) < 1e-6
    loss = torch.sum((d1 - d2)**2)
    loss.backward()
    assert abs(loss.item() - g["loss"]) / g["loss"] < 1e-5
    for k, p in dyn.named_parameters():
        ref = g["g." + k]
        if k == "mass":
            assert float(p.grad.abs().max()) == 0.0 and float(np.abs(ref).max()) == 0.0
        elif k == "torch_inertia_vector":
            # autograd differentiates through J and inverse(J) separately;
            # the closed form keeps only what does not cancel
            assert rel_err(N(p.grad), ref) < 2e-3, k
        else:
            assert rel_err(N(p.grad), ref) < 1e-4, k
    # four momentum-SGD steps (golden G10 `steps.*`): the reference keeps
    # simulating with the kinv / inertia of construction time while the
    # parameters drift (its torch.diag copies, quad_dynamics_trained.py:48-50;
    # the inertia parameter even changes sign here) - the loss sequence only
    # matches if the drop-in does the same
    opt = torch.optim.SGD(dyn.parameters(), lr=1e-4, momentum=0.9)
    for want in g["steps.loss"]:
        opt.zero_grad()
        l = torch.sum((dyn(state, action, float(g["dt"])) - d2.detach())**2)
        l.backward()
        opt.step()
        assert abs(l.item() - want) / want < 2e-5, (l.item(), want)
    with torch.no_grad():
        assert rel_err(N(dyn(state, action, float(g["dt"]))), g["steps.next"]) < 1e-5
    for k, v in dyn.state_dict().items():
        tol = 5e-3 if k == "torch_inertia_vector" else 1e-4
        assert rel_err(N(v), g["steps.w." + k]) < tol, k
    # trainer-level: fitting reduces the one-step model error
    cfg = dict(delta_t=0.1, delta_t_train=0.1, epoch_size=512, self_play=0,
               batch_size=128, state_size=12, horizon=10,
               train_mode="concurrent", ref_dim=9, action_dim=4, l2_lambda=0.01,
               learning_rate_controller=1e-5, learning_rate_dynamics=1e-4,
               system="quad", modified_params={})
    torch.manual_seed(0)
    learnt = LearntDynamics().to(dev)
    trainer = TrainDrone(learnt, target, cfg)
    trainer.initialize_model(device=dev, seed=4)
    losses = [trainer.run_epoch("dynamics", epoch=e) for e in range(5)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0]


def test_controller_through_learnt_dynamics(dev):
    """N3, controller phase of run_dynamics: with a learnable simulator the
    unroll goes through LearntDynamics.forward (action transform, analytic
    step, residual network).  Two routes: the fused kernel
    (apg_quad_learnt_rollout_fwd_bwd, default) and the step-by-step autograd
    unroll (`fused_learnt = False`).  A freshly initialised LearntDynamics IS
    the analytic simulator (identity transform, zero residual), so both must
    equal the analytic fused step; with trained-looking weights the two routes
    must still agree with each other and differ from the analytic one."""
    import copy
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import (
        LearntDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    B = 96
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=77)
    state0, in_ref, ref = (d["state0"].to(dev), d["in_ref"].to(dev),
                           d["ref"].to(dev))
    torch.manual_seed(5)
    base = Net(15, 10, 9, 40, conv=1)
    learnt = LearntDynamics().to(dev)

    def one_step(train_dynamics, fused_learnt=True):
        t = TrainDrone(train_dynamics, FlightmareDynamics(), dict(QUAD_CFG))
        t.fused_learnt = fused_learnt
        t.net = copy.deepcopy(base).to(dev)
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=0.0)
        in_state = state_preprocessing(state0)
        fused = t.train_concurrent_fused(in_state, state0, in_ref, ref)
        if fused is not None:
            return float(fused), {k: N(p.grad) for k, p in
                                  t.net.named_parameters() if p.grad is not None}
        actions = torch.sigmoid(t.net(in_state, in_ref)).reshape(B, 10, 4)
        loss = t.train_controller_model(state0, actions, in_ref, ref)
        return float(loss.detach()), {k: N(p.grad) for k, p in
                             t.net.named_parameters() if p.grad is not None}

    def same(l1, g1, l2, g2, tol=1e-4):
        assert abs(l1 - l2) <= tol * abs(l1), (l1, l2)
        assert set(g1) == set(g2) and len(g1) >= 12
        for k in g1:
            assert rel_err(g2[k], g1[k]) < tol, k

    l_an, g_an = one_step(FlightmareDynamics())
    same(l_an, g_an, *one_step(learnt, True))
    for p in learnt.parameters():
        p.grad = None
    same(l_an, g_an, *one_step(learnt, False))
    # only the autograd unroll differentiates the simulator's own parameters
    assert learnt.linear_at.grad is not None
    assert float(learnt.linear_state_2.bias.grad.abs().sum()) > 0
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        learnt.linear_at.add_(0.05 * torch.randn(4, 4, generator=g).to(dev))
        for lin, sc in ((learnt.linear_state_1, 0.2), (learnt.linear_state_2, 0.02)):
            lin.weight.add_(sc * torch.randn(lin.weight.shape, generator=g).to(dev))
            lin.bias.add_(sc * torch.randn(lin.bias.shape, generator=g).to(dev))
    for p in learnt.parameters():
        p.grad = None
    l_f, g_f = one_step(learnt, True)
    assert all(p.grad is None for p in learnt.parameters())
    l_s, g_s = one_step(learnt, False)
    same(l_s, g_s, l_f, g_f)
    assert abs(l_f - l_an) > 1e-3 * abs(l_an)


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("B,H", [(1, 1), (8, 1), (100, 10), (257, 5), (64, 24)])
def test_learnt_rollout_kernel_matches_autograd_unroll(dev, B, H, layout):
    """apg_quad_learnt_rollout_fwd_bwd against an autograd unroll through the
    golden-pinned LearntDynamics.forward + the quad_mpc_loss kernel: loss,
    dL/dactions, dL/dstate0, and every intermediate state.  H = 1 with the
    golden weights reproduces G10's `next` through the kernel."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.drone_loss import quad_mpc_loss
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import (
        LearntDynamics)
    g = load_golden("learnt_dynamics.npz")
    dyn = LearntDynamics(initial_params={"rotational_drag": [.01, .02, .03]})
    dyn.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith("w.")})
    dyn.to(dev)
    dt = float(g["dt"])
    if (B, H) == (8, 1):
        state0, actions = D(g["state"], dev), D(g["action"], dev)[:, None, :]
        B = state0.shape[0]
        ref = torch.zeros(B, 1, 9, device=dev)
    else:
        d = synthetic.quad_polynomial_batch(B, H, dt, seed=B + H)
        state0, ref = d["state0"].to(dev), d["ref"].to(dev)
        actions = torch.rand(B, H, 4, generator=torch.Generator().manual_seed(B)).to(dev)
    s0 = state0.clone().requires_grad_(True)
    a = actions.clone().requires_grad_(True)
    s, states = s0, []
    for k in range(H):
        s = dyn(s, a[:, k], dt)
        states.append(s)
    states = torch.stack(states, 1)
    loss = quad_mpc_loss(states, ref, a)
    loss.backward()
    if layout == "soa":
        args = (synthetic.to_soa_state(state0), synthetic.to_soa_seq(actions),
                synthetic.to_soa_seq(ref))
    else:
        args = (state0, actions, ref)
    res = F.quad_learnt_rollout_fwd_bwd(dyn, *args, dt, layout=layout,
                                        want_states=True)
    soa = layout == "soa"
    seq = synthetic.from_soa_seq if soa else (lambda t: t)
    st = synthetic.from_soa_state if soa else (lambda t: t)
    if states.shape[0] == g["next"].shape[0] and H == 1:
        assert rel_err(N(seq(res["states"]))[:, 0], g["next"]) < 1e-5
    assert rel_err(N(seq(res["states"])), N(states.detach())) < 1e-5
    loss = loss.detach()
    assert abs(float(res["loss"]) - float(loss)) <= 1e-5 * abs(float(loss))
    assert rel_err(N(seq(res["grad_actions"])), N(a.grad)) < 1e-4
    assert rel_err(N(st(res["grad_state0"])), N(s0.grad)) < 1e-4


def test_learnt_rollout_kernel_rejects_bad_arguments(dev):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import (
        LearntDynamics)
    dyn = LearntDynamics().to(dev)
    s = torch.zeros(4, 12, device=dev)
    a = torch.zeros(4, 10, 4, device=dev)
    r = torch.zeros(4, 10, 9, device=dev)
    with pytest.raises(ValueError):
        F.quad_learnt_rollout_fwd_bwd(dyn, s, a, r, 0.1, layout="packed")
    with pytest.raises(ValueError):
        F.quad_learnt_rollout_fwd_bwd(dyn, s, a, r[:, :9], 0.1)
    with pytest.raises(RuntimeError):
        F.quad_learnt_rollout_fwd_bwd(LearntDynamics(), s, a, r, 0.1)
    with pytest.raises(ValueError):           # horizon beyond the LDS stash
        F.quad_learnt_rollout_fwd_bwd(dyn, s, torch.zeros(4, 400, 4, device=dev),
                                      torch.zeros(4, 400, 9, device=dev), 0.1)


@pytest.mark.parametrize("B", [1, 100, 129])
def test_fused_lstm_ragged_batches_match_unfused(dev, B):
    """K7 on batch sizes that do not fill a workgroup: loss and parameter
    gradients equal the per-step-kernel path (same golden-pinned semantics)."""
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(QUAD_CFG, batch_size=B, train_mode="LSTM")
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=40 + B, ref_length=20)
    state0, in_ref, ref = (d["state0"].to(dev), d["in_ref"].to(dev),
                           d["ref"].to(dev))
    g = torch.Generator().manual_seed(B)
    h0, c0 = (torch.randn(B, 8, generator=g).to(dev),
              torch.randn(B, 8, generator=g).to(dev))
    torch.manual_seed(7)
    base = LSTM_NEW(15, 10, 9, 4, conv=1)
    results = []
    for fused in (False, True):
        trainer = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
        import copy
        net = copy.deepcopy(base).to(dev)

        def fixed_reset(batch_size=1, generator=None, net=net):
            net.hidden_state, net.cell_state = h0.clone(), c0.clone()
        net.reset_hidden_state = fixed_reset
        trainer.net = net
        trainer.fused_policy = fused
        trainer.optimizer_controller = torch.optim.SGD(net.parameters(), lr=0.0)
        loss = trainer.train_recurrent_model(None, state0, in_ref, ref)
        results.append((loss.item(), {k: N(p.grad) for k, p in
                                      net.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = results
    assert abs(l0 - l1) / abs(l0) < 1e-5
    assert set(g0) == set(g1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-4, k


@pytest.mark.parametrize("case", ["train", "test", "tight", "tight_test"])
def test_closed_loop_matches_reference_evaluator(dev, case):
    """N2 / G11: the batched closed-loop kernel against the reference's
    QuadEvaluator.follow_trajectory with the shipped quad controller: drone
    states, projected reference, divergences, actions and the number of steps
    - tracking, break (test_time) and reset-to-reference branches."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("closed_loop.npz")
    ck = load_golden("checkpoints.npz")
    sd = {k[len("quad.w."):]: torch.from_numpy(ck[k]) for k in ck.files
          if k.startswith("quad.w.")}
    net = build_policy("quad", sd).to(dev)
    traj = D(g["trajs"], dev).clone()
    traj[:, :, 2] += 3          # Random.__init__, random_traj.py:34
    out = F.quad_mlp_closed_loop(
        net, traj, float(g["dt"]), FlightmareDynamics().params,
        max_steps=int(g["max_steps"]), thresh_div=float(g[f"{case}.thresh_div"]),
        thresh_stable=float(g[f"{case}.thresh_stable"]),
        test_time=int(g[f"{case}.test_time"]), want_trajectory=True)
    for i in range(traj.shape[0]):
        n = len(g[f"{case}.{i}.div"])
        assert int(out["steps"][i]) == n, (case, i)
        assert rel_err(N(out["drone"][:n + 1, :, i]), g[f"{case}.{i}.drone"]) < 1e-4
        assert np.abs(N(out["div"][:n, i]) - g[f"{case}.{i}.div"]).max() < 2e-4
        assert rel_err(N(out["actions"][:n, :, i]), g[f"{case}.{i}.actions"][:, 0]) < 1e-4


def test_closed_loop_evaluator_and_oracle_large_batch(dev):
    """N2 at a batch that spans several workgroups with a ragged tail: the
    kernel against the batched oracle loop (same net, random smooth
    trajectories), and QuadEvaluator.run_eval statistics against the same
    statistics computed from the oracle's divergences."""
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.evaluate_drone import QuadEvaluator
    from oracle import torch_port as tp
    ck = load_golden("checkpoints.npz")
    sd = {k[len("quad.w."):]: torch.from_numpy(ck[k]) for k in ck.files
          if k.startswith("quad.w.")}
    B, L, steps = 300, 40, 30
    traj = synthetic.quad_eval_trajectories(B, L, 0.1, seed=5)
    traj[:, :, 2] += 3
    for test_time, td in ((0, 0.25), (1, 0.25)):
        net_cpu = build_policy("quad", sd)
        ref = tp.quad_closed_loop(net_cpu, tp.QuadOracle(), traj, 0.1, 10, steps,
                                  td, 1.0, test_time)
        ev = QuadEvaluator(build_policy("quad", sd).to(dev), FlightmareDynamics(),
                           ref_length=10, dt=0.1, test_time=test_time)
        _, drone, divs, acts = ev.follow_trajectory(
            "rand", max_nr_steps=steps, thresh_stable=1.0, thresh_div=td,
            trajectories=traj)
        # a divergence within rounding of the threshold may flip a branch:
        # compare the trajectories that took the same branches
        same = [i for i in range(B) if len(divs[i]) == int(ref["steps"][i])]
        assert len(same) > 0.97 * B
        worst = 0.0
        for i in same:
            n = len(divs[i])
            worst = max(worst, float(np.abs(N(divs[i]) - ref["div"][i, :n].numpy()).max()))
        # resets re-synchronise both sides, a flipped reset shows up as one
        # large difference: allow a handful of trajectories to disagree
        bad = [i for i in same
               if np.abs(N(divs[i]) - ref["div"][i, :len(divs[i])].numpy()).max() > 2e-3]
        assert len(bad) <= 0.03 * B, (len(bad), worst)
        stats = ev.run_eval("rand", nr_test=B, max_steps=steps, thresh_div=td,
                            thresh_stable=1.0, trajectories=traj)
        assert all(np.isfinite(stats[k]) for k in (0, 1, 4, 5))
        st = np.array([int((ref["div"][i, :int(ref["steps"][i])] < td).sum())
                       for i in range(B)])
        assert abs(stats[0] - st.mean()) < 0.05 * max(1.0, st.mean())


def test_self_play_slots_and_evaluate_model(dev):
    """N1/N2: the data set's self-play slots (DroneDataset.get_eval_index /
    get_and_add_eval_data semantics, vectorised) and TrainDrone.evaluate_model
    (closed-loop statistics, slot replacement, threshold curriculum)."""
    from apg_trajectory_tracking_amd.dataset import SyntheticQuadDataset
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from oracle import torch_port as tp
    ds = SyntheticQuadDataset(40, 10, 0.1, seed=3, device=dev, self_play=0.5)
    assert (ds.num_sampled_states, ds.num_self_play, len(ds)) == (40, 20, 60)
    before = ds.states.clone()
    g = torch.Generator().manual_seed(1)
    st = torch.randn(7, 12, generator=g) * 0.3
    win = torch.randn(7, 10, 9, generator=g)
    assert ds.add_eval_data(st.to(dev), win.to(dev)) == 7
    sl = slice(40, 47)
    z = st.clone()
    z[:, :3] = 0
    assert torch.equal(ds.states[sl].cpu(), z)
    assert rel_err(N(ds.normed_states[sl]), tp.quad_state_features(z).numpy()) < 1e-6
    rel = win.clone()
    rel[:, :, :3] -= st[:, None, :3]
    want_in = torch.cat((rel[:, :, :3], rel[:, :, 6:9], rel[:, :, 6:9] - st[:, None, 6:9]), 2)
    assert rel_err(N(ds.in_ref_states[sl]), want_in.numpy()) < 1e-6
    assert rel_err(N(ds.ref_states[sl]), rel.numpy()) < 1e-6
    assert torch.equal(ds.states[:40], before[:40]) and torch.equal(ds.states[47:], before[47:])
    # wrap-around: 30 more entries, counter 7 -> 37; the last 20 survive
    st2 = torch.randn(30, 12, generator=g) * 0.3
    ds.add_eval_data(st2.to(dev), torch.randn(30, 10, 9, generator=g).to(dev))
    assert ds.eval_counter == 37 and ds.get_eval_index() == 40 + 17
    for j in range(10, 30):
        slot = 40 + (7 + j) % 20
        assert torch.equal(ds.states[slot, 3:].cpu(), st2[j, 3:])
    ds.resample_data()      # renews the sampled part only
    assert not torch.equal(ds.states[:40], before[:40])
    assert torch.equal(ds.states[slot, 3:].cpu(), st2[29, 3:])

    cfg = dict(QUAD_CFG, batch_size=32, epoch_size=64, self_play=0.5,
               self_play_every_x=5, train_mode="concurrent", nr_test=6,
               max_steps=40, thresh_div_start=1.0, thresh_div_end=2.0,
               thresh_stable_start=1.0)
    trainer = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), cfg)
    trainer.initialize_model(device=dev, seed=2)
    trainer.save_path = "/tmp/apg_eval_test"
    data = trainer.state_data
    assert data.num_self_play == 32
    old = data.states[64:].clone()
    res = trainer.evaluate_model(0)
    assert res is not None and np.isfinite(res[0])
    assert data.eval_counter == 6 * 40 // 5          # every 5th policy call
    assert not torch.equal(data.states[64:], old)
    assert abs(trainer.config["thresh_div"] - 1.05) < 1e-9
    for key in ("mean_success", "std_success", "mean_divergence", "thresh_div"):
        assert len(trainer.results_dict[key]) == 1


@pytest.mark.parametrize("case", ["lstm_train", "lstm_test"])
def test_closed_loop_lstm_matches_reference_evaluator(dev, case):
    """N2 / G11 with the LSTM controller (hidden state carried through the run)."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    g = load_golden("closed_loop.npz")
    net = LSTM_NEW(15, 10, 9, 4, conv=1)
    load_weights(net, g, "lstm.w.")
    net.to(dev)
    traj = D(g["trajs"], dev).clone()
    traj[:, :, 2] += 3
    out = F.quad_lstm_closed_loop(
        net, traj, float(g["dt"]), FlightmareDynamics().params,
        D(g["lstm.h0"], dev), D(g["lstm.c0"], dev), max_steps=int(g["max_steps"]),
        thresh_div=float(g[f"{case}.thresh_div"]),
        thresh_stable=float(g[f"{case}.thresh_stable"]),
        test_time=int(g[f"{case}.test_time"]), want_trajectory=True)
    for i in range(traj.shape[0]):
        n = len(g[f"{case}.{i}.div"])
        assert int(out["steps"][i]) == n, (case, i)
        assert rel_err(N(out["drone"][:n + 1, :, i]), g[f"{case}.{i}.drone"]) < 1e-4
        assert np.abs(N(out["div"][:n, i]) - g[f"{case}.{i}.div"]).max() < 2e-4
        assert rel_err(N(out["actions"][:n, :, i]), g[f"{case}.{i}.actions"]) < 1e-4


@pytest.mark.parametrize("B", [1, 100, 700])
def test_wing_concurrent_fused_policy_matches_unfused(dev, B):
    """Fixed-wing concurrent step: policy on the matrix cores around the fused
    rollout (apg_wing_policy_fwd/_bwd) == torch policy + the same rollout:
    loss, every parameter gradient, and the weights after two SGD steps."""
    import copy
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    cfg = dict(delta_t=0.05, delta_t_train=0.05, epoch_size=B, self_play=0,
               batch_size=B, state_size=12, horizon=20, ref_dim=3, action_dim=4,
               learning_rate_controller=1e-7, system="fixed_wing", modified_params={})
    data = SyntheticWingDataset(B, 20, 0.05, seed=31 + B, device=dev)
    torch.manual_seed(6)
    base = Net(9, 1, 3, 80, conv=False)
    out = []
    for fused in (False, True):
        t = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), dict(cfg))
        t.net = copy.deepcopy(base).to(dev)
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-7, momentum=0.9)
        losses, grads = [], None
        for step in range(2):
            if fused:
                loss = t.train_concurrent_fused(data.normed_states, data.states,
                                                data.in_ref_states, data.ref_states)
                assert loss is not None
            else:
                acts = torch.sigmoid(t.net(data.normed_states, data.in_ref_states))
                loss = t.train_controller_model(data.states, acts.reshape(-1, 20, 4),
                                                data.in_ref_states, data.ref_states)
            losses.append(loss.item())
            if step == 0:
                grads = {k: N(p.grad) for k, p in t.net.named_parameters()
                         if p.grad is not None}
        out.append((losses, grads, {k: N(v) for k, v in t.net.state_dict().items()}))
    (l0, g0, w0), (l1, g1, w1) = out
    assert np.allclose(l0, l1, rtol=1e-5)
    assert set(g0) == set(g1)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 1e-4, k
    for k in w0:
        assert rel_err(w1[k], w0[k]) < 1e-5, k


def test_fused_ar_large_batch_is_chunked(dev, monkeypatch):
    """Beyond the per-launch batch limit of the fused autoregressive path the
    direct-gradient function processes chunks and adds them up."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B = 1000
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=8, ref_length=20)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    torch.manual_seed(2)
    net = Net(15, 10, 9, 4, conv=1).to(dev)
    dyn = FlightmareDynamics()
    l0, g0, _ = F.quad_mlp_rollout_grads(net, s0, in_ref, ref, 0.1, dyn.params)
    g0 = {k: v.clone() for k, v in g0.items()}
    monkeypatch.setattr(F, "_MAX_FUSED_AR_BATCH", 384)     # 3 ragged chunks
    l1, g1, flat = F.quad_mlp_rollout_grads(net, s0, in_ref, ref, 0.1, dyn.params)
    assert abs(l0.item() - l1.item()) / l0.item() < 1e-5
    for k in g0:
        assert g1[k].data_ptr() >= flat.data_ptr()          # still views of `flat`
        assert rel_err(N(g1[k]), N(g0[k])) < 1e-4, k


def test_run_epoch_indexed_fused_path_equals_batch_path(dev):
    """run_epoch's fast path (fused concurrent step with the minibatch gather
    folded into the layout change, functional.to_soa(index=...)) against the
    same fused step fed with materialised batches."""
    import copy
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    t = torch.randn(50, 4, 3, device=dev)
    idx = torch.tensor([7, 0, 49, 7, 13], device=dev)
    assert torch.equal(F.to_soa(t, index=idx), t[idx].permute(1, 2, 0).contiguous())
    cfg = dict(QUAD_CFG, batch_size=96, epoch_size=300, self_play=0,
               learning_rate_controller=1e-7)
    a = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
    a.shuffle = False
    a.initialize_model(device=dev, seed=3)
    b = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
    b.shuffle = False
    b.initialize_model(device=dev, seed=3)
    b.net.load_state_dict(copy.deepcopy(a.net.state_dict()))
    assert a.train_concurrent_fused(None, None, None, None, probe=True)
    la = a.run_epoch("controller", 0)
    losses = []
    d = b.state_data
    for lo in range(0, 300, 96):
        sl = slice(lo, lo + 96)
        losses.append(b.train_concurrent_fused(
            d.normed_states[sl], d.states[sl], d.in_ref_states[sl],
            d.ref_states[sl]).item())
    assert abs(la - sum(losses) / (len(losses) - 1)) / abs(la) < 1e-5
    for (k, va), (_, vb) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert rel_err(N(va), N(vb)) < 1e-6, k


def test_wing_run_epoch_indexed_fused_path(dev):
    """Fixed-wing run_epoch at H = 20: the indexed fused path (gather folded
    into the layout change) == the fused step on materialised batches."""
    import copy
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    cfg = dict(delta_t=0.05, delta_t_train=0.05, epoch_size=200, self_play=0,
               batch_size=64, state_size=12, horizon=20, ref_dim=3, action_dim=4,
               learning_rate_controller=1e-7, system="fixed_wing", modified_params={})
    a = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), dict(cfg))
    a.shuffle = False
    a.initialize_model(device=dev, seed=5)
    b = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), dict(cfg))
    b.shuffle = False
    b.initialize_model(device=dev, seed=5)
    b.net.load_state_dict(copy.deepcopy(a.net.state_dict()))
    assert a.train_concurrent_fused(None, None, None, None, probe=True)
    la = a.run_epoch("controller", 0)
    d, losses = b.state_data, []
    for lo in range(0, 200, 64):
        sl = slice(lo, lo + 64)
        losses.append(b.train_concurrent_fused(
            d.normed_states[sl], d.states[sl], d.in_ref_states[sl],
            d.ref_states[sl]).item())
    assert abs(la - sum(losses) / (len(losses) - 1)) / abs(la) < 1e-5
    for (k, va), (_, vb) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert rel_err(N(va), N(vb)) < 1e-6, k


@pytest.mark.parametrize("mode", ["autoregressive", "LSTM"])
def test_recurrent_run_epoch_indexed_path(dev, mode):
    """run_epoch for the recurrent modes: index batches handed to the fused
    unroll (gather inside the layout change) == materialised batches."""
    import copy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(QUAD_CFG, batch_size=80, epoch_size=250, self_play=0,
               train_mode=mode, learning_rate_controller=1e-7)
    a = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
    a.shuffle = False
    a.initialize_model(device=dev, seed=4)
    b = TrainDrone(FlightmareDynamics(), FlightmareDynamics(), dict(cfg))
    b.shuffle = False
    b.initialize_model(device=dev, seed=4)
    b.net.load_state_dict(copy.deepcopy(a.net.state_dict()))
    a.hidden_generator = torch.Generator().manual_seed(5)
    b.hidden_generator = torch.Generator().manual_seed(5)
    assert a.recurrent_indexed_ok()
    la = a.run_epoch("controller", 0)
    d, losses = b.state_data, []
    for lo in range(0, 250, 80):
        sl = slice(lo, lo + 80)
        losses.append(b.train_recurrent_model(
            d.normed_states[sl], d.states[sl], d.in_ref_states[sl],
            d.ref_states[sl]).item())
    assert abs(la - sum(losses) / (len(losses) - 1)) / abs(la) < 1e-5
    for (k, va), (_, vb) in zip(a.net.state_dict().items(), b.net.state_dict().items()):
        assert rel_err(N(va), N(vb)) < 1e-6, k


@pytest.mark.parametrize("mode", ["concurrent", "autoregressive", "LSTM"])
def test_fused_policy_paths_at_baseline_batch(dev, mode):
    """BASELINE.json size (65 536 trajectories, H = 10): the fused-policy
    training step of every mode against the per-step / torch-policy path on the
    same data - loss and every parameter gradient (sums over the batch)."""
    import copy
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    B = 65536
    conc = mode == "concurrent"
    d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=17,
                                        ref_length=10 if conc else 20)
    state0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    with torch.no_grad():
        normed = state_preprocessing(state0)
    torch.manual_seed(12)
    base = (LSTM_NEW(15, 10, 9, 4, conv=1) if mode == "LSTM"
            else Net(15, 10, 9, 40 if conc else 4, conv=1))
    gen = torch.Generator().manual_seed(3)
    h0, c0 = (torch.randn(B, 8, generator=gen).to(dev) for _ in range(2))
    res = []
    for fused in (False, True):
        t = TrainDrone(FlightmareDynamics(), FlightmareDynamics(),
                       dict(QUAD_CFG, batch_size=B, train_mode=mode))
        t.net = copy.deepcopy(base).to(dev)
        if mode == "LSTM":
            def fixed_reset(batch_size=1, generator=None, net=t.net):
                net.hidden_state, net.cell_state = h0.clone(), c0.clone()
            t.net.reset_hidden_state = fixed_reset
        t.fused_policy = fused
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=0.0)
        if conc and fused:
            loss = t.train_concurrent_fused(normed, state0, in_ref, ref)
        elif conc:
            acts = torch.sigmoid(t.net(normed, in_ref)).reshape(-1, 10, 4)
            loss = t.train_controller_model(state0, acts, in_ref, ref)
        else:
            loss = t.train_recurrent_model(normed, state0, in_ref, ref)
        res.append((loss.item(), {k: N(p.grad) for k, p in
                                  t.net.named_parameters() if p.grad is not None}))
        del t
        torch.cuda.empty_cache()
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) / abs(l0) < 2e-5
    assert set(g0) == set(g1)
    for k in g0:      # 655 360-term fp32 sums in different orders
        assert rel_err(g1[k], g0[k]) < 1e-3, k


@pytest.mark.parametrize("fused", [False, True])
def test_wing_train_controller_two_sgd_steps(dev, fused):
    """G12: the fixed-wing training step against the reference trainer's
    recording - torch policy around the fused rollout, and the policy on the
    matrix cores (apg_wing_policy_fwd/_bwd)."""
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    g = load_golden("wing_train.npz")
    cfg = dict(delta_t=float(g["dt"]), delta_t_train=float(g["dt"]), epoch_size=48,
               self_play=0, batch_size=48, state_size=12, horizon=20, ref_dim=3,
               action_dim=4, learning_rate_controller=float(g["lr"]),
               system="fixed_wing", modified_params={})
    t = TrainFixedWing(FixedWingDynamics(), FixedWingDynamics(), cfg)
    net = Net(9, 1, 3, 80, conv=False)
    load_weights(net, g, "w0.")
    t.net = net.to(dev)
    t.optimizer_controller = torch.optim.SGD(
        t.net.parameters(), lr=float(g["lr"]), momentum=float(g["momentum"]))
    in_state, in_ref = D(g["in_state"], dev), D(g["in_ref"], dev)
    state0, ref = D(g["state0"], dev), D(g["ref"], dev)
    for step in (1, 2):
        if fused:
            loss = t.train_concurrent_fused(in_state, state0, in_ref, ref)
            assert loss is not None
        else:
            acts = torch.sigmoid(t.net(in_state, in_ref)).reshape(-1, 20, 4)
            loss = t.train_controller_model(state0, acts, in_ref, ref)
        assert abs(loss.item() - g[f"loss{step}"]) / g[f"loss{step}"] < 1e-5
        if step == 1:
            for k, p in t.net.named_parameters():
                if "g1." + k in g.files:
                    assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k
        for k, v in t.net.state_dict().items():
            assert rel_err(N(v), g[f"w{step}.{k}"]) < 1e-5, (step, k)


@pytest.mark.filterwarnings("ignore::RuntimeWarning")   # mean of no complete run
def test_train_control_and_train_dynamics_end_to_end(dev, tmp_path, monkeypatch):
    """The reference's entry points (scripts/train_drone.py:241-278) run
    through: evaluation (closed loop + self play) -> resampling -> epoch, for a
    few epochs; run_dynamics fits the learnt simulator first and then trains the
    controller through it, step by step."""
    from apg_trajectory_tracking_amd import train_drone
    monkeypatch.chdir(tmp_path)
    cfg = dict(QUAD_CFG, epoch_size=256, batch_size=64, self_play=0.5,
               self_play_every_x=3, nr_test=4, max_steps=30, nr_epochs=4,
               resample_every=2, thresh_div_start=1.0, thresh_div_end=2.0,
               thresh_stable_start=1.0, learning_rate_controller=1e-6,
               save_name="e2e")
    torch.manual_seed(0)
    t = train_drone.train_control(None, dict(cfg), device=dev)
    assert len(t.results_dict["loss"]) == 1 + 4
    assert all(np.isfinite(t.results_dict["loss"]))
    assert len(t.results_dict["mean_success"]) == 4
    assert t.state_data.eval_counter == 4 * (4 * 30 // 3)
    assert t.sampled_data_count == 2 * 256
    out = tmp_path / "trained_models" / "quad" / "e2e"
    # scripts/train_base.py:253-287: weights, every statistics table,
    # results.json; a checkpoint per evaluated epoch but the first
    assert sorted(os.listdir(out)) == sorted([
        "config.json", "loss.csv", "mean_successes.csv", "std_success.csv",
        "mean_divergence.csv", "std_divergence.csv", "mean_divergence_full.csv",
        "std_divergence_full.csv", "results.json", "model_quad", "model_quad1",
        "model_quad2", "model_quad3"])
    assert len(np.loadtxt(out / "mean_successes.csv", delimiter=",")) == 4
    sd = torch.load(out / "model_quad", map_location="cpu")
    assert sd["fc_out.weight"].shape == (40, 64)

    cfg2 = dict(cfg, nr_epochs=3, train_dyn_for_epochs=1, save_name="e2e_dyn",
                learning_rate_dynamics=1e-5, l2_lambda=0.01,
                modified_params={"rotational_drag": [.01, .02, .03]})
    t2 = train_drone.train_dynamics(None, dict(cfg2), device=dev)
    assert t2.results_dict["trained"] == ["dynamics", "dynamics", "controller"]
    assert all(np.isfinite(t2.results_dict["loss"]))
    assert t2.count_finetune_data == 2 * 384
    assert len(t2.results_dict["mean_success"]) == 3     # flown in the analytic env
    # the fitted simulator is saved next to the policy (:279-285)
    dyn_sd = torch.load(tmp_path / "trained_models" / "quad" / "e2e_dyn" /
                        "dynamics_model", map_location="cpu")
    assert "linear_at" in dyn_sd and "torch_kinv_vector" in dyn_sd


@pytest.mark.filterwarnings("ignore::RuntimeWarning")   # mean of no complete run
@pytest.mark.parametrize("case", ["train", "test"])
def test_self_play_matches_the_reference_evaluator_gpu(dev, case):
    """GPU twin of tests/test_host_cpu.py::test_self_play_matches_the_
    reference_evaluator (golden G13): the reference's REAL
    QuadEvaluator.run_eval + NetworkWrapper + QuadDataset flew the G11
    trajectories with self play on.  Here NOTHING is replaced: the closed loop
    is mlp_closed_loop_kernel, the features are apg_quad_features_fwd, the
    slots are filled on the device by SyntheticQuadDataset.add_eval_data
    (neural_control/dataset.py:88-119,155-204; scripts/evaluate_drone.py:
    237-300)."""
    from apg_trajectory_tracking_amd import dataset as ds_mod
    from apg_trajectory_tracking_amd import evaluate_drone
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g, traj_g = load_golden("self_play.npz"), load_golden("closed_loop.npz")
    ck = load_golden("checkpoints.npz")
    net = build_policy("quad", {k[len("quad.w."):]: torch.from_numpy(ck[k])
                                for k in ck.files if k.startswith("quad.w.")})
    net.to(dev)
    n_s, n_p = int(g["num_sampled"]), int(g["num_self_play"])
    data = ds_mod.SyntheticQuadDataset.__new__(ds_mod.SyntheticQuadDataset)
    data.num_sampled_states, data.num_self_play = n_s, n_p
    data.ref_length, data.device, data.eval_counter = 10, dev, 0
    data.normed_states = torch.zeros(n_s + n_p, 15, device=dev)
    data.states = torch.zeros(n_s + n_p, 12, device=dev)
    data.in_ref_states = torch.zeros(n_s + n_p, 10, 9, device=dev)
    data.ref_states = torch.zeros(n_s + n_p, 10, 9, device=dev)
    traj = torch.from_numpy(traj_g["trajs"]).clone()
    traj[:, :, 2] += 3                      # Random.__init__ lifts the reference
    ev = evaluate_drone.QuadEvaluator(net, FlightmareDynamics(), ref_length=10,
                                      dt=0.1, test_time=int(g[f"{case}.test_time"]))
    stats = ev.run_eval("rand", nr_test=traj.shape[0], max_steps=int(g["max_steps"]),
                        thresh_div=float(g[f"{case}.thresh_div"]), thresh_stable=1.0,
                        trajectories=traj.to(dev), dataset=data,
                        take_every_x=int(g["take_every_x"]))
    np.testing.assert_allclose(stats, g[f"{case}.stats"], rtol=2e-4, equal_nan=True)
    assert data.eval_counter == int(g[f"{case}.eval_counter"])
    sl = slice(n_s, None)
    for name, got in (("states", data.states), ("normed", data.normed_states),
                      ("in_ref", data.in_ref_states), ("ref", data.ref_states)):
        assert got.is_cuda
        assert rel_err(N(got[sl]), g[f"{case}.{name}"]) < 2e-4, name
    assert torch.count_nonzero(data.states[:n_s]) == 0     # sampled part untouched


def test_quad_packed_path_two_sgd_steps(dev):
    """G3 (scripts/train_drone.py:175-203 recorded from the real TrainDrone)
    through the row-layout path: Net.forward_packed -> [H, B, 4] action rows ->
    quad_rollout_rows_kernel (APG_LAYOUT_PACKED) -> dL/dactions rows back into
    the head's backward.  Loss, every gradient, post-SGD weights of two steps."""
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    g = load_golden("quad_train.npz")
    trainer = make_trainer(TrainDrone, FlightmareDynamics(), QUAD_CFG)
    net = Net(15, 10, 9, 40, conv=1)
    load_weights(net, g, "w0.")
    trainer.net = net.to(dev)
    trainer.optimizer_controller = torch.optim.SGD(
        trainer.net.parameters(), lr=float(g["lr"]), momentum=float(g["momentum"]))
    in_state, in_ref = D(g["in_state"], dev), D(g["in_ref"], dev)
    s0_rows = synthetic.to_packed_state(D(g["state0"], dev))
    ref = D(g["ref"], dev)
    ref_rows = synthetic.to_packed_seq(torch.cat((ref[:, :, :3], ref[:, :, 6:9]), 2))
    for step in (1, 2):
        loss = trainer.train_controller_packed(in_state, in_ref, s0_rows, ref_rows)
        assert abs(loss.item() - g[f"loss{step}"]) / g[f"loss{step}"] < 1e-5
        if step == 1:
            for k, p in trainer.net.named_parameters():
                if "g1." + k in g.files:
                    assert rel_err(N(p.grad), g["g1." + k]) < 1e-4, k
        for k, v in trainer.net.state_dict().items():
            assert rel_err(N(v), g[f"w{step}.{k}"]) < 1e-5, (step, k)


class _PlainPolicy(torch.nn.Module):
    """Not the reference architecture (no conv branch, relu, other widths):
    what `run_epoch` must still drive through the fast kernel."""

    def __init__(self, horizon=10):
        super().__init__()
        self.a = torch.nn.Linear(15 + horizon * 9, 48)
        self.b = torch.nn.Linear(48, 4 * horizon)

    def forward(self, state, ref):
        x = torch.cat((state, ref.flatten(1)), 1)
        return self.b(torch.relu(self.a(x)))


def test_run_epoch_packed_path_with_an_arbitrary_policy(dev):
    """run_epoch with a PyTorch policy that is NOT the reference architecture:
    the epoch goes through TrainDrone.train_controller_packed (row-layout
    tensors, quad_rollout_rows_kernel) and reproduces - epoch loss and every
    weight after three optimizer steps - the reference-layout (AoS) path."""
    from apg_trajectory_tracking_amd.dataset import SyntheticQuadDataset
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    import apg_trajectory_tracking_amd.functional as F
    cfg = dict(QUAD_CFG, batch_size=256, learning_rate_controller=1e-7)
    torch.manual_seed(5)
    proto = _PlainPolicy()
    results = []
    for packed in (True, False):
        t = make_trainer(TrainDrone, FlightmareDynamics(), cfg)
        t.state_data = SyntheticQuadDataset(700, 10, 0.1, seed=11, device=dev)
        t.net = _PlainPolicy().to(dev)
        t.net.load_state_dict(proto.state_dict())
        t.shuffle = False
        t.use_packed_path = packed
        t.init_optimizer()
        assert t.packed_path_ok()
        calls = []
        real = F.quad_rollout_fwd_bwd

        def spy(*a, **kw):
            calls.append(kw.get("layout", "aos"))
            return real(*a, **kw)
        F.quad_rollout_fwd_bwd = spy
        try:
            loss = t.run_epoch(train="controller", epoch=0)
        finally:
            F.quad_rollout_fwd_bwd = real
        assert calls == (["packed"] * 3 if packed else ["aos"] * 3), calls
        results.append((loss, {k: N(v) for k, v in t.net.state_dict().items()}))
    (l_p, w_p), (l_a, w_a) = results
    assert abs(l_p - l_a) / abs(l_a) < 1e-5
    for k in w_a:
        assert rel_err(w_p[k], w_a[k]) < 1e-5, k
    # the whole-set batch is the cached packed tensors themselves (no gather)
    t = make_trainer(TrainDrone, FlightmareDynamics(), dict(cfg, batch_size=700))
    t.state_data = SyntheticQuadDataset(700, 10, 0.1, seed=11, device=dev)
    t.net = _PlainPolicy().to(dev)
    t.shuffle = False
    t.init_optimizer()
    seen = []
    real = t.train_controller_packed
    t.train_controller_packed = lambda *a: (seen.append(a), real(*a))[1]
    with pytest.raises(ZeroDivisionError):   # one batch: running_loss / 0, as upstream
        t.run_epoch(train="controller", epoch=0)
    s0_rows, ref_rows = t.state_data.packed()
    assert seen[0][2].data_ptr() == s0_rows.data_ptr()
    assert seen[0][3].data_ptr() == ref_rows.data_ptr()


@pytest.mark.parametrize("mode", ["concurrent", "autoregressive", "LSTM"])
def test_static_shard_keeps_plane_copies_only_while_unchanged(dev, mode):
    """TrainDrone.static_shard: stepping on the same resident tensors re-uses
    their plane-layout copies (functional._StaticPlanes) - results equal the
    uncached trainer's step by step, and an in-place change of the data
    (`resample_data` style) is seen by the next step."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from apg_trajectory_tracking_amd.train_base import momentum_sgd
    B, H = 300, 10
    cfg = dict(QUAD_CFG, train_mode=mode, batch_size=B, learning_rate_controller=1e-7)
    R = H if mode == "concurrent" else 2 * H
    d = synthetic.quad_polynomial_batch(B, H, 0.1, seed=9, ref_length=R)
    torch.manual_seed(1)
    proto = (LSTM_NEW(15, H, 9, 4, conv=1) if mode == "LSTM" else
             Net(15, H, 9, 40 if mode == "concurrent" else 4, conv=1))
    losses = []
    finals = []
    # (the single-process concurrent step runs from a step plan by default -
    # _PlannedStep, no capture; plan False: its captured-graph form)
    for static, graph, split, plan in (
            (False, False, None, True), (True, False, None, True),
            (True, True, None, True), (True, True, True, True),
            (True, True, None, False)):
        F._STATIC_PLANES.entries.clear()
        t = make_trainer(TrainDrone, FlightmareDynamics(), cfg)
        t.net = type(proto)(15, H, 9, proto.fc_out.out_features, conv=1).to(dev)
        t.net.load_state_dict(proto.state_dict())
        t.optimizer_controller = momentum_sgd(t.net.parameters(), 1e-7)
        t.static_shard = static
        t.graph_steps = graph        # + the step replayed from a HIP graph
        t.split_graph = split        # ... as two graphs around the all-reduce slot
        t.plan_steps = plan
        # (graph or stream order is a MEASURED choice, tests/test_gpu_round5.py; here
        # the graph form itself is under test: pinned, not left to this box's timing)
        t.measure_launch_form = False
        if mode == "LSTM":           # (h0, c0): the default generator, re-seeded
            torch.cuda.manual_seed(77)
        s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
        normed = state_preprocessing(s0)
        out = []
        for step in range(4):
            if step == 2:            # the data changes in place
                s0[:, 6:9] += 0.25
                normed.copy_(state_preprocessing(s0))
            if mode == "concurrent":
                loss = t.train_concurrent_fused(normed, s0, in_ref, ref)
            else:
                loss = t.train_recurrent_model(None, s0, in_ref, ref)
            out.append(loss.item())
        losses.append(out)
        finals.append({k: v.clone() for k, v in t.net.state_dict().items()})
        assert (len(F._STATIC_PLANES.entries) > 0) == static
        assert (len(t._graphs) > 0) == graph
        for g in t._graphs.values():
            planned = getattr(g, "planned", False)
            assert planned == (mode == "concurrent" and plan and not split)
            assert (g.capture or planned) and g.split == bool(split)
    a, b, c, d_, e_ = losses
    assert abs(a[1] - a[2]) / abs(a[1]) > 1e-4       # the change matters
    for x, y in zip(a, b):
        assert abs(x - y) / abs(x) < 1e-6, (a, b)
    if mode != "LSTM":
        # the capture's warm-up steps leave no trace (parameters and momentum
        # are restored): graphed == eager step by step, incl. the re-capture
        # after the data changed
        for x, y in zip(a, c):
            assert abs(x - y) / abs(x) < 1e-6, (a, c)
        # the N > 1 form on one GPU - graph A, (empty) all-reduce slot, graph B,
        # the loss through the flat buffer's last element - is the eager step
        # BIT FOR BIT: losses and the weights after four updates
        assert d_ == b and d_ == c and d_ == e_, (b, c, d_, e_)
        for k in finals[1]:
            assert torch.equal(finals[3][k], finals[1][k]), k
            assert torch.equal(finals[3][k], finals[2][k]), k
            assert torch.equal(finals[3][k], finals[4][k]), k
    else:
        # fresh (h0, c0) ~ N(0, 1) every step: graphed steps draw through the
        # captured generator state, so only the statistics agree
        assert all(np.isfinite(c)) and abs(c[0] - a[0]) / abs(a[0]) < 0.2
    F._STATIC_PLANES.entries.clear()


def test_graphed_step_follows_lr_and_physics_changes(dev):
    """the captured kernels get the simulator's parameters, dt and
    the learning rate BY VALUE.  Changing any of them after the capture must
    re-capture (they are part of the signature): lr = 0 stops the weights, a
    new parameter struct gives the loss of an eager trainer built with it."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    B, H = 300, 10
    cfg = dict(QUAD_CFG, train_mode="concurrent", batch_size=B,
               learning_rate_controller=1e-7)
    d = synthetic.quad_polynomial_batch(B, H, 0.1, seed=19, ref_length=H)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    normed = state_preprocessing(s0)
    torch.manual_seed(2)
    proto = Net(15, H, 9, 40, conv=1)
    windy = {"translational_drag": [0.3, -0.2, 0.1]}

    def trainer(params, graph):
        F._STATIC_PLANES.entries.clear()
        t = make_trainer(TrainDrone, FlightmareDynamics(modified_params=params), cfg)
        t.net = Net(15, H, 9, 40, conv=1).to(dev)
        t.net.load_state_dict(proto.state_dict())
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-7,
                                                 momentum=0.9)
        t.static_shard, t.graph_steps = True, graph
        return t
    step = lambda t: t.train_concurrent_fused(normed, s0, in_ref, ref).item()
    t = trainer({}, True)
    l0 = step(t)
    first = t._graphs["concurrent"]
    step(t)
    assert t._graphs["concurrent"] is first          # replayed
    # lr -> 0 (momentum 0 too: the buffers must not move the weights either)
    t.optimizer_controller.param_groups[0].update(lr=0.0, momentum=0.0)
    before = {k: v.clone() for k, v in t.net.state_dict().items()}
    step(t)
    assert t._graphs["concurrent"] is not first      # re-captured
    for k, v in t.net.state_dict().items():
        assert torch.equal(v, before[k]), k
    # new physics on the SAME weights: the loss of an eager trainer with it
    second = t._graphs["concurrent"]
    t.train_dynamics = FlightmareDynamics(modified_params=windy)
    l_new = step(t)
    assert t._graphs["concurrent"] is not second
    e = trainer(windy, False)
    e.net.load_state_dict(before)
    l_eager = step(e)
    assert l_new == l_eager and abs(l_new - l0) / abs(l0) > 1e-4, (l0, l_new, l_eager)
    # a captured graph keeps the plane copies it reads alive past an eviction
    assert t._graphs["concurrent"].planes
    F._STATIC_PLANES.entries.clear()


@pytest.mark.parametrize("graph", [False, True])
def test_in_kernel_update_is_the_optimizers_step(dev, graph):
    """The concurrent step's second stage applies torch.optim.SGD's update
    itself (apg_quad_mlp_concurrent_train_step) when one process trains:
    parameters, momentum buffers and losses of four steps equal a trainer that
    calls optimizer.step() after the same kernels; the optimizer's state_dict
    is the one torch would hold; optimizers the kernel does not implement
    (nesterov, weight decay, Adam) keep optimizer.step()."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from apg_trajectory_tracking_amd.train_base import momentum_sgd
    B, H = 1500, 10
    cfg = dict(QUAD_CFG, train_mode="concurrent", batch_size=B)
    d = synthetic.quad_polynomial_batch(B, H, 0.1, seed=23, ref_length=H)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    normed = state_preprocessing(s0)
    torch.manual_seed(4)
    proto = Net(15, H, 9, 40, conv=1)

    def run(in_kernel, make_opt=None):
        F._STATIC_PLANES.entries.clear()
        t = make_trainer(TrainDrone, FlightmareDynamics(), cfg)
        t.net = Net(15, H, 9, 40, conv=1).to(dev)
        t.net.load_state_dict(proto.state_dict())
        t.optimizer_controller = (make_opt or (lambda ps: momentum_sgd(ps, 3e-6)))(
            t.net.parameters())
        t.static_shard, t.graph_steps, t.in_kernel_update = True, graph, in_kernel
        losses = [t.train_concurrent_fused(normed, s0, in_ref, ref).item()
                  for _ in range(4)]
        return t, losses
    a, la = run(True)
    b, lb = run(False)
    assert a._in_kernel_update() is not None and b._in_kernel_update() is None
    assert la == lb, (la, lb)
    assert la[3] < la[0]
    moved = 0
    # (the trainer's optimizer is torch's FUSED momentum SGD, whose arithmetic
    # - double, one rounding - the kernel repeats: the same bits)
    for (k, pa), (_, pb) in zip(a.net.named_parameters(), b.net.named_parameters()):
        assert torch.equal(pa, pb), (k, float((pa - pb).abs().max()), float(pb.abs().max()))
        if pa.grad is not None:
            assert torch.equal(pa.grad, pb.grad), k
            ma = a.optimizer_controller.state[pa]["momentum_buffer"]
            mb = b.optimizer_controller.state[pb]["momentum_buffer"]
            assert torch.equal(ma, mb), (k, float((ma - mb).abs().max()))
            moved += int(not torch.equal(pa.detach().cpu(), proto.state_dict()[k]))
    assert moved == 12
    # the state survives a round trip through the optimizer's own format
    sd = a.optimizer_controller.state_dict()
    fresh = momentum_sgd(a.net.parameters(), 3e-6)
    fresh.load_state_dict(sd)
    a.optimizer_controller = fresh
    l5 = a.train_concurrent_fused(normed, s0, in_ref, ref).item()
    l5b = b.train_concurrent_fused(normed, s0, in_ref, ref).item()
    assert l5 == l5b
    # anything but plain momentum SGD: the optimizer object steps
    for make in (lambda ps: torch.optim.SGD(ps, lr=3e-6, momentum=0.9, nesterov=True),
                 lambda ps: torch.optim.SGD(ps, lr=3e-6, momentum=0.9, weight_decay=1e-3),
                 lambda ps: torch.optim.SGD(ps, lr=3e-6),
                 lambda ps: torch.optim.Adam(ps, lr=1e-5)):
        c, lc = run(True, make)
        assert c._in_kernel_update() is None and lc[1] != lc[0]
    F._STATIC_PLANES.entries.clear()


@pytest.mark.parametrize("prefetch", [True, False])
@pytest.mark.parametrize("mode", ["concurrent", "autoregressive", "LSTM"])
def test_run_epoch_replays_one_graph_over_shuffled_minibatches(dev, mode, prefetch):
    """TrainBase.graph_steps with the REAL epoch loop: shuffled index batches
    (two full ones and a ragged tail per epoch).  `prefetch_batches` (opt-in):
    the batch's layout change + row gather runs one batch ahead on a side
    stream into one of two buffer sets, every minibatch replays the step
    captured for (its size, its buffer set); without it the index batch is
    copied into a persistent buffer the captured gather reads.  Same seed ->
    the epochs' losses and the final weights equal the eager loop's (which
    runs the same pipeline, or none), graphs are captured once.  From the
    second epoch on the whole epoch is ONE graph (`graph_epochs`): a fresh
    permutation is copied into the buffer its gathers read."""
    import copy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    cfg = dict(QUAD_CFG, train_mode=mode, epoch_size=1000, self_play=0,
               batch_size=384, learning_rate_controller=1e-6)
    runs = []
    for graph, pre in ((False, False), (False, prefetch), (True, prefetch)):
        torch.manual_seed(4)
        t = make_trainer(TrainDrone, FlightmareDynamics(), cfg)
        t.initialize_model(device=dev, seed=6)
        if runs:
            t.net.load_state_dict(runs[0][2])      # same start
        start = copy.deepcopy(t.net.state_dict())
        t.graph_steps, t.prefetch_batches = graph, pre
        # (one process, fp32 data set: the concurrent epoch names its batches by
        # rows - tests/test_gpu_round5.py; this is the gather pipeline that
        # N > 1 ranks, other optimizers and other data sets run)
        t.rows_in_kernel = False
        torch.manual_seed(11)                      # the permutations
        torch.cuda.manual_seed(12)
        losses = [t.run_epoch(train="controller", epoch=e) for e in range(3)]
        runs.append((losses, {k: v.clone() for k, v in t.net.state_dict().items()},
                     start))
        if graph:
            # the first epoch stepped through per-batch graphs, the second was
            # captured WHOLE (graph_epochs) and replayed, the third replayed
            (eg,) = t._epoch_graphs.values()
            assert eg["graph"] is not None and eg["last"] == 2
        # (the concurrent mode's epoch graph always pipelines: the next batch's
        # gather is forked behind the reverse kernel)
        if graph and (pre or mode == "concurrent"):
            # batch i of an epoch uses buffer set i & 1: 384, 384, 232
            assert sorted((k[1], k[3]) for k in t._graphs) == [(232, 0), (384, 0), (384, 1)]
            assert not t._index_bufs
            first = dict(t._graphs)
            t.run_epoch(train="controller", epoch=3)
            assert all(t._graphs[k] is g for k, g in first.items())   # no re-capture
        elif graph:
            assert sorted(k[1] for k in t._graphs) == [232, 384]
            assert sorted(t._index_bufs) == [232, 384]
        else:
            assert not t._graphs
    (la, wa, _), (lb, wb, _), (lc, wc, _) = runs
    assert all(np.isfinite(la)) and all(np.isfinite(lc))
    if mode != "LSTM":
        for l_, w_ in ((lb, wb), (lc, wc)):
            assert np.allclose(la, l_, rtol=1e-5), (la, l_)
            for k in wa:
                assert rel_err(w_[k].cpu().numpy(), wa[k].cpu().numpy()) < 1e-5, k
    else:      # fresh (h0, c0) per step from the captured generator state
        assert abs(lb[0] - la[0]) / abs(la[0]) < 1e-5     # eager: same draws
        assert abs(lc[0] - la[0]) / abs(la[0]) < 0.2
