"""LearntFixedWingDynamics on the GPU (csrc/wing_learnt.hip: the fixed-wing
step with live, trainable physical parameters and a general 3x3 inertia matrix,
and the cotangents of all 37 + 9 of them): against the recordings of the REAL
module (G16, tests/golden/make_golden.py) and against the float64 oracle that
the CPU suite pins to the same recordings."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


def weights(g, prefix="w."):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files
            if k.startswith(prefix)}


def module(g, dev, prefix="w."):
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        LearntFixedWingDynamics)
    dyn = LearntFixedWingDynamics()
    missing = dyn.load_state_dict(weights(g, prefix))   # the reference's names
    assert not missing.missing_keys and not missing.unexpected_keys
    return dyn.to(dev)


def target_of(g):
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    mp = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in g["target_mod"]}
    return FixedWingDynamics(modified_params=mp)


def test_forward_and_every_parameter_gradient_vs_reference(dev):
    g = load_golden("learnt_wing.npz")
    dyn = module(g, dev)
    state, action = (torch.from_numpy(g[k]).to(dev) for k in ("state", "action"))
    dt = float(g["dt"])
    nxt = dyn(state, action, dt)
    assert rel_err(nxt.detach().cpu().numpy(), g["next"]) < 1e-5
    with torch.no_grad():
        tgt = target_of(g)(state, action, dt)
    assert rel_err(tgt.cpu().numpy(), g["target_next"]) < 1e-5
    loss = torch.sum((nxt - tgt)**2)
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-4
    loss.backward()
    for k, p in dyn.named_parameters():
        if not bool(g["has_grad." + k]):
            assert p.grad is None or float(p.grad.abs().max()) == 0, k
            continue
        e = rel_err(p.grad.cpu().numpy(), g["g." + k])
        assert e < 1e-4, (k, e)
    # the parameters after the reference's own optimizer steps: a general I
    after = module(g, dev, "steps.w.")
    with torch.no_grad():
        assert rel_err(after(state, action, dt).cpu().numpy(), g["steps.next"]) < 1e-5


def test_four_optimizer_steps_vs_reference(dev):
    """The simulator fit itself: momentum SGD on every parameter; losses and
    final parameters as the reference's loop left them."""
    g = load_golden("learnt_wing.npz")
    dyn = module(g, dev)
    state, action = (torch.from_numpy(g[k]).to(dev) for k in ("state", "action"))
    dt = float(g["dt"])
    tgt = torch.from_numpy(g["target_next"]).to(dev)
    opt = torch.optim.SGD(dyn.parameters(), lr=float(g["steps.lr"]), momentum=0.9)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        l = torch.sum((dyn(state, action, dt) - tgt)**2)
        l.backward()
        opt.step()
        losses.append(l.item())
    assert np.allclose(losses, g["steps.loss"], rtol=2e-4)
    for k, v in dyn.state_dict().items():
        want = g["steps.w." + k]
        assert np.abs(v.cpu().numpy() - want).max() < 2e-4 * max(
            np.abs(want).max(), 1e-3), k
    with torch.no_grad():
        assert rel_err(dyn(state, action, dt).cpu().numpy(), g["steps.next"]) < 1e-4


@pytest.mark.parametrize("B,which", [(1, "w."), (1000, "w."), (1000, "steps.w."),
                                     (70000, "steps.w.")])
def test_gradients_vs_float64_oracle(dev, B, which):
    """Other batch sizes (one trajectory; ragged multi-wave; many blocks) and
    the general inertia matrix: dL/dstate, dL/daction and every parameter
    gradient against float64 autograd through the oracle."""
    from apg_trajectory_tracking_amd import synthetic
    from oracle import torch_port as tp
    g = load_golden("learnt_wing.npz")
    d = synthetic.wing_batch(B, 1, 0.05, seed=40 + B)
    gen = torch.Generator().manual_seed(B)
    state = d["state0"].clone()
    state[:, :3] = torch.randn(B, 3, generator=gen)
    state[:, 9:12] += 0.3 * torch.randn(B, 3, generator=gen)
    action = torch.rand(B, 4, generator=gen)
    cot = torch.randn(B, 12, generator=gen)
    dt = 0.05
    ora = tp.LearntWingOracle({k: v.numpy() for k, v in weights(g, which).items()})
    s64 = state.double().requires_grad_(True)
    a64 = action.double().requires_grad_(True)
    want_next = ora(s64, a64, dt)
    (want_next * cot.double()).sum().backward()
    dyn = module(g, dev, which)
    s = state.to(dev).requires_grad_(True)
    a = action.to(dev).requires_grad_(True)
    nxt = dyn(s, a, dt)
    (nxt * cot.to(dev)).sum().backward()
    assert rel_err(nxt.detach().cpu().numpy(), want_next.detach().numpy()) < 1e-5
    assert rel_err(s.grad.cpu().numpy(), s64.grad.numpy()) < 1e-4
    assert rel_err(a.grad.cpu().numpy(), a64.grad.numpy()) < 1e-4
    for k, p in dyn.named_parameters():
        want = ora.p[k].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, k
            continue
        e = rel_err(p.grad.cpu().numpy(), want.numpy())
        # sums of B signed float32 terms: allow their cancellation noise
        scale = 1e-4 if B <= 1000 else 5e-4
        assert e < scale, (k, e)


def test_trainer_with_a_learnt_wing_simulator(dev, tmp_path, monkeypatch):
    """TrainFixedWing with LearntFixedWingDynamics as train dynamics
    (scripts/train_fixed_wing.py:train_dynamics): the simulator fit moves the
    physical parameters towards the modified evaluation dynamics, and the
    controller phase unrolls through the module step by step - neither uses
    the fused analytic rollout."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics, LearntFixedWingDynamics)
    from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
    monkeypatch.chdir(tmp_path)
    cfg = dict(delta_t=0.05, delta_t_train=0.05, epoch_size=256, self_play=0,
               batch_size=64, state_size=12, horizon=20, ref_dim=3, action_dim=4,
               train_mode="concurrent", learning_rate_controller=1e-7,
               learning_rate_dynamics=2e-5, l2_lambda=0.01, system="wing",
               save_name="t", sample_in="train_env")
    learnt = LearntFixedWingDynamics().to(dev)
    t = TrainFixedWing(learnt, FixedWingDynamics({"mass": 1.2, "rho": 1.1}), cfg)
    torch.manual_seed(2)
    t.initialize_model(device=dev, seed=3)

    def forbidden(*a, **k):
        raise AssertionError("fused analytic rollout used with a learnt simulator")
    monkeypatch.setattr(F, "wing_rollout_loss", forbidden)
    monkeypatch.setattr(F, "wing_concurrent_policy_grads", forbidden)
    before = {k: v.clone() for k, v in learnt.state_dict().items()}
    first = t.run_epoch(train="dynamics")
    for _ in range(3):
        last = t.run_epoch(train="dynamics")
    assert np.isfinite(last) and last < first
    moved = [k for k, v in learnt.state_dict().items()
             if not torch.equal(v, before[k])]
    assert "I" in moved and "cfg.mass" in moved and "cfg.rho" in moved
    assert torch.equal(learnt.cfg["g"], before["cfg.g"])      # no gradient, :197
    w0 = [p.clone() for p in t.net.parameters()]
    loss = t.run_epoch(train="controller")
    assert np.isfinite(loss)
    assert any(not torch.equal(a, b) for a, b in zip(w0, t.net.parameters()))
