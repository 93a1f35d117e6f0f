"""apg_linear_wgrad / apg_trajectory_tracking_amd.nn.Linear on the GPU: the
weight gradient of a PyTorch-side policy layer against torch's own (float64)
and the drop-in module against torch.nn.Linear through a training step."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,M,N,bias", [
    (1, 4, 8, True), (7, 64, 15, True), (300, 64, 224, True), (301, 40, 64, False),
    (65536, 64, 64, True), (65537, 80, 64, True), (4099, 200, 300, True),
    (131072, 64, 128, True)])
def test_wgrad_vs_float64(dev, B, M, N, bias):
    """dW = dY^T X and db = sum dY: one row, odd row counts (the pair of a last
    single row lies beyond the tensors), widths that are no multiple of 32,
    several 64 x 128 tiles, the benchmark batch."""
    from apg_trajectory_tracking_amd import nn as apg_nn
    g = torch.Generator().manual_seed(B + M)
    x = torch.randn(B, N, generator=g)
    w = torch.randn(M, N, generator=g) * 0.1
    b = torch.randn(M, generator=g) if bias else None
    dy = torch.randn(B, M, generator=g)
    xd = x.to(dev)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True) if bias else None
    y = apg_nn.linear(xd, wd, bd)
    assert isinstance(y.grad_fn, apg_nn._LinearWgrad._backward_cls)
    y.backward(dy.to(dev))
    want_w = dy.double().t() @ x.double()
    assert rel_err(wd.grad.cpu().numpy(), want_w.numpy()) < 1e-5
    if bias:
        assert rel_err(bd.grad.cpu().numpy(), dy.double().sum(0).numpy()) < 1e-5
    ref = torch.nn.functional.linear(x.double(), w.double(), None if b is None else b.double())
    assert rel_err(y.detach().cpu().numpy(), ref.numpy()) < 1e-5


def test_input_gradient_and_leading_dimensions(dev):
    """dL/dx is the plain product; inputs with more than two dimensions are
    flattened over their leading ones like torch.nn.Linear does."""
    from apg_trajectory_tracking_amd import nn as apg_nn
    torch.manual_seed(3)
    lin = apg_nn.Linear(9, 20).to(dev)
    ref = torch.nn.Linear(9, 20).to(dev)
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(6, 10, 9, device=dev, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    cot = torch.randn(6, 10, 20, device=dev)
    (lin(x) * cot).sum().backward()
    (ref(x2) * cot).sum().backward()
    assert rel_err(x.grad.cpu().numpy(), x2.grad.cpu().numpy()) < 1e-6
    assert rel_err(lin.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 1e-5
    assert rel_err(lin.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 1e-5
    with torch.no_grad():                      # no tape: plain F.linear
        assert lin(x).grad_fn is None
    cpu = apg_nn.Linear(9, 20)                 # CPU tensors: it IS torch.nn.Linear
    xc = torch.randn(5, 9)
    assert torch.equal(cpu(xc), torch.nn.functional.linear(xc, cpu.weight, cpu.bias))
    cpu(xc).sum().backward()
    assert cpu.weight.grad is not None


def test_policy_step_equals_torch_linear_layers(dev):
    """hutter_model.Net (the package's policy class, built on the drop-in
    Linear) through two optimizer steps of the row-layout training path
    against the same network with torch.nn.Linear layers."""
    import copy
    from apg_trajectory_tracking_amd.dataset import SyntheticQuadDataset
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    B, H = 1000, 10
    cfg = dict(delta_t=0.1, delta_t_train=0.1, epoch_size=B, self_play=0, batch_size=B,
               state_size=12, horizon=H, train_mode="concurrent", ref_dim=9,
               action_dim=4, learning_rate_controller=1e-6, system="quad",
               modified_params={})
    data = SyntheticQuadDataset(B, H, 0.1, seed=3, device=dev)
    torch.manual_seed(0)
    proto = Net(15, H, 9, 4 * H, conv=1)
    runs = []
    for plain in (False, True):
        net = copy.deepcopy(proto)
        if plain:
            for name, mod in list(net.named_children()):
                if isinstance(mod, torch.nn.Linear):
                    lin = torch.nn.Linear(mod.in_features, mod.out_features)
                    lin.load_state_dict(mod.state_dict())
                    setattr(net, name, lin)
        dyn = FlightmareDynamics()
        t = TrainDrone(dyn, dyn, dict(cfg))
        t.net = net.to(dev)
        t.state_data = data
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-6, momentum=0.9)
        rows = data.packed()
        losses = [t.train_controller_packed(data.normed_states, data.in_ref_states,
                                            *rows).item() for _ in range(2)]
        runs.append((losses, {k: v.clone() for k, v in t.net.state_dict().items()}))
    (la, wa), (lb, wb) = runs
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    for k in wa:
        assert rel_err(wa[k].cpu().numpy(), wb[k].cpu().numpy()) < 1e-5, k


def test_packed_step_replayed_from_a_graph_equals_eager(dev):
    """The arbitrary-policy step on a resident shard (static_shard +
    graph_steps): forward, autograd's backward through the drop-in Linear
    layers, the fused rollout and the SGD update are captured once and
    replayed - losses and weights equal the eager steps; a replaced network
    is not served by the old capture."""
    import copy
    from apg_trajectory_tracking_amd import nn as apg_nn
    from apg_trajectory_tracking_amd.dataset import SyntheticQuadDataset
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone

    class Policy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = apg_nn.Linear(15 + 90, 48), apg_nn.Linear(48, 40)

        def forward(self, state, ref):
            return self.b(torch.tanh(self.a(torch.cat((state, ref.flatten(1)), 1))))
    B, H = 700, 10
    cfg = dict(delta_t=0.1, delta_t_train=0.1, epoch_size=B, self_play=0, batch_size=B,
               state_size=12, horizon=H, train_mode="concurrent", ref_dim=9,
               action_dim=4, learning_rate_controller=1e-6, system="quad",
               modified_params={})
    data = SyntheticQuadDataset(B, H, 0.1, seed=8, device=dev)
    rows = data.packed()
    torch.manual_seed(1)
    proto = Policy()
    runs = []
    from apg_trajectory_tracking_amd.parallel import GradAllReducer
    for graph, split in ((False, None), (True, None), (True, True)):
        dyn = FlightmareDynamics()
        t = TrainDrone(dyn, dyn, dict(cfg))
        t.net = copy.deepcopy(proto).to(dev)
        t.state_data = data
        t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-6, momentum=0.9)
        t.grad_sync = GradAllReducer(t.net.parameters())
        t.static_shard, t.graph_steps, t.split_graph = True, graph, split
        losses = [t.train_controller_packed(data.normed_states, data.in_ref_states,
                                            *rows).item() for _ in range(3)]
        runs.append((losses, {k: v.clone() for k, v in t.net.state_dict().items()}))
        assert (len(t._graphs) == 1) == graph
        if graph:       # a new network: the old capture must not be replayed
            first = t._graphs["packed"]
            assert first.split == bool(split)
            t.net = copy.deepcopy(proto).to(dev)
            t.optimizer_controller = torch.optim.SGD(t.net.parameters(), lr=1e-6,
                                                     momentum=0.9)
            t.grad_sync = GradAllReducer(t.net.parameters())
            again = t.train_controller_packed(data.normed_states, data.in_ref_states,
                                              *rows).item()
            assert t._graphs["packed"] is not first
            assert abs(again - losses[0]) / losses[0] < 1e-6
    (la, wa), (lb, wb), (lc, wc) = runs
    assert np.allclose(la, lb, rtol=1e-6) and la[0] != la[2]
    for k in wa:
        assert rel_err(wb[k].cpu().numpy(), wa[k].cpu().numpy()) < 1e-6, k
    # the N > 1 form (graph A: forward + backward + bucket pack; empty
    # all-reduce slot; graph B: unpack + update) on one GPU: same numbers
    assert np.allclose(la, lc, rtol=1e-6)
    for k in wa:
        assert rel_err(wc[k].cpu().numpy(), wa[k].cpu().numpy()) < 1e-6, k


def test_unmodified_user_policy_gets_the_library_weight_gradient(dev):
    """a user's policy written with stock torch.nn.Linear layers
    takes the fast weight-gradient path without the caller doing anything -
    init_optimizer switches the layers in place (same Parameter objects, same
    state_dict keys), run_epoch's row-layout path then calls apg_linear_wgrad
    once per layer and step; `swap_linear = False` opts out.  Both give the
    same epoch."""
    import copy
    from apg_trajectory_tracking_amd import nn as apg_nn
    from apg_trajectory_tracking_amd.dataset import SyntheticQuadDataset
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.train_drone import TrainDrone

    class UserPolicy(torch.nn.Module):      # nothing of this package in here
        def __init__(self):
            super().__init__()
            self.inp = torch.nn.Linear(15 + 90, 64)
            self.mid = torch.nn.Linear(64, 32)
            self.out = torch.nn.Linear(32, 40)

        def forward(self, state, ref):
            x = torch.cat((state, ref.flatten(1)), 1)
            return self.out(torch.tanh(self.mid(torch.tanh(self.inp(x)))))
    B, H = 900, 10
    cfg = dict(delta_t=0.1, delta_t_train=0.1, epoch_size=B, self_play=0, batch_size=300,
               state_size=12, horizon=H, train_mode="concurrent", ref_dim=9,
               action_dim=4, learning_rate_controller=1e-6, system="quad",
               modified_params={})
    torch.manual_seed(5)
    proto = UserPolicy()
    calls = []
    real_check = apg_nn._capi.check
    runs = []
    try:
        apg_nn._capi.check = lambda st, name: (calls.append(name), real_check(st, name))[1]
        for swap in (True, False):
            dyn = FlightmareDynamics()
            t = TrainDrone(dyn, dyn, dict(cfg))
            t.net = copy.deepcopy(proto).to(dev)
            t.state_data = SyntheticQuadDataset(B, H, 0.1, seed=4, device=dev)
            t.swap_linear = swap
            t.graph_steps = False        # count the calls of every step
            keys, ids = list(t.net.state_dict()), [id(p) for p in t.net.parameters()]
            t.init_optimizer()
            assert list(t.net.state_dict()) == keys
            assert [id(p) for p in t.net.parameters()] == ids
            assert all((type(m) is apg_nn.Linear) == swap
                       for m in (t.net.inp, t.net.mid, t.net.out))
            del calls[:]
            torch.manual_seed(9)         # the permutation
            loss = t.run_epoch(train="controller", epoch=0)
            assert calls.count("apg_linear_wgrad") == (3 * 3 if swap else 0), calls
            runs.append((loss, {k: v.clone() for k, v in t.net.state_dict().items()}))
    finally:
        apg_nn._capi.check = real_check
    (la, wa), (lb, wb) = runs
    assert abs(la - lb) / abs(lb) < 1e-5
    for k in wa:
        assert rel_err(wa[k].cpu().numpy(), wb[k].cpu().numpy()) < 1e-5, k


def test_drop_in_linear_refuses_double_backward_and_falls_back_when_unsupported(dev):
    """the backward is once-differentiable (create_graph raises
    instead of returning silent garbage); operands the kernel is not built for
    take autograd's own formulas instead of raising."""
    from apg_trajectory_tracking_amd import nn as apg_nn
    torch.manual_seed(3)
    lin = apg_nn.Linear(7, 5).to(dev)
    x = torch.randn(33, 7, device=dev, requires_grad=True)
    g2 = torch.autograd.grad(lin(x).pow(2).sum(), lin.weight, create_graph=True)[0]
    with pytest.raises(RuntimeError):
        g2.sum().backward()
    # an operand the kernel is not built for (>= 4 GiB; here the limit is
    # lowered instead): autograd's own formulas, same numbers
    calls = []
    real_check = apg_nn._capi.check
    apg_nn._capi.check = lambda st, name: (calls.append(name), real_check(st, name))[1]
    limit = apg_nn._MAX_OPERAND_BYTES
    try:
        grads = []
        for lim in (limit, 64):
            apg_nn._MAX_OPERAND_BYTES = lim
            lin.zero_grad()
            lin(x.detach()).pow(2).sum().backward()
            grads.append((lin.weight.grad.clone(), lin.bias.grad.clone()))
        assert calls == ["apg_linear_wgrad"]          # the second pass fell back
    finally:
        apg_nn._MAX_OPERAND_BYTES, apg_nn._capi.check = limit, real_check
    assert rel_err(grads[1][0].cpu().numpy(), grads[0][0].cpu().numpy()) < 1e-6
    assert rel_err(grads[1][1].cpu().numpy(), grads[0][1].cpu().numpy()) < 1e-6
