"""world_size-2 gloo test of the N > 1 path on CPU: batch sharding + ONE
all-reduce(sum) of the flattened policy gradient (parallel.GradAllReducer)
inside TrainBase._step reproduces the single-process step on the
concatenated batch (the reference losses are sums, so no rescaling).

No HIP kernel can run here, so the rollout of the trainer under test is
replaced by the CPU oracle (test infrastructure) - what is exercised is the
host logic: sharding, bucket packing, collective, SGD on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, rel_err

H, DT, B = 10, 0.1, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_trainer():
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from oracle import torch_port as tp

    class OracleBackedTrainDrone(TrainDrone):
        def train_controller_model(self, current_state, action_seq,
                                   in_ref_states, ref_states):
            self.optimizer_controller.zero_grad()
            inter = tp.unroll(tp.QuadOracle(), current_state, action_seq, DT)
            loss = tp.quad_mpc_loss(inter, ref_states, action_seq)
            return self._step(loss)

        def train_concurrent_fused(self, in_state, current_state, in_ref_states,
                                   ref_states):
            """Emulates the fused-policy path: parameter gradients arrive as
            contiguous views of one flat buffer with a trailing loss slot."""
            acts = torch.sigmoid(self.net(in_state, in_ref_states)).reshape(-1, H, 4)
            inter = tp.unroll(tp.QuadOracle(), current_state, acts, DT)
            loss = tp.quad_mpc_loss(inter, ref_states, acts)
            named = [(k, p) for k, p in self.net.named_parameters()]
            grads = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
            used = [(k, g) for (k, _), g in zip(named, grads) if g is not None]
            flat = torch.empty(sum(g.numel() for _, g in used) + 1)
            views, off = {}, 0
            for k, g in used:
                views[k] = flat[off:off + g.numel()].view_as(g)
                views[k].copy_(g)
                off += g.numel()
            return self._step_direct(loss.detach(), views, flat)

    cfg = dict(delta_t=DT, horizon=H, batch_size=B, ref_dim=9, action_dim=4,
               train_mode="concurrent", learning_rate_controller=1e-5,
               system="quad")
    trainer = OracleBackedTrainDrone(None, None, cfg)
    torch.manual_seed(11)
    trainer.net = Net(15, H, 9, 4 * H, conv=1)

    class Data:
        pass
    from apg_trajectory_tracking_amd import synthetic
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=21)
    Data.states, Data.ref_states, Data.in_ref_states = (
        d["state0"], d["ref"], d["in_ref"])
    Data.normed_states = tp.quad_state_features(d["state0"])
    trainer.state_data = Data
    trainer.shuffle = False
    trainer.init_optimizer()
    return trainer


def _one_step(trainer, lo, hi):
    d = trainer.state_data
    acts = torch.sigmoid(trainer.net(d.normed_states[lo:hi],
                                     d.in_ref_states[lo:hi]))
    return trainer.train_controller_model(
        d.states[lo:hi], acts.reshape(-1, H, 4), d.in_ref_states[lo:hi],
        d.ref_states[lo:hi])


def _one_step_direct(trainer, lo, hi):
    d = trainer.state_data
    return trainer.train_concurrent_fused(
        d.normed_states[lo:hi], d.states[lo:hi], d.in_ref_states[lo:hi],
        d.ref_states[lo:hi])


class _ToyLearntDynamics(torch.nn.Module):
    """Stand-in for LearntDynamics on CPU: learnable, with the two residual
    layers train_dynamics_model penalises."""
    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.linear_state_1 = torch.nn.Linear(16, 8)
        self.linear_state_2 = torch.nn.Linear(8, 12)

    def forward(self, state, action, dt):
        x = torch.cat((state, action), 1)
        return state + dt * self.linear_state_2(torch.tanh(self.linear_state_1(x)))


def _dynamics_steps(lo, hi):
    """Two train_dynamics_model steps on rows [lo, hi) of the data set."""
    from oracle import torch_port as tp
    trainer = _make_trainer()
    trainer.train_dynamics = _ToyLearntDynamics()
    trainer.eval_dynamics = tp.QuadOracle()
    trainer.l2_lambda = 0.1
    trainer.learning_rate_dynamics = 1e-3
    trainer.init_optimizer()
    d = trainer.state_data
    with torch.no_grad():
        acts = torch.sigmoid(trainer.net(d.normed_states[lo:hi],
                                         d.in_ref_states[lo:hi])).reshape(-1, H, 4)
    losses = [float(trainer.train_dynamics_model(d.states[lo:hi], acts))
              for _ in range(2)]
    return losses, {k: v.numpy() for k, v in
                    trainer.train_dynamics.state_dict().items()}


def _epoch_run(world_sharded):
    """Two epochs of the REAL run_epoch loop (shuffled loader) on a trainer
    whose global minibatch is 16 of the 48 trajectories."""
    trainer = _make_trainer()
    trainer.batch_size = 16
    trainer.shuffle = True
    trainer.shard_seed = 123
    with torch.no_grad():           # replicas start DIFFERENT on purpose
        for p in trainer.net.parameters():
            p.add_(0.01 * (dist.get_rank() if dist.is_initialized() else 0))
    trainer.init_optimizer()        # broadcast + sharded loader
    if not world_sharded:           # single process: same permutation stream
        trainer.trainloader.generator = torch.Generator().manual_seed(123)
    trainer.train_concurrent_fused = lambda *a, **k: (False if k.get("probe")
                                                      else None)
    losses = [trainer.run_epoch("controller", epoch=e) for e in range(2)]
    return losses, {k: v.numpy() for k, v in trainer.net.state_dict().items()}


def _ar_steps(lo, hi, graphed=False, split=None, new_buffers_on_rank=None):
    """The code path of `bench.py --mode ar` / BASELINE configs[2]: the REAL
    TrainDrone.train_recurrent_model (autoregressive, fused branch) ->
    TrainBase._step_direct -> one flat all-reduce + SGD.  Only the kernel
    call (functional.quad_mlp_rollout_grads) is replaced by the CPU oracle's
    autoregressive unroll, handing back the same (loss, views, flat) triple.
    `graphed`: through TrainBase._graphed / _GraphedStep's scheduling (part A
    -> all-reduce slot -> part B), emulated eagerly as there is no GPU here -
    with two ranks that is the split form the GPUs run."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from oracle import torch_port as tp

    def oracle_grads(net, state0, in_ref, ref, dt, params, weights=None,
                     index=None, static_inputs=False, update=None):
        assert update is None      # (CPU tensors: the optimizer object steps)
        _, _, loss = tp.quad_recurrent_unroll(net, tp.QuadOracle(), state0,
                                              in_ref, ref, H, dt)
        named = list(net.named_parameters())
        grads = torch.autograd.grad(loss, [p for _, p in named], allow_unused=True)
        used = [(k, g) for (k, _), g in zip(named, grads) if g is not None]
        flat = torch.empty(sum(g.numel() for _, g in used) + 1)
        views, off = {}, 0
        for k, g in used:
            views[k] = flat[off:off + g.numel()].view_as(g)
            views[k].copy_(g)
            off += g.numel()
        return loss.detach(), views, flat
    orig = F.quad_mlp_rollout_grads
    F.quad_mlp_rollout_grads = oracle_grads
    try:
        class Dyn:           # analytic simulator as far as the trainer can tell
            params = object()
        cfg = dict(delta_t=DT, horizon=H, batch_size=B, ref_dim=9, action_dim=4,
                   train_mode="autoregressive", learning_rate_controller=1e-5,
                   system="quad")
        t = TrainDrone(Dyn(), Dyn(), cfg)
        torch.manual_seed(13)
        t.net = Net(15, H, 9, 4, conv=1)
        d = synthetic.quad_polynomial_batch(B, H, DT, seed=22, ref_length=2 * H)

        class Data:
            states, in_ref_states, ref_states = d["state0"], d["in_ref"], d["ref"]
            normed_states = d["state0"]
        t.state_data = Data
        t.init_optimizer()
        assert t._fusable_mlp()
        shard = (Data.states[lo:hi], Data.in_ref_states[lo:hi], Data.ref_states[lo:hi])
        if graphed:
            t.static_shard, t.graph_steps, t.graph_emulation = True, True, True
            t.split_graph = split
        else:
            t.graph_steps = False
        losses = [float(t.train_recurrent_model(None, *shard)) for _ in range(2)]
        if graphed:      # one _GraphedStep, replayed; split iff it has the slot
            g, = t._graphs.values()
            assert g.split == (dist.is_initialized() or bool(split)) and not g.capture
        if new_buffers_on_rank is not None:
            # ONE rank's optimizer state is re-installed (new momentum
            # buffer objects) - every rank re-captures, so the warm-up collectives
            # of a real capture would pair up; the held buffers stay referenced
            opt = t.optimizer_controller
            assert all(any(b is h for h in g.keep)
                       for b in (st["momentum_buffer"] for st in opt.state.values()))
            if dist.get_rank() == new_buffers_on_rank:
                opt.load_state_dict(__import__("copy").deepcopy(opt.state_dict()))
            t._epoch_sigs = {}      # (a new epoch: where run_epoch takes signatures afresh)
            losses.append(float(t.train_recurrent_model(None, *shard)))
            t._epoch_sigs = None
            g2, = t._graphs.values()
            assert g2 is not g, "this rank replayed while another re-captured"
        return losses, {k: v.numpy() for k, v in t.net.state_dict().items()}
    finally:
        F.quad_mlp_rollout_grads = orig


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from apg_trajectory_tracking_amd.parallel import shard_range
        trainer = _make_trainer()
        lo, hi = shard_range(B)
        losses = [float(_one_step(trainer, lo, hi)) for _ in range(2)]
        sd = {k: v.numpy() for k, v in trainer.net.state_dict().items()}
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # the fused-policy path: flat gradient buffer all-reduced in place
        trainer = _make_trainer()
        losses = [float(_one_step_direct(trainer, lo, hi)) for _ in range(2)]
        sd = {k: v.numpy() for k, v in trainer.net.state_dict().items()}
        np.savez(os.path.join(out_dir, f"direct_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # the simulator fit (N3): its gradients are all-reduced too
        losses, sd = _dynamics_steps(lo, hi)
        np.savez(os.path.join(out_dir, f"dyn_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # bench.py --mode ar: autoregressive trainer step, flat all-reduce
        losses, sd = _ar_steps(lo, hi)
        np.savez(os.path.join(out_dir, f"ar_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # the same through the graphed step's scheduling (two parts around the
        # eager all-reduce: what `graph_steps` runs with more than one rank)
        losses, sd = _ar_steps(lo, hi, graphed=True)
        np.savez(os.path.join(out_dir, f"arg_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # the epoch loop itself: parameter broadcast, shared permutation,
        # per-rank slices of every global minibatch
        losses, sd = _epoch_run(True)
        np.savez(os.path.join(out_dir, f"epoch_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
        # host decisions every rank takes alike
        from apg_trajectory_tracking_amd import parallel
        assert parallel.any_rank(rank == 1) and not parallel.any_rank(False)
        # bench.py's `parallel_efficiency` leg: the step's collective switched off
        # and on again on a live group (any_rank is not part of it)
        msg = torch.full((3,), float(rank + 1))
        with parallel.collectives_suspended():
            parallel.reduce_sum(msg)
            assert msg.tolist() == [float(rank + 1)] * 3
            assert parallel.any_rank(rank == 0)
        parallel.reduce_sum(msg)
        assert msg.tolist() == [3.0] * 3
        losses, sd = _ar_steps(lo, hi, graphed=True, new_buffers_on_rank=1)
        np.savez(os.path.join(out_dir, f"recap_rank{rank}.npz"),
                 losses=np.array(losses), **sd)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_step_equals_single_process(tmp_path):
    ref = _make_trainer()
    ref_losses = [float(_one_step(ref, 0, B)) for _ in range(2)]
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world,
             join=True)
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_allclose(g["losses"], ref_losses, rtol=1e-5)
        for k, v in ref.net.state_dict().items():
            assert rel_err(g[k], v.numpy()) < 1e-5, (r, k)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ref.net.state_dict():
        assert np.array_equal(a[k], b[k]), k    # replicas stay bit-identical
    for r in range(world):                      # _step_direct, flat all-reduce
        g = np.load(tmp_path / f"direct_rank{r}.npz")
        np.testing.assert_allclose(g["losses"], ref_losses, rtol=1e-5)
        for k, v in ref.net.state_dict().items():
            assert rel_err(g[k], v.numpy()) < 1e-5, (r, k)


    ref_losses, ref_sd = _ar_steps(0, B)            # configs[2] code path
    for r in range(world):
        g = np.load(tmp_path / f"ar_rank{r}.npz")
        np.testing.assert_allclose(g["losses"], ref_losses, rtol=1e-5)
        for k, v in ref_sd.items():
            assert rel_err(g[k], v) < 1e-5, (r, k)

    for r in range(world):          # split-graph scheduling == the eager step
        g, e = np.load(tmp_path / f"arg_rank{r}.npz"), np.load(tmp_path / f"ar_rank{r}.npz")
        assert np.array_equal(g["losses"], e["losses"])
        for k in ref_sd:
            assert np.array_equal(g[k], e[k]), (r, k)

    a, b = (np.load(tmp_path / f"recap_rank{r}.npz") for r in range(2))
    assert np.array_equal(a["losses"], b["losses"]) and len(a["losses"]) == 3
    for k in ref_sd:                # ... and the re-captured step trained alike
        assert np.array_equal(a[k], b[k]), k

    ref_losses, ref_sd = _epoch_run(False)          # run_epoch, sharded loader
    for r in range(world):
        g = np.load(tmp_path / f"epoch_rank{r}.npz")
        np.testing.assert_allclose(g["losses"], ref_losses, rtol=1e-5)
        for k, v in ref_sd.items():
            assert rel_err(g[k], v) < 1e-5, (r, k)
    a, b = (np.load(tmp_path / f"epoch_rank{r}.npz") for r in range(2))
    for k in ref_sd:
        assert np.array_equal(a[k], b[k]), k

    ref_losses, ref_sd = _dynamics_steps(0, B)      # train_dynamics_model
    for r in range(world):
        g = np.load(tmp_path / f"dyn_rank{r}.npz")
        np.testing.assert_allclose(g["losses"], ref_losses, rtol=1e-5)
        for k, v in ref_sd.items():
            assert rel_err(g[k], v) < 1e-5, (r, k)


def test_graphed_step_scheduling_single_process():
    """One rank: the graphed step is ONE part-A + part-B sequence without a
    message; with `split_graph` forced it has the (empty) all-reduce slot and
    the loss travels in the flat buffer's last element - both equal the eager
    step bit for bit."""
    eager = _ar_steps(0, B)
    for split in (None, True):
        losses, sd = _ar_steps(0, B, graphed=True, split=split)
        assert losses == eager[0]
        for k, v in eager[1].items():
            assert np.array_equal(sd[k], v), k


def test_graph_signature_covers_values_captured_by_value():
    """a captured step bakes in the simulator's parameter struct,
    dt and the optimizer's learning rate - each of them must change the
    signature (so the step is re-captured, not replayed stale)."""
    import ctypes
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    from apg_trajectory_tracking_amd.models.hutter_model import Net

    class P(ctypes.Structure):
        _fields_ = [("mass", ctypes.c_float), ("g", ctypes.c_float)]

    class Dyn:
        params = P(1.0, 9.81)
    cfg = dict(delta_t=DT, horizon=H, batch_size=B, ref_dim=9, action_dim=4,
               train_mode="autoregressive", system="quad")
    t = TrainDrone(Dyn(), Dyn(), cfg)
    t.net = Net(15, H, 9, 4, conv=1)
    from apg_trajectory_tracking_amd.train_base import momentum_sgd
    t.optimizer_controller = momentum_sgd(t.net.parameters(), 1e-4)
    x = torch.zeros(4, 12)
    base = t._graph_signature((x,), ())
    assert t._graph_signature((x,), ()) == base
    t.train_dynamics.params.mass = 1.5
    s1 = t._graph_signature((x,), ())
    assert s1 != base
    t.delta_t = 0.05
    s2 = t._graph_signature((x,), ())
    assert s2 != s1
    t.optimizer_controller.param_groups[0]["lr"] = 3e-4
    s3 = t._graph_signature((x,), ())
    assert s3 != s2
    x.add_(1)                    # resident-shard captures follow the content
    assert t._graph_signature((x,), ()) != s3
    idx = torch.zeros(4, dtype=torch.int64)
    v0 = t._graph_signature((x,), (idx,))
    x.add_(1)                    # index-batch captures read the data set live
    assert t._graph_signature((x,), (idx,)) == v0
    # a parameter whose STORAGE moves (`p.data = ...` keeps the object): the
    # captured kernels hold the old address
    t.net.fc2.weight.data = t.net.fc2.weight.data.clone()
    v1 = t._graph_signature((x,), (idx,))
    assert v1 != v0
    # the fused paths pass the 12 parameter objects they read themselves
    from apg_trajectory_tracking_amd import functional as F
    objs = F.mlp_param_objects(t.net)
    assert len(objs) == 12 and all(isinstance(p, torch.nn.Parameter) for p in objs)
    assert objs[6] is t.net.fc2.weight
    w0 = t._graph_signature((x,), (idx,), objs)
    t.net.fc3.bias = torch.nn.Parameter(t.net.fc3.bias.detach().clone())
    assert t._graph_signature((x,), (idx,), F.mlp_param_objects(t.net)) != w0
    assert F.mlp_param_objects(torch.nn.Linear(3, 3)) == ()


def test_grad_allreducer_is_noop_single_process():
    from apg_trajectory_tracking_amd.parallel import GradAllReducer
    lin = torch.nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    g = lin.weight.grad.clone()
    loss = torch.tensor(2.5)
    assert GradAllReducer(lin.parameters()).sync(loss) is loss
    assert torch.equal(lin.weight.grad, g)


def test_tensor_batches_shards_partition_every_global_batch():
    """Every rank's slices of a global minibatch are disjoint, contiguous in
    the shared permutation and cover it; all ranks see the same batch count."""
    from apg_trajectory_tracking_amd.dataset import TensorBatches
    data = (torch.arange(103),)
    single = TensorBatches(data, 16, shuffle=True,
                           generator=torch.Generator().manual_seed(7))
    want = [b.tolist() for b in single.iter_indices()]
    world = 3
    parts = [list(TensorBatches(data, 16, shuffle=True, shard=(r, world),
                                shard_seed=7).iter_indices()) for r in range(world)]
    assert all(len(p) == len(want) for p in parts)
    for i, w in enumerate(want):
        got = sum((parts[r][i].tolist() for r in range(world)), [])
        assert got == w
    # __iter__ (materialised batches) agrees with iter_indices, unshuffled too
    for shuffle in (True, False):
        for r in range(world):
            a = TensorBatches(data, 16, shuffle=shuffle, shard=(r, world), shard_seed=3)
            b = TensorBatches(data, 16, shuffle=shuffle, shard=(r, world), shard_seed=3)
            for (x,), idx in zip(a, b.iter_indices()):
                assert x.tolist() == idx.tolist()


def test_bench_rank_logic_under_gloo_with_two_ranks():
    """`bench.py --gpus 2 --dry-run-cpu` under torch.distributed.run: the file's
    own rank plumbing (process group from the launcher's environment, replay
    agreement, barriers, max over ranks, the flat-buffer all-reduce, rank-0
    print) runs on two CPU ranks and yields exactly ONE contract line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(repo, "bench.py"),
         "--gpus", "2", "--dry-run-cpu", "--steps", "5", "--warmup", "2"],
        capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 5
    assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"]
    # rank 1 proposed one replay more than rank 0: the maximum was agreed on
    assert d["config"]["replays"] >= 2
    assert d["config"]["timed_steps"] == 5 * d["config"]["replays"]
    assert abs(d["value"] - 2 * 65536 * 10 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["allreduce_check"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None


def test_bench_gpus_2_as_a_plain_command_launches_itself():
    """`python3 bench.py --gpus 2 ... --dry-run-cpu` with NO
    launcher around it (the form the driver uses for N = 1) re-executes itself
    under torch.distributed.run, exits 0 and prints ONE line with n_gpus 2 that
    says what the process group saw (`rccl`: world, backend, the three
    all-reduce latencies, the sum checked) and ends in `steps_summary`."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run(
        [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "20",
         "--warmup", "5", "--dry-run-cpu"],
        capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["global_batch"] == 2 * 65536
    assert d["rccl"]["world"] == 2 and d["rccl"]["backend"] == "gloo"
    assert sorted(d["rccl"]["allreduce_us"]) == ["12342", "30390", "32730"]
    assert d["rccl"]["sums_ok"] is True
    assert list(d)[-1] == "steps_summary"
    assert d["steps_summary"]["rccl"]["world"] == 2
    assert len(json.dumps(d["steps_summary"])) <= 1200
    # N = 1 stays a single process with no group and no `rccl` key
    r1 = subprocess.run(
        [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "5",
         "--warmup", "2", "--dry-run-cpu"],
        capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and "rccl" not in d1
    # without a GPU the real bench says so (and never hangs in a rendezvous)
    r2 = subprocess.run(
        [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "5",
         "--warmup", "2"], capture_output=True, text=True, timeout=300, env=env)
    if not torch.cuda.is_available():
        assert r2.returncode != 0 and "MI355X" in r2.stderr


def test_launcher_world_and_gpus_flag_must_agree():
    """A launcher that started W ranks for `--gpus N != W` is an error on every
    rank (before any process group exists), not a silently different job."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2",
                        "--dry-run-cpu"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_suspended_collectives_skip_only_the_gradient_all_reduce():
    """parallel.collectives_suspended (bench.py's `parallel_efficiency` leg):
    inside it `reduce_sum` is a no-op, outside it is the all-reduce; with no
    process group both are no-ops."""
    from apg_trajectory_tracking_amd import parallel
    t = torch.ones(4)
    parallel.reduce_sum(t)
    parallel.reduce_sum(None)
    with parallel.collectives_suspended():
        assert parallel._suspended
        parallel.reduce_sum(t)
    assert not parallel._suspended and t.tolist() == [1.0] * 4


def test_launch_form_is_a_measured_choice_not_a_constant():
    """TrainBase.launch_form (round 5): whether a step is
    replayed from a graph or launched in stream order is measured once per
    train mode (tests/test_gpu_round5.py does that on the GPU); without a
    measurement every mode is graphable, a recorded or pinned "eager" turns
    graphs off for that mode only, `graph_steps = False` for all."""
    from apg_trajectory_tracking_amd.train_drone import TrainDrone

    class Dyn:
        params = None
    mk = lambda mode, bs: TrainDrone(Dyn(), Dyn(), dict(
        delta_t=DT, horizon=H, batch_size=bs, ref_dim=9, action_dim=4, train_mode=mode,
        system="quad"))
    for mode, bs in (("autoregressive", 65536), ("autoregressive", 8), ("LSTM", 65536),
                     ("concurrent", 65536)):
        t = mk(mode, bs)
        t.graph_emulation = True          # (no GPU here: the scheduling only)
        assert t._graphable(), (mode, bs)
        t.launch_form = {"autoregressive": "eager"}
        assert t._graphable() == (mode != "autoregressive")
        t.launch_form = {mode: "graph"}
        assert t._graphable()
        t.graph_steps = False
        assert not t._graphable()


def test_epoch_table_falls_through_to_what_applies_without_a_gpu():
    """run_epoch is a walk over ONE table.  Its first
    entry - the concurrent epoch that names its batches by rows - needs the
    kernels; on a CPU data set it does not apply and the walk goes on; the
    loader's order prefetch is a no-op off the GPU."""
    from apg_trajectory_tracking_amd.dataset import TensorBatches
    from apg_trajectory_tracking_amd.train_drone import TrainDrone

    class Dyn:
        params = None
    t = TrainDrone(Dyn(), Dyn(), dict(delta_t=DT, horizon=H, batch_size=8, ref_dim=9,
                                      action_dim=4, train_mode="concurrent", system="quad"))
    names = [name for name, _, _ in t._epoch_table()]
    assert "rows" in names[0] and names[-1] == "loader" and len(names) == 6
    ld = TensorBatches((torch.arange(20.).reshape(10, 2),), 4, shuffle=True)
    t.trainloader = ld
    assert not t.concurrent_rows_ok()
    assert not t._rows_ok(torch.zeros(4, 15), torch.zeros(4, 12), torch.zeros(4, 10, 9),
                          torch.zeros(4, 10, 9), torch.zeros(4, dtype=torch.int64))
    ld.prefetch_order(None)                      # CPU: nothing is drawn ahead
    assert not hasattr(ld, "_order_ahead")
    assert sorted(ld.epoch_order().tolist()) == list(range(10))
