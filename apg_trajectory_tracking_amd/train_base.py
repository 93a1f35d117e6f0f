"""Drop-in for scripts/train_base.py:30-375 (`TrainBase`) restricted to the
APG hot path: constructor keywords / derived dims (:32-128), init_optimizer
(:130-150), run_epoch (:188-218), sample_new_data (:220-231), plus a minimal
run_control loop.  `evaluate_model` is a hook: TrainDrone provides the batched
closed-loop evaluation (evaluate_drone.py); the speed
curriculum and plots are out of scope.

Differences that matter for speed, not for results:
  * minibatches come from whole device tensors (dataset.TensorBatches)
    instead of per-sample DataLoader collate (:132-137);
  * the running loss is accumulated on the device and read back ONCE per
    epoch instead of `loss.item()` per batch (:211);
  * with torch.distributed initialised (one process per GPU): init_optimizer
    broadcasts rank 0's policy (and learnable simulator) to every replica,
    the loader hands each rank its slice of every GLOBAL minibatch (same
    permutation on all ranks, dataset.TensorBatches), the gradients are
    all-reduced once per step (parallel.GradAllReducer / the flat buffer of
    the fused paths) and only rank 0 writes checkpoints and result files.
The quirk `epoch_loss = running_loss / i` (i = last batch INDEX, :213) is kept.
"""
import os
from collections import defaultdict

import numpy as np
import torch
import torch.optim as optim

from .dataset import TensorBatches
from . import parallel
from .parallel import GradAllReducer


def momentum_sgd(params, lr):
    """optim.SGD(params, lr, momentum=0.9) as the reference builds it
    (scripts/train_base.py:140-150); with every parameter on the GPU the fused
    implementation does the whole step in ONE launch instead of three
    (same arithmetic: buf = 0.9 buf + grad, p -= lr buf)."""
    params = list(params)
    fused = bool(params) and all(p.is_cuda for p in params)
    return optim.SGD(params, lr=lr, momentum=0.9, fused=fused)


def F_lstm_tables(net):
    """The resident LSTM operand tables of `net`, if a step has made them."""
    from . import functional as F
    reg = F._LSTM_TABLES
    return reg.get(net) if reg is not None else None


def _snapshot_optimizer_state(optimizer):
    """{parameter: {key: clone of a tensor | copy of a plain value}} of an
    optimizer's per-parameter state (before a graph capture's warm-up steps)."""
    import copy
    return {p: {k: (v.detach().clone() if torch.is_tensor(v) else copy.deepcopy(v))
                for k, v in st.items()}
            for p, st in optimizer.state.items()}


def _restore_optimizer_state(optimizer, before):
    """Put an optimizer's state back IN PLACE: tensors that existed are copied
    back into the very tensor objects (graphs and step plans hold them by
    address), tensors created since are zeroed (= the state every torch
    optimizer's first step starts from), plain values are restored or, when
    new, removed."""
    with torch.no_grad():
        for p, st in optimizer.state.items():
            old = before.get(p, {})
            for k in list(st.keys()):
                v = st[k]
                if torch.is_tensor(v):
                    if k in old and torch.is_tensor(old[k]) and old[k].shape == v.shape:
                        v.copy_(old[k])
                    else:
                        v.zero_()
                elif k in old:
                    st[k] = old[k]
                else:
                    del st[k]


def _zero_state_is_fresh_state(optimizer):
    """A capture's warm-up steps are undone by copying old state tensors back and
    ZEROING the ones the warm-up created (`_restore_optimizer_state`): that equals
    "never stepped" only where an optimizer's first step starts from zeros.  True
    for the optimizers where it does - SGD without dampening (buf = grad either
    way), Adam / AdamW / Adamax / RMSprop (zero moments, step 0); anything else
    (SGD with dampening: its first step sets buf = grad, a zero buffer gives
    (1 - dampening) grad; Adagrad's initial accumulator; NAdam's mu_product = 1;
    unknown classes) steps eagerly - no warm-up, nothing to undo.  """
    if optimizer is None:          # (no optimizer yet: nothing a warm-up could leave)
        return True
    if type(optimizer) is optim.SGD:
        return all(not g.get("dampening") or not g.get("momentum")
                   for g in optimizer.param_groups)
    return type(optimizer) in (optim.Adam, optim.AdamW, optim.Adamax, optim.RMSprop)


def _make_capturable(optimizer):
    """An optimizer whose step() is to be captured into a HIP graph must keep
    its step counters on the device (torch: `capturable=True`; Adam and its
    relatives default to host-side counters, whose capture fails).  Switches
    the flag on for every group that has one and moves already existing
    counters over; SGD has no such flag and is left alone."""
    changed = False
    for g in optimizer.param_groups:
        if g.get("capturable") is False and all(p.is_cuda for p in g["params"]):
            g["capturable"] = True
            changed = True
            for p in g["params"]:
                st = optimizer.state.get(p)
                if st and torch.is_tensor(st.get("step")) and not st["step"].is_cuda:
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
    return changed


class _GraphedStep:
    """One optimizer step captured into HIP graphs and replayed: the ~12
    launches of a step cost the host ~0.25 ms of Python and launch overhead,
    more than the GPU needs for the concurrent step.

    The step is cut at the slot of the data-parallel all-reduce:
        part A  everything up to the gradient message (sweeps, weight-gradient
                products, loss into the message's last slot; or forward +
                autograd's backward + the bucket pack)
        slot    all-reduce(sum) of the message - ALWAYS an eager call (a
                captured RCCL collective that misbehaves hangs instead of
                failing); nothing at all with one rank
        part B  (bucket unpack,) the fused SGD update, the step's loss
    One rank: A and B are captured into ONE graph (`split` False).  More than
    one rank - or `TrainBase.split_graph` forced, which is how a single GPU
    tests this scheduling - : two graphs that share a memory pool, the
    collective between their replays.  So the N > 1 step replays the same
    kernels from graphs as the N = 1 step does; only the collective is added.

    Valid while the step's input tensors are the same objects (for resident
    shards: with the same in-place version), the parameters / optimizer are
    the same objects and the values the captured kernels got BY VALUE are
    unchanged - the simulator's parameter struct, dt, the optimizer's
    hyper-parameters (TrainBase._graph_signature re-captures otherwise).
    Allocations made during capture live in the graph's private pool, so the
    gradient views the optimizer reads keep their addresses across replays;
    the plane-layout copies of a resident shard that the captured kernels read
    are referenced from here (functional._StaticPlanes may evict them).

    `capture` False (no GPU; tests/test_distributed_cpu.py): the same
    A -> slot -> B scheduling, run eagerly."""

    def __init__(self, part_a, part_b, reduce, signature, keep, net, optimizer,
                 capture=True, split=False):
        self.signature, self.keep = signature, keep
        self.reduce, self.capture, self.split = reduce, capture, split
        if not capture:
            self.part_a, self.part_b = part_a, part_b
            return
        # (a captured step does NOT keep the closures: they reference the
        # trainer, and trainer -> graphs -> closures -> trainer would leave the
        # graphs to the cyclic collector, which may then destroy a HIP graph
        # in the middle of somebody else's capture)
        import gc

        def eager():
            msg = part_a()
            reduce(msg)
            return part_b(msg)
        # the warm-up steps (allocator, optimizer state, lazy inits, plane
        # caches - all outside the capture) must not train: the parameters and
        # the WHOLE optimizer state are put back afterwards, in place (the
        # captured kernels and the in-kernel update hold these tensors by
        # address).  State a warm-up step created is set to zero, which is what
        # every torch optimizer's first step starts from (SGD: a missing
        # momentum buffer is a zero one, buf = grad = 0.9 * 0 + grad; Adam:
        # step = 0, exp_avg = exp_avg_sq = 0) - so a user-supplied optimizer
        # does not start from two phantom steps' moments either.
        params = [p for p in net.parameters()]
        self.params = [p.detach() for p in params]
        saved = [p.detach().clone() for p in params]
        state_before = _snapshot_optimizer_state(optimizer)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(2):
                    eager()
        finally:
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                _restore_optimizer_state(optimizer, state_before)
        # with a process group alive its watchdog thread polls events; only
        # THIS thread's calls may invalidate the capture
        mode = ({"capture_error_mode": "thread_local"}
                if parallel.group_live() else {})
        # no garbage collection inside a capture: a collected tensor or graph
        # of somebody else would issue HIP calls the capture forbids
        was_enabled = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            if split:
                self.graph_a = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a, **mode):
                    self.msg = part_a()
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), **mode):
                    self.out = part_b(self.msg)
            else:
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, **mode):
                    self.out = part_b(part_a())
        finally:
            if was_enabled:
                gc.enable()
        # the gradient tensors the captured step writes (its `.grad`s): attached
        # again after every replay - zero_grad(), or an eager step in between,
        # would otherwise leave `.grad` pointing somewhere else (or nowhere)
        self.grads = [(p, p.grad) for p in params]
        from . import functional
        self.planes = functional.static_plane_refs()

    def _eager(self):         # emulation (capture False) only
        msg = self.part_a()
        self.reduce(msg)
        return self.part_b(msg)

    def __call__(self, borrow=False):
        """`borrow`: hand out the captured output tensor itself (valid until the
        next replay; run_epoch adds it to its running loss at once) instead of
        a private copy - one small launch less per step."""
        if not self.capture:
            return self._eager()
        if self.split:
            self.graph_a.replay()
            self.reduce(self.msg)
            self.graph_b.replay()
        else:
            self.graph.replay()
        # (the replay has written the parameters - optimizer or in-kernel update -
        # without running Python: bump their version counters here)
        torch.autograd.graph.increment_version(self.params)
        for p, g in self.grads:
            p.grad = g
        # a private copy: the next replay overwrites the captured output
        if borrow or not torch.is_tensor(self.out):
            return self.out
        return self.out.clone()


class _PlannedStep:
    """A fused step whose buffers, structs and arguments were made once
    (functional.QuadConcurrentStepPlan): calling it is `before()` - the batch's
    gather, if any - and ONE call into the library.  Lives in the trainer's
    `_graphs` table under the key a captured graph of the same step would
    have, with the same life cycle (rebuilt when the signature changes); it is
    what a single-process concurrent step uses instead of a graph - nothing
    is captured, the ~20 us the call costs the host keep the GPU busy, and the
    idle gap between two graph replays is gone."""
    capture, split, planned = False, False, True

    def __init__(self, plan, net, before=None):
        self.plan, self.before, self.signature = plan, before, None
        self.grads = [(p, plan.named.get(name)) for name, p in net.named_parameters()]
        from . import functional
        self.planes = functional.static_plane_refs()

    def __call__(self, borrow=False, events=None, index=None):
        if self.before is not None:
            self.before()
        loss = (self.plan.launch(events) if index is None
                else self.plan.launch(events, index=index))
        for p, g in self.grads:
            p.grad = g
        return loss      # (a tensor of this launch's own: nothing to clone)


class _EpochLoss:
    """The running loss of an epoch, kept on the device.  `fresh`: every step
    hands out a loss tensor of its own (eager launches - also the ones being
    captured into an epoch graph): the tensors are collected and summed by ONE
    launch at the end of the epoch.  Otherwise the loss is the replayed step
    graph's output buffer, overwritten by the next replay: it is added to the
    running sum at once."""

    def __init__(self, fresh):
        self.fresh, self.items, self.running = fresh, [], None

    def add(self, loss):
        if self.fresh:
            self.items.append(loss.reshape(()))
        elif self.running is None:
            self.running = loss.clone()
        else:
            self.running.add_(loss)

    def total(self):
        if self.fresh and self.items:
            return torch.stack(self.items).sum(dtype=torch.float64)
        return self.running


class _NullWriter:
    """Stand-in for the reference's `self.writer` (scripts/train_base.py:8-22,
    116: a tensorboard SummaryWriter, or a shim whose `add_scalar` does nothing
    when tensorboard is missing).  Scripts written against the reference call
    `trainer.writer.add_scalar(...)` (scripts/train_drone.py:198); nothing is
    logged here - the per-step losses are in `results_dict`."""

    def add_scalar(self, *args, **kwargs):
        return None

    add_scalars = add_histogram = add_text = flush = close = add_scalar


# The config schema of scripts/train_base.py:32-64 - keyword -> default.  Every
# key becomes an attribute of the trainer (except `system` / `save_name`, which
# only name the output directory); unknown keys are accepted and ignored, as
# the reference's **kwargs does (its JSON configs carry extra entries).
_CONFIG_DEFAULTS = dict(
    sample_in="train_env", delta_t=0.05, delta_t_train=0.05, epoch_size=500,
    vec_std=0.15, self_play=1.5, self_play_every_x=2, batch_size=8,
    reset_strength=1.2, max_drone_dist=0.25, max_steps=1000,
    thresh_div_start=4, thresh_div_end=20, thresh_stable_start=.4,
    thresh_stable_end=.8, state_size=12, horizon=10, ref_dim=3, action_dim=4,
    l2_lambda=0.1, learning_rate_controller=0.0001, learning_rate_dynamics=0.001,
    speed_factor=.6, resample_every=3, suc_up_down=1, train_mode="concurrent",
    system="quad", save_name="test_model")


class TrainBase:

    # default of `measure_launch_form` for new trainers (the test suite pins it to
    # False: whether a step replays a graph must not depend on the box's timing
    # where the graph form itself is under test - tests/conftest.py)
    MEASURE_LAUNCH_FORM = True

    def __init__(self, train_dynamics, eval_dynamics, **config):
        cfg = {k: config.get(k, d) for k, d in _CONFIG_DEFAULTS.items()}
        system, save_name = cfg.pop("system"), cfg.pop("save_name")
        for key, value in cfg.items():
            setattr(self, key, value)
        suc_up_down = self.suc_up_down

        self.results_dict = defaultdict(list)
        self.results_dict["loss"].append(0)

        self.save_name = save_name
        self.save_path = os.path.join("trained_models", system, save_name)
        self.save_model_name = "model_" + system

        self.eval_dynamics = eval_dynamics
        self.train_dynamics = train_dynamics

        self.count_finetune_data = 0
        self.sampled_data_count = 0
        self.current_score = 0 if suc_up_down == 1 else np.inf

        self.writer = _NullWriter()
        self.state_data = None
        self.net = None
        self.trainloader = None
        self.optimizer_controller = None
        self.grad_sync = None
        self.shuffle = True
        # True (default since round 4): the fused steps of run_epoch (index
        # batches) and steps on unchanged resident tensors (static_shard) are
        # replayed from captured HIP graphs (_GraphedStep: with more than one
        # rank as two graphs around the eager all-reduce).  The simulator's
        # parameters, dt and the optimizer's hyper-parameters are part of the
        # capture's signature: changing them re-captures, nothing goes stale.
        self.graph_steps = True
        # None: two graphs around the all-reduce slot iff world > 1; True
        # forces that form on one rank (tests, bench.py's like-for-like number)
        self.split_graph = None
        # True: a step plan's operand tables stay resident between steps (no pack
        # launch); False: packed at every step - for callers that write the
        # parameters through `.data` between steps without calling
        # `plan.invalidate()` ("resident operand tables")
        self.resident_tables = True
        # True: run the graphed scheduling without a GPU (parts executed
        # eagerly; the gloo tests of the N > 1 step)
        self.graph_emulation = False
        self._graphs = {}
        self._index_bufs = {}
        # True: run_epoch issues the layout change + row gather of batch i + 1
        # on a side stream while batch i steps (_pipelined_epoch).  Off by
        # default: measured on the MI355X at B = 65 536 it does not pay - the
        # overlapped gather slows the step's memory-bound kernels by as much as
        # it costs alone (concurrent 0.240 vs 0.238 ms per batch, LSTM 0.565 vs
        # 0.525, autoregressive 1.164 vs 1.143; profiles/r04_run_epoch.jsonl)
        self.prefetch_batches = False
        self._prefetch = {}
        # True (with graph_steps, one process): a whole EPOCH of the fused index
        # loops is captured into one HIP graph - per batch: gather, step,
        # update, running loss - and replayed per epoch with a fresh
        # permutation copied into the buffer its gathers read.  The host then
        # issues one launch per epoch instead of ~10 calls per batch (which,
        # at ~0.21 ms per batch, had become the bound of run_epoch).
        self.graph_epochs = True
        self._epoch_graphs = {}
        # Graph replay or stream-order launches?  For steps whose kernels
        # outlast the host's launch work the answer is a 2-5 % effect that
        # differs from box to box (round 4: the autoregressive step 1.111 ms in
        # stream order against 1.054 from one graph on the driver's box, the
        # other way round on the builder's; LSTM 0.480 eager against 0.494
        # graphed) - so it is MEASURED, once per train mode: the first time a
        # step is captured, `launch_form_steps` replays are timed against as
        # many stream-order steps (behind 30 ms of untimed replays, alternately,
        # twice; the training they do is undone), the faster form is kept and
        # recorded in results_dict["launch_form"].
        # `launch_form[train_mode]` = "graph" | "eager" pins the answer.
        self.launch_form = {}
        self.measure_launch_form = type(self).MEASURE_LAUNCH_FORM
        self.launch_form_steps = 10
        # True: a graphed step returns the captured loss buffer itself - valid
        # until the NEXT step overwrites it (run_epoch's loops take it that
        # way) - instead of a private copy, which is one more launch behind
        # every replay (~14 us per step at the concurrent step's size)
        self.borrow_loss = False
        # which of the step's events the next batch's gather is issued behind
        # (_pipelined_epoch, late fork): "after_forward" | "after_reverse" |
        # "before" (the step's first launch).  Measured inside the epoch graph
        # at 32 batches of 65 536 (tools/time_run_epoch.py, three runs each):
        # 0.174-0.176 | 0.179-0.181 | 0.192-0.194 ms per batch - behind the
        # forward kernel the gather trickles through the reverse kernel's 69 us
        # without slowing it and is nearly done when the second stage starts
        self.gather_fork = "after_forward"
        # one process, plain momentum SGD: the concurrent step applies the
        # optimizer's update inside its second-stage kernel (_in_kernel_update)
        self.in_kernel_update = True
        # ... and runs from a step plan instead of a captured graph: buffers
        # and argument structs made once, one library call per step
        # (_PlannedStep; needs graph_steps and the in-kernel update)
        self.plan_steps = True
        self._in_epoch_capture = False

        # horizon / reference-window length (scripts/train_base.py:118-128)
        if self.train_mode in ["autoregressive", "LSTM"]:
            self.actions_out_dim = self.action_dim
            self.ref_length = self.horizon * 2
        elif self.train_mode == "concurrent":
            self.actions_out_dim = self.action_dim * self.horizon
            self.ref_length = self.horizon
        else:
            raise ValueError(
                "Train mode must be one of concurrent, autoregressive, or LSTM"
            )

    # ------------------------------------------------------------ set-up
    def dataset_tensors(self):
        """The 4-tuple of whole-dataset tensors a minibatch is cut from
        (neural_control/dataset.py:125-132)."""
        d = self.state_data
        return (d.normed_states, d.states, d.in_ref_states, d.ref_states)

    def init_optimizer(self):
        """scripts/train_base.py:130-150: shuffled minibatch loader and
        SGD(lr, momentum 0.9) over the policy."""
        self.trainloader = TensorBatches(
            self.dataset_tensors(), self.batch_size, shuffle=self.shuffle,
            shard=(parallel.rank(), parallel.world_size()),
            shard_seed=getattr(self, "shard_seed", 0))
        # an arbitrary user policy gets the library's weight gradient: plain
        # torch.nn.Linear layers become nn.Linear in place (dW = dY^T X over
        # 65 536 rows: 15-70 us instead of rocBLAS' 155-225 us per layer);
        # state_dict keys and parameter objects are unchanged
        if getattr(self, "swap_linear", True) and isinstance(self.net, torch.nn.Module):
            from .nn import use_apg_linear
            use_apg_linear(self.net)
        # replicas must start equal: the sum all-reduce keeps them equal
        parallel.broadcast_module(self.net)
        if isinstance(self.train_dynamics, torch.nn.Module):
            parallel.broadcast_module(self.train_dynamics)
        self.optimizer_controller = momentum_sgd(
            self.net.parameters(), self.learning_rate_controller)
        self.grad_sync = GradAllReducer(self.net.parameters())
        self._epoch_runners = self._epoch_table()
        # a learnable simulator gets its own optimizer (:144-150)
        if isinstance(self.train_dynamics, torch.nn.Module):
            self.optimizer_dynamics = momentum_sgd(
                self.train_dynamics.parameters(), self.learning_rate_dynamics)
            self.grad_sync_dynamics = GradAllReducer(
                self.train_dynamics.parameters())

    def _step(self, loss):
        """backward -> (all-reduce) -> SGD step; returns the (global) loss."""
        loss.backward()
        if self.grad_sync is not None:
            loss = self.grad_sync.sync(loss.detach())
        self.optimizer_controller.step()
        return loss

    def _in_kernel_update(self, available=True, names=None, tensors=None):
        """(lr, momentum, {name: momentum buffer}) when the fused step may apply
        the optimizer's update itself (`in_kernel_update`, one process, the
        optimizer is plain momentum SGD as init_optimizer builds it, float32
        contiguous parameters) - else None: the step is followed by
        optimizer.step().  Missing momentum buffers are created as zeros, which
        is what SGD's first step assumes (buf = grad = momentum * 0 + grad)."""
        from . import functional as F
        opt = self.optimizer_controller
        if (not available or not getattr(self, "in_kernel_update", True)
                or self._reducing() or type(opt) is not optim.SGD
                or len(opt.param_groups) != 1):
            return None
        # a user's step hooks run around optimizer.step(): with hooks registered
        # (on this optimizer or globally) the step is left to the optimizer
        if self._step_hooks():
            return None
        g = opt.param_groups[0]
        if (not g["momentum"] or g["dampening"] or g["nesterov"] or g["weight_decay"]
                or g.get("maximize")):
            return None
        # (called once per step: the walk over the network and the optimizer's
        # state is repeated only when one of them is a different object -
        # load_state_dict installs a new state mapping - or the policy tensors,
        # the buffers or the settings the answer depends on have changed)
        # (names / tensors: another policy's parameter list - the LSTM's)
        names = F._MLP_PARAMS if names is None else names
        tensors = F.mlp_param_objects(self.net) if tensors is None else tensors
        fast = (id(opt), id(opt.state), g["lr"], g["momentum"]) + tuple(map(id, tensors))
        hit = getattr(self, "_iku", None)
        if hit is not None and hit[0] == fast and all(
                p.requires_grad and opt.state[p].get("momentum_buffer") is b
                for p, b in zip(tensors, hit[1][2].values())):
            # (an LR scheduler asks whether the optimizer has stepped: it has,
            # inside the kernel - said only where the kernel really steps)
            opt._opt_called = True
            return hit[1]
        self._iku = None
        named = dict(self.net.named_parameters())
        listed = {id(p) for p in g["params"]}
        bufs = {}
        for name in names:
            p = named.get(name)
            if (p is None or id(p) not in listed or not p.is_cuda or not p.requires_grad
                    or p.dtype != torch.float32 or not p.is_contiguous()):
                return None
            st = opt.state[p]
            if st.get("momentum_buffer") is None:
                st["momentum_buffer"] = torch.zeros_like(p)
            bufs[name] = st["momentum_buffer"]
        out = float(g["lr"]), float(g["momentum"]), bufs
        if len(tensors) == len(bufs):
            self._iku = (fast, out)
        opt._opt_called = True
        return out

    def _step_direct(self, loss, named_grads, flat=None, stepped=False):
        """_step for the fused-policy paths: the kernels already produced the
        parameter gradients (contiguous views of one flat buffer), so they are
        attached as `.grad` directly - no autograd tape, no per-parameter
        clone - then (all-reduce) -> optimizer step.  With more than one rank
        the flat buffer (gradients + loss in its last slot) is the message of
        the ONE all-reduce: nothing is packed or unpacked."""
        for name, p in self.net.named_parameters():
            p.grad = named_grads.get(name)
        if parallel.world_size() > 1:
            if flat is not None:
                flat[-1] = loss.detach().reshape(())
                parallel.reduce_sum(flat)
                loss = flat[-1].clone()
            elif self.grad_sync is not None:
                loss = self.grad_sync.sync(loss.detach())
        if not stepped:      # (stepped: the kernels have applied the update)
            self.optimizer_controller.step()
        return loss

    def _held_state(self):
        """The optimizer-state tensors a captured step or a plan addresses."""
        return [v for st in self.optimizer_controller.state.values()
                for v in st.values() if torch.is_tensor(v)]

    def _step_hooks(self):
        """The optimizer has step hooks (its own or global ones)."""
        from torch.optim import optimizer as _opt_mod
        opt = self.optimizer_controller
        return bool(getattr(opt, "_optimizer_step_pre_hooks", None)
                    or getattr(opt, "_optimizer_step_post_hooks", None)
                    or getattr(_opt_mod, "_global_optimizer_pre_hooks", None)
                    or getattr(_opt_mod, "_global_optimizer_post_hooks", None))

    def _graphable(self):
        # (inside the capture of a whole epoch the steps run "eagerly": their
        # launches are what is being captured)
        if not (bool(self.graph_steps) and not self._in_epoch_capture
                and (torch.cuda.is_available() or self.graph_emulation)):
            return False
        # (a step whose kernels outlast the host's launch work may run faster in
        # stream order: measured once, see launch_form.  Asked first: a step in
        # stream order calls this three times, and the checks below walk the
        # optimizer's state)
        if self.launch_form.get(self.train_mode) == "eager":
            return False
        # (a replayed graph runs no Python: step hooks would be skipped, and a
        # capture's warm-up steps would call them for steps that are undone)
        if self._step_hooks():
            return False
        # (the warm-up steps of a capture are undone by zeroing the state they
        # created: only where zero state IS fresh state)
        return _zero_state_is_fresh_state(self.optimizer_controller)

    def _reducing(self):
        """The step has an all-reduce slot (real or, when forced, empty)."""
        return parallel.world_size() > 1 or bool(self.split_graph)

    @staticmethod
    def _reduce(msg):
        """The all-reduce slot of a step.  With a live process group the
        collective is issued even for a world of one (a forced split step on
        one GPU then runs graph A -> RCCL -> graph B exactly as N ranks do:
        communicator, RCCL's stream and events, the watchdog - everything but
        the wire)."""
        parallel.reduce_sum(msg)

    def _graph_signature(self, inputs, volatile, params=None):
        """What a captured step is tied to.  Resident-shard captures (no
        `volatile` index buffer) also depend on the CONTENT of their inputs
        (kept plane copies): the in-place version counters are part of it.
        `params`: the parameter tensors the step reads, when the caller knows
        them (the fused paths: 12 attribute reads instead of a walk over the
        module tree, which costs more than the rest of the signature)."""
        import ctypes
        dyn = self.train_dynamics
        phys = getattr(dyn, "params", None)
        phys = bytes(phys) if isinstance(phys, ctypes.Structure) else id(dyn)
        opt = self.optimizer_controller
        hyper = tuple(tuple(sorted((k, v) for k, v in g.items()
                                   if isinstance(v, (int, float, bool))))
                      for g in opt.param_groups)
        versions = not volatile
        return (tuple((id(t), t._version if versions else 0, tuple(t.shape))
                      for t in inputs)
                # (volatile buffers by ADDRESS: views of one persistent buffer
                # are new tensor objects every time they are taken)
                + tuple((t.data_ptr(), tuple(t.shape)) for t in volatile)
                # a replaced network or optimizer must not replay the old graph
                # (`p.data = other` keeps the Parameter object and moves its
                # storage: the address is part of it)
                + tuple((id(p), p.data_ptr())
                        for p in (self.net.parameters() if params is None else params))
                # (load_state_dict replaces the momentum buffers; graphs and plans
                # hold them by ADDRESS and keep a reference - `_held_state` - so
                # a freed buffer's id cannot come back under a new one: ADVICE r4)
                + tuple((id(b), b.data_ptr() if torch.is_tensor(b) else 0)
                        for b in (st.get("momentum_buffer") for st in opt.state.values()))
                + (id(opt), hyper, phys, float(self.delta_t),
                   float(self.delta_t_train), self._reducing()))

    def _graphed(self, key, inputs, parts, volatile=()):
        """Run the step `parts` = (part_a, part_b) (see _GraphedStep) - through
        captured graphs when `graph_steps` is on and the capture's signature
        still holds.  `volatile`: tensors the captured kernels read whose
        CONTENT may differ from replay to replay (same object, same shape) -
        the index batch."""
        part_a, part_b = parts
        if not self._graphable():
            msg = part_a()
            self._reduce(msg)
            return part_b(msg)
        # inside an epoch loop the signature of a key is computed once (~30 us
        # of host time per step otherwise: parameter ids, optimizer settings)
        cache = getattr(self, "_epoch_sigs", None)
        sig = cache.get(key) if cache is not None else None
        fresh_sig = sig is None
        if fresh_sig:
            sig = self._graph_signature(inputs, volatile)
            if cache is not None:
                cache[key] = sig
        g = self._graphs.get(key)
        stale = g is None or getattr(g, "planned", False) or g.signature != sig
        if fresh_sig and cache is not None:
            # a re-capture runs warm-up steps with their own all-reduces: every
            # rank re-captures when one has to.  Agreed where the signature is
            # taken afresh inside run_epoch - once per key and epoch; a direct
            # step call takes its signature every time, and a collective with a
            # read-back per step would cost more than the step's own all-reduce:
            # there the ranks are expected to change what a capture depends on
            # together (as SPMD code does)
            stale = parallel.any_rank(stale)
        if stale:
            if torch.cuda.is_available() and _make_capturable(self.optimizer_controller):
                sig = self._graph_signature(inputs, volatile)
            failed = None
            try:
                g = _GraphedStep(
                    part_a, part_b, self._reduce, sig,
                    list(inputs) + list(volatile) + self._held_state(), self.net,
                    self.optimizer_controller,
                    capture=torch.cuda.is_available(), split=self._reducing())
            except RuntimeError as e:
                failed = e
            # a capture that the runtime refuses must not end the run: the trainer
            # steps eagerly from here on (the warm-up steps have been undone, the
            # failed capture trained nothing) - on EVERY rank when it failed on
            # one: the others would otherwise enter the launch-form measurement's
            # collectives alone
            if parallel.any_rank(failed is not None):
                import warnings
                warnings.warn(f"graph capture of the {key} step failed "
                              f"({failed if failed is not None else 'on another rank'}); "
                              "graph_steps switched off for this trainer")
                self.graph_steps = False
                self._graphs.clear()
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                msg = part_a()
                self._reduce(msg)
                return part_b(msg)
            self._graphs[key] = g
            # the capture's own warm-up steps may have bumped versions; re-read
            g.signature = self._graph_signature(inputs, volatile)
            if cache is not None:
                cache[key] = g.signature
            if (g.capture and self.measure_launch_form
                    and self.train_mode not in self.launch_form
                    and self._measure_launch_form(g, part_a, part_b) == "eager"):
                self._graphs.pop(key, None)
                msg = part_a()
                self._reduce(msg)
                return part_b(msg)
        return g(borrow=getattr(self, "_borrow_loss", False) or self.borrow_loss)

    def _measure_launch_form(self, g, part_a, part_b):
        """Time `launch_form_steps` replays of the captured step `g` against as
        many stream-order executions of the same step, undo the training they
        did (parameters and optimizer state, in place), keep the faster form for
        this train mode.  All ranks decide alike (the times are summed)."""
        import time
        n = max(2, int(self.launch_form_steps))
        opt = self.optimizer_controller
        params = list(self.net.parameters())
        saved = [p.detach().clone() for p in params]
        state = _snapshot_optimizer_state(opt)
        # the timing steps' other side effects: the device generator
        # the LSTM's hidden-state draws consume, a dedicated hidden-state generator,
        # the running loss sums of step plans
        dev = params[0].device
        rng = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None
        hgen = getattr(self, "hidden_generator", None)
        hstate = hgen.get_state() if isinstance(hgen, torch.Generator) else None
        sums = [(pl.plan.running, pl.plan.running.clone(), pl.plan, pl.plan.launches)
                for pl in self._graphs.values() if getattr(pl, "planned", False)]
        # (an LSTM hands out hidden states from a pool drawn ahead: the timing steps
        # draw for themselves, from the generator state that is restored below)
        pool = getattr(self.net, "hidden_pool", None)
        if pool is not None:
            self.net.hidden_pool = 0

        def eager():
            msg = part_a()
            self._reduce(msg)
            return part_b(msg)

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        try:
            # (the capture left the device idle and it clocks down: the form
            # timed first paid the ramp - 12-20 ms, profiles/r06_step_ramp.txt -
            # and lost.  Untimed replays take the ramp, then the two forms are
            # timed alternately, twice, and each keeps its better time)
            replay = lambda: g(borrow=True)
            # (30 ms worth, counted from one timed run - and the same count on
            # every rank: a multi-rank step holds an all-reduce)
            warm = int(0.03 / max(timed(replay), 1e-6)) + 1
            if parallel.world_size() > 1:
                agreed = torch.tensor([warm], dtype=torch.int64, device=params[0].device)
                parallel.dist.all_reduce(agreed, op=parallel.dist.ReduceOp.MAX)
                warm = int(agreed.item())
            for _ in range(min(warm, 100 * n)):
                replay()
            t_graph = t_eager = float("inf")
            for _ in range(2):
                t_graph = min(t_graph, timed(replay))
                t_eager = min(t_eager, timed(eager))
        finally:
            with torch.no_grad():
                for p, v in zip(params, saved):
                    p.copy_(v)
                _restore_optimizer_state(opt, state)
                for running, was, plan, launches in sums:
                    running.copy_(was)
                    plan.launches = launches
            if pool is not None:
                self.net.hidden_pool = pool
            if rng is not None:
                torch.cuda.set_rng_state(rng, dev)
            if hstate is not None:
                hgen.set_state(hstate)
        if parallel.world_size() > 1:
            both = torch.tensor([t_graph, t_eager], dtype=torch.float64,
                                device=params[0].device)
            parallel.dist.all_reduce(both)
            t_graph, t_eager = (float(v) for v in both)
        choice = "graph" if t_graph <= t_eager else "eager"
        self.launch_form[self.train_mode] = choice
        self.results_dict["launch_form"].append(
            dict(train_mode=self.train_mode, chosen=choice,
                 ms_graph=t_graph * 1e3, ms_stream_order=t_eager * 1e3, steps=n))
        return choice

    def _plannable(self):
        """A single-process fused step may run from a step plan (_PlannedStep)
        instead of a captured graph."""
        return (bool(self.plan_steps) and self._graphable() and not self._reducing()
                and torch.cuda.is_available())

    def _planned(self, key, inputs, build, volatile=(), events=None, params=None,
                 index=None):
        """Run the step from its plan, (re)built by `build()` when the signature
        of `_graphed` no longer holds (same meaning of `inputs` / `volatile`)."""
        cache = getattr(self, "_epoch_sigs", None)
        sig = cache.get(key) if cache is not None else None
        if sig is None:
            sig = self._graph_signature(inputs, volatile, params)
            if cache is not None:
                cache[key] = sig
        g = self._graphs.get(key)
        if not isinstance(g, _PlannedStep) or g.signature != sig:
            g = self._graphs[key] = build()
            if not self.resident_tables:
                g.plan.resident_tables = False
            g.held_state = self._held_state()    # (addressed by the plan)
            g.signature = self._graph_signature(inputs, volatile, params)
            if cache is not None:
                cache[key] = g.signature
        if index is not None:     # (a rows plan: the batch is named per launch)
            return g(borrow=getattr(self, "_borrow_loss", False) or self.borrow_loss,
                     events=events, index=index)
        return g(borrow=getattr(self, "_borrow_loss", False) or self.borrow_loss,
                 events=events)

    def _direct_parts(self, compute, stepped=False):
        """(part_a, part_b) of a fused-policy step: compute() -> (loss,
        {name: gradient}, flat) as the functional `*_grads` entry points
        return them; the gradients are attached as `.grad` directly."""
        state = {}

        def part_a():
            loss, grads, flat = compute()
            for name, p in self.net.named_parameters():
                p.grad = grads.get(name)
            state["loss"], state["flat"] = loss, flat
            if not self._reducing():
                return None
            if flat is not None:
                flat[-1] = loss.detach().reshape(())
                return flat
            return self.grad_sync.pack(loss.detach())

        def part_b(msg):
            if msg is None:
                loss = state["loss"]
            elif msg is state["flat"]:
                loss = msg[-1]
            else:
                loss = self.grad_sync.unpack()
            if not stepped:
                self.optimizer_controller.step()
            return loss
        return part_a, part_b

    def _autograd_parts(self, forward_loss):
        """(part_a, part_b) of a step whose policy is plain PyTorch:
        forward_loss() builds the loss, autograd's backward fills `.grad`."""
        state = {}

        def part_a():
            self.optimizer_controller.zero_grad()
            loss = forward_loss()
            loss.backward()
            state["loss"] = loss.detach()
            if not self._reducing() or self.grad_sync is None:
                return None
            return self.grad_sync.pack(state["loss"])

        def part_b(msg):
            loss = state["loss"] if msg is None else self.grad_sync.unpack()
            self.optimizer_controller.step()
            return loss
        return part_a, part_b

    def _graph_index(self, index):
        """The persistent device copy of an index batch that a captured step
        reads (one buffer per batch size: the ragged last batch of an epoch
        has its own graph): lets the shuffled minibatches of run_epoch replay
        ONE captured step - only the buffer's content changes.  None when
        steps are not graphed."""
        if not self._graphable():
            return None
        buf = self._index_bufs.get(index.numel())
        if buf is None:
            buf = self._index_bufs[index.numel()] = torch.empty_like(index)
        buf.copy_(index)
        return buf

    def analytic_train_dynamics(self):
        """The fused rollouts integrate the ANALYTIC simulator described by
        `train_dynamics.params`.  A learnable simulator (an nn.Module such as
        LearntDynamics: action transform + residual network on top of the
        physics) must be called step by step instead, as the reference does
        (scripts/train_drone.py:185-191)."""
        d = self.train_dynamics
        return hasattr(d, "params") and not isinstance(d, torch.nn.Module)

    # --------------------------------------------------------- hot loop
    def train_controller_model(
        self, current_state, action_seq, in_ref_state, ref_states
    ):
        raise NotImplementedError("implemented in the system trainers")

    def train_recurrent_model(
        self, in_state, current_state, in_ref_states, ref_states
    ):
        raise NotImplementedError("only the quadrotor trainer is recurrent")

    def train_concurrent_fused(
        self, in_state, current_state, in_ref_states, ref_states, index=None,
        probe=False
    ):
        """Optional: the whole concurrent step (policy forward, rollout, loss,
        backward, optimizer) in fused kernels; None = not available.
        `index`: the four tensors are the whole data set, the batch is
        rows `index`.  `probe=True` only asks whether the indexed form is
        available (returns a bool, runs nothing)."""
        return False if probe else None

    def _residual_weight_norm(self):
        """The regulariser of the simulator fit: 2-norms of the residual
        network's four tensors, summed (scripts/train_base.py:171-178)."""
        net = self.train_dynamics
        return sum(torch.norm(t) for t in (
            net.linear_state_2.weight, net.linear_state_2.bias,
            net.linear_state_1.weight, net.linear_state_1.bias))

    def train_dynamics_model(self, current_state, action_seq):
        """One optimizer step of the simulator fit (scripts/train_base.py:
        160-186): the learnable train dynamics is pulled towards the eval
        dynamics on (state, first action of the sequence) - squared error of
        the two predicted next states, summed over the batch, plus
        l2_lambda x the residual network's weight norms."""
        first_action = action_seq[:, 0]
        self.optimizer_dynamics.zero_grad()
        predicted = self.train_dynamics(current_state, first_action,
                                        dt=self.delta_t)
        with torch.no_grad():
            target = self.eval_dynamics(current_state, first_action,
                                        dt=self.delta_t)
        loss = torch.sum((predicted - target)**2)
        if self.l2_lambda > 0:
            # data parallel: the data term is a sum over the shard, the
            # penalty is not - every rank carries 1 / world of it so that the
            # all-reduced gradient equals the single-process one
            loss = loss + (self.l2_lambda / parallel.world_size()
                           ) * self._residual_weight_norm()
        loss.backward()
        if getattr(self, "grad_sync_dynamics", None) is not None:
            loss = self.grad_sync_dynamics.sync(loss.detach())  # replicas stay equal
        self.optimizer_dynamics.step()
        self.results_dict["loss_dyn_per_step"].append(loss.detach())
        return loss

    def _pipelined_epoch(self, prepare, step, indices=None):
        """One epoch over the loader's index batches with the input pipeline
        one batch ahead: `prepare(index, out)` - the layout change with the row
        gather folded in, ~31 us per 65 536-trajectory batch - runs on a side
        stream into one of two buffer sets; `step(prepared, slot, after)` then
        starts from planes that are already there.  Events order the streams:
        `ready` (gather done -> step may read), `freed` (step done -> the next
        gather into this slot may write).
        WHEN the gather of batch i + 1 starts decides whether it pays: next to
        the sweeps of step i it slows them by as much as it costs alone
        (measured: no gain).  So where the launches are under this loop's
        control - inside the capture of a whole epoch, or without step graphs
        - it is forked LATE: behind the event the step records once its reverse
        kernel, the last reader of the inputs, is enqueued (`after`; the
        second stage, the update and the running-loss add then run next to
        the gather: ~30 us of small, latency-bound kernels).  With per-step
        graphs the event would be inside a replayed graph; the gather is then
        issued up front, before step i.  Returns (running_loss, last index)."""
        main = torch.cuda.current_stream()
        st = self._prefetch
        if "stream" not in st:
            st["stream"] = torch.cuda.Stream()
        st.setdefault("slots", {})
        side = st["stream"]
        import inspect
        takes_events = "events" in inspect.signature(step).parameters
        # (a planned step is a plain launch as well: it takes the events)
        late = (self._in_epoch_capture or not self._graphable()
                or (takes_events and self._plannable()))
        side.wait_stream(main)       # the permutation, the data set's last update

        def issue(i, index, behind=None):
            slot = st["slots"].setdefault((index.numel(), i & 1), {})
            with torch.cuda.stream(side):
                if behind is not None:
                    side.wait_event(behind)
                if "freed" in slot:
                    side.wait_event(slot["freed"])
                slot["bufs"] = prepare(index, slot.get("bufs"))
                slot["ready"] = side.record_event()
            return slot
        batches = enumerate(self.trainloader.iter_indices() if indices is None else indices)
        cur = next(batches, None)
        slot = issue(*cur) if cur is not None else None
        running, i = _EpochLoss(fresh=not self._graphable()), -1
        self._borrow_loss = True     # the loss is consumed right here
        self._epoch_sigs = {}        # nothing a capture depends on changes in here
        try:
            while cur is not None:
                i = cur[0]
                nxt = next(batches, None)
                nxt_slot = None
                if nxt is not None and not late:
                    nxt_slot = issue(*nxt)
                if late and takes_events:
                    # the step waits for its planes itself - after its pack
                    # launch - and records where the next gather may start
                    after = torch.cuda.Event()
                    after.record(main)       # (creates the handle the step re-records)
                    events = {"inputs_ready": slot["ready"]}
                    if self.gather_fork != "before":
                        events[self.gather_fork] = after
                    loss = step(slot["bufs"], i & 1, events)
                else:
                    main.wait_event(slot["ready"])
                    after = None
                    if late:
                        after = torch.cuda.Event()
                        after.record(main)
                    loss = step(slot["bufs"], i & 1)
                loss = loss.detach()
                slot["freed"] = main.record_event()
                if nxt is not None and late:
                    nxt_slot = issue(*nxt, behind=after)
                running.add(loss)
                cur, slot = nxt, nxt_slot
        finally:
            self._borrow_loss, self._epoch_sigs = False, None
        return running.total(), i

    def _indexed_epoch(self, step, indices=None):
        """One epoch of `step(index)` over the loader's index batches (the
        gather is folded into the fused step's layout change, inside its
        captured graph).  Inside the loop nothing a capture depends on changes:
        graph signatures are computed once per key, the step's loss is taken
        without a private copy and added to the running loss in place.
        Returns (running_loss, last batch index)."""
        running, i = _EpochLoss(fresh=not self._graphable()), -1
        self._borrow_loss, self._epoch_sigs = True, {}
        ld = self.trainloader
        ahead = indices is None and hasattr(ld, "epoch_order")
        try:
            for i, index in enumerate(
                    (ld.iter_indices(ld.epoch_order()) if ahead else ld.iter_indices())
                    if indices is None else indices, 0):
                running.add(step(index).detach())
            if ahead and i >= 0 and not getattr(self, "_in_epoch_capture", False):
                # stream-order epochs (a step whose launch form was measured
                # "eager" takes no epoch graph): the next epoch's permutation is
                # issued behind the last batch - same order of draws as without
                # it - and sorts while the device drains and the loss is read
                self._prefetch_order(ld)
        finally:
            self._borrow_loss, self._epoch_sigs = False, None
        return running.total(), i

    def _epoch_graph_ok(self):
        ld = self.trainloader
        return (self.graph_epochs and self._graphable() and torch.cuda.is_available()
                and not self._reducing() and hasattr(ld, "epoch_order")
                and getattr(ld, "shard", None) is None and ld.tensors[0].is_cuda)

    def _graphed_epoch(self, key, loop):
        """One epoch of `loop(indices) -> (running_loss, last index)` - through
        ONE captured graph of the whole epoch when `graph_epochs` applies.  The
        first epoch on a given configuration runs eagerly (it trains, and warms
        everything a capture may not do: lazy allocations, descriptors, the
        range check of the inputs); the second one is captured with its
        batches as views of a persistent order buffer and replayed; from then
        on an epoch is: draw the permutation, copy it into that buffer, one
        graph launch.  Re-captured (after another eager epoch) when anything
        the capture is tied to changes: data-set tensors, batch size,
        parameters, optimizer settings, simulator parameters, dt."""
        ld = self.trainloader
        if not self._epoch_graph_ok():
            return loop(None)
        order = ld.epoch_order()          # the one draw an eager epoch makes
        sig = (self._graph_signature(ld.tensors, (ld.tensors[0],))
               + (ld.batch_size, ld.shuffle))
        eg = self._epoch_graphs.get(key)
        if eg is None or eg["sig"] != sig:
            self._epoch_graphs[key] = eg = {"sig": sig, "graph": None}
            out = loop(ld.iter_indices(order=order))
            # (the first steps create what the signature also covers: the
            # optimizer's momentum buffers)
            eg["sig"] = (self._graph_signature(ld.tensors, (ld.tensors[0],))
                         + (ld.batch_size, ld.shuffle))
            return out
        if eg["graph"] is None:
            import gc
            perm = torch.empty_like(order)
            graph = torch.cuda.CUDAGraph()
            stale = self._prefetch.pop("slots", None)     # no events from outside
            self._prefetch["slots"] = {}
            was_enabled = gc.isenabled()
            gc.collect()
            gc.disable()
            self._in_epoch_capture = True
            try:
                with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"}
                                                if parallel.group_live() else {})):
                    running, last = loop(ld.iter_indices(order=perm))
            except RuntimeError as e:
                import warnings
                warnings.warn(f"capture of the {key} epoch failed ({e}); graph_epochs "
                              "switched off for this trainer")
                self.graph_epochs = False
                self._epoch_graphs.clear()
                torch.cuda.synchronize()
                return loop(ld.iter_indices(order=order))
            finally:
                self._in_epoch_capture = False
                self._prefetch["slots"] = stale if stale is not None else {}
                if was_enabled:
                    gc.enable()
            eg.update(graph=graph, perm=perm, running=running, last=last)
        eg["perm"].copy_(order)
        eg["graph"].replay()
        # the next epoch's permutation is drawn while this one runs (the host has
        # nothing else to do behind the replay; the draw is 0.25 ms of sort kernels)
        self._prefetch_order(ld)
        torch.autograd.graph.increment_version([p.detach() for p in self.net.parameters()])
        return eg["running"], eg["last"]

    def _prefetch_order(self, ld):
        if hasattr(ld, "prefetch_order"):
            if "order_stream" not in self._prefetch:
                self._prefetch["order_stream"] = torch.cuda.Stream()
            ld.prefetch_order(self._prefetch["order_stream"])

    # ---- which loop an epoch runs: ONE table, built by init_optimizer --------
    # Hooks a system trainer may override (the quadrotor trainer does):
    def prefetch_plan(self):
        """(prepare, step) of the pipelined epoch (_pipelined_epoch) or None."""
        return None

    def packed_path_ok(self):
        """The row-layout rollout path (train_controller_packed) applies."""
        return False

    def recurrent_indexed_ok(self):
        """train_recurrent_model accepts `index=` (fused recurrent steps)."""
        return False

    def _epoch_table(self):
        """[(name, predicate(train) -> bool, runner(train) -> epoch loss)], first
        match wins; the last entry always matches.  Built once per
        init_optimizer: run_epoch is a walk over it, every specialised loop is
        a named method."""
        indexed = lambda: hasattr(self.trainloader, "iter_indices")
        ctrl = lambda train: train == "controller" and indexed()
        return [
            ("concurrent, rows read through the index by the forward kernel",
             lambda t: ctrl(t) and self.train_mode == "concurrent"
             and self.concurrent_rows_ok(), self._epoch_concurrent_rows),
            ("pipelined", lambda t: ctrl(t) and self._wants_pipeline()
             and self.prefetch_plan() is not None, self._epoch_pipelined),
            ("concurrent, gather folded into the fused step",
             lambda t: ctrl(t) and self.train_mode == "concurrent"
             and self.train_concurrent_fused(None, None, None, None, probe=True),
             self._epoch_concurrent_indexed),
            ("concurrent, packed rows around any policy",
             lambda t: ctrl(t) and self.train_mode == "concurrent"
             and getattr(self, "use_packed_path", True) and self.packed_path_ok(),
             self._epoch_packed),
            ("recurrent, gather folded into the fused step",
             lambda t: ctrl(t) and self.train_mode != "concurrent"
             and self.recurrent_indexed_ok(), self._epoch_recurrent_indexed),
            ("loader", lambda t: True, self._epoch_loader),
        ]

    def concurrent_rows_ok(self):
        """Hook: the fused concurrent step reads index batches itself."""
        return False

    def _wants_pipeline(self):
        # (the concurrent epoch graph forks the next batch's gather behind the
        # reverse kernel: that is where it is free)
        return self.prefetch_batches or (
            self.train_mode == "concurrent" and self.trainloader is not None
            and hasattr(self.trainloader, "epoch_order") and self._epoch_graph_ok())

    def _epoch_pipelined(self, train):
        plan = self.prefetch_plan()
        return self._finish_epoch(*self._graphed_epoch(
            (self.train_mode, "prefetch"),
            lambda indices: self._pipelined_epoch(*plan, indices=indices)), train)

    def _epoch_concurrent_indexed(self, train):
        tensors = self.trainloader.tensors
        return self._finish_epoch(*self._graphed_epoch(
            ("concurrent", "indexed"), lambda indices: self._indexed_epoch(
                lambda index: self.train_concurrent_fused(*tensors, index=index),
                indices)), train)

    def _epoch_concurrent_rows(self, train):
        """The concurrent epoch with every batch named by its rows: per batch
        ONE library call (apg_quad_mlp_concurrent_train_step_rows - the forward
        kernel reads the data set through the index, the second stage applies
        the update and adds the loss to the plan's running sum).  The first
        batch of each size goes through the step's own entry point (which builds
        or re-validates the plan, once per epoch); the others call the plan
        directly - ~25 us of host time against ~140 us of kernels, so the launches
        run back to back in stream order: no graph, no gather, no loss kernel."""
        ld = self.trainloader
        tensors = ld.tensors
        for g in self._graphs.values():
            if getattr(g, "planned", False) and hasattr(g.plan, "running"):
                g.plan.running.zero_()
        fast, i, extra = {}, -1, None
        self._borrow_loss, self._epoch_sigs = True, {}
        order = ld.epoch_order() if hasattr(ld, "prefetch_order") else None
        try:
            for i, index in enumerate(ld.iter_indices(order)):
                if i == 2 and order is not None:
                    # the next epoch's permutation, drawn beside this one - its
                    # ~20 launches are 0.2 ms of HOST time: issued here, behind
                    # two queued batches, not in front of the epoch's first
                    # (profiles/r06_epoch_concurrent.txt)
                    self._prefetch_order(ld)
                plan = fast.get(index.numel())
                if plan is not None:
                    plan.launch(index=index)
                    continue
                key = ("concurrent", index.numel(), "rows")
                g0 = self._graphs.get(key)
                n0 = getattr(getattr(g0, "plan", None), "launches", None)
                loss = self.train_concurrent_fused(*tensors, index=index)
                g = self._graphs.get(key)
                if (getattr(g, "planned", False) and hasattr(g.plan, "running")
                        and (g is not g0 or g.plan.launches != n0)):
                    fast[index.numel()] = g.plan     # (its running sum has this step)
                else:      # the step took another route: its loss is added here
                    extra = loss.detach().clone() if extra is None else extra + loss.detach()
            if order is not None and i < 2:        # (an epoch of one or two batches)
                self._prefetch_order(ld)
        finally:
            self._borrow_loss, self._epoch_sigs = False, None
        total = None if extra is None else extra.reshape(1)
        for plan in fast.values():
            total = plan.running.clone() if total is None else total + plan.running
        return self._finish_epoch(total if total is not None
                                  else torch.zeros(1, device=tensors[0].device), i, train)

    def _epoch_packed(self, train):
        # any PyTorch policy on the row-layout tensors of the fastest rollout
        # kernel (TrainDrone.train_controller_packed); the minibatch is a gather
        # of rows out of the data set's cached packed tensors, the whole-set
        # batch is those tensors themselves
        running_loss, i = None, -1
        normed, _, in_ref, _ = self.trainloader.tensors
        s0_rows, ref_rows = self.state_data.packed()
        n = normed.shape[0]
        for i, index in enumerate(self.trainloader.iter_indices(), 0):
            whole = (not self.shuffle and index.numel() == n)
            if whole:
                batch = (normed, in_ref, s0_rows, ref_rows)
            else:
                batch = (normed.index_select(0, index),
                         in_ref.index_select(0, index),
                         s0_rows.index_select(1, index),
                         ref_rows.index_select(1, index))
            loss = self.train_controller_packed(*batch).detach()
            running_loss = loss if running_loss is None else running_loss + loss
        return self._finish_epoch(running_loss, i, train)

    def _epoch_recurrent_indexed(self, train):
        tensors = self.trainloader.tensors
        step = lambda index: self.train_recurrent_model(*tensors, index=index)
        if (self.train_mode == "LSTM"
                and getattr(self, "hidden_generator", None) is not None):
            # a private generator is not registered with graphs: eager steps
            return self._finish_epoch(*self._indexed_epoch(step), train)
        return self._finish_epoch(*self._graphed_epoch(
            (self.train_mode, "indexed"),
            lambda indices: self._indexed_epoch(step, indices)), train)

    def _epoch_loader(self, train):
        running_loss, i = None, -1
        for i, data in enumerate(self.trainloader, 0):
            in_state, current_state, in_ref_state, ref_states = data
            if train == "dynamics":
                with torch.no_grad():
                    actions = torch.sigmoid(self.net(in_state, in_ref_state))
                action_seq = torch.reshape(
                    actions, (-1, self.actions_out_dim // self.action_dim,
                              self.action_dim))
                loss = self.train_dynamics_model(current_state, action_seq)
                self.count_finetune_data += len(current_state)
            elif self.train_mode != "concurrent":
                loss = self.train_recurrent_model(
                    in_state, current_state, in_ref_state, ref_states
                )
            elif (loss := self.train_concurrent_fused(
                    in_state, current_state, in_ref_state, ref_states)) is not None:
                pass        # policy + rollout + backward in the fused kernels
            else:
                # policy -> (0, 1) actions for the whole horizon -> [B, H, A]
                # (scripts/train_base.py:199-208), then the fused rollout step
                plan = torch.sigmoid(self.net(in_state, in_ref_state))
                loss = self.train_controller_model(
                    current_state, plan.view(-1, self.horizon, self.action_dim),
                    in_ref_state, ref_states)
            loss = loss.detach()
            running_loss = loss if running_loss is None else running_loss + loss
        return self._finish_epoch(running_loss, i, train)

    def run_epoch(self, train="controller", epoch=0):
        """scripts/train_base.py:188-218.  The loop that runs is the first
        entry of the epoch table whose predicate holds (`last_epoch_loop` names
        it)."""
        if train not in ("controller", "dynamics"):
            raise ValueError("train must be 'controller' or 'dynamics'")
        table = getattr(self, "_epoch_runners", None) or self._epoch_table()
        self._guard_data_set(train)
        # step plans keep their packed operand tables from step to step and notice
        # foreign parameter writes by version counter + address; a write through
        # `p.data` moves neither: the tables are packed afresh at the
        # first batch of every epoch - one ~6 us launch per epoch and plan
        for g in self._graphs.values():
            if getattr(g, "planned", False):
                if not self.resident_tables:
                    g.plan.resident_tables = False
                g.plan.invalidate()
        if self.net is not None and F_lstm_tables(self.net) is not None:
            F_lstm_tables(self.net).invalidate()
        for name, applies, runner in table:
            if applies(train):
                self.last_epoch_loop = name
                return runner(train)

    def _guard_data_set(self, train):
        """The in-kernel policies take finite inputs below 2^14 (fp16-split
        operands, include/apg.h).  Steps that replay a graph, a plan or an epoch
        graph never pass the per-call host check, and the data set is rewritten
        in place between epochs (resample_data, self-play add_eval_data): its
        tensors are checked here, before the epoch's first replay - one
        read-back per tensor and in-place version, nothing when unchanged."""
        d = self.state_data
        if (train != "controller" or d is None
                or not getattr(self, "fused_policy", False)):
            return
        from . import functional as F
        tensors = {k: getattr(d, k, None)
                   for k in ("normed_states", "states", "in_ref_states", "ref_states")}
        F._guard_policy_inputs(
            "run_epoch (data set)",
            **{k: t for k, t in tensors.items() if torch.is_tensor(t) and t.is_cuda})

    def _finish_epoch(self, running_loss, i, train):
        # one host read-back per epoch; divides by the last index as the
        # reference does (ZeroDivisionError with a single batch, as there)
        epoch_loss = float(running_loss.item()) / i
        self.results_dict["loss"].append(epoch_loss)
        self.results_dict["trained"].append(train)
        if parallel.is_main():
            print(f"Loss ({train}): {round(epoch_loss, 2)}")
        return epoch_loss

    def sample_new_data(self, epoch):
        if (epoch + 1) % self.resample_every == 0:
            self.state_data.resample_data()
            self.sampled_data_count += self.state_data.num_sampled_states

    # -------------------------------------------------------------- cold
    def evaluate_model(self, epoch):
        """Hook (scripts/train_base.py:298): TrainDrone evaluates in closed loop."""
        return None

    def save_model(self, epoch, success=0.0, suc_std=0.0):
        """scripts/train_base.py:233-251: a checkpoint per evaluated epoch
        (not for epoch 0) and the score bookkeeping.  A state_dict is written
        instead of the pickled module; checkpoint.load_policy reads those
        (whole-module pickles of the reference are converted once, see
        checkpoint.py).  Rank 0 writes."""
        if epoch > 0:
            self.current_score = success
            if parallel.is_main():
                os.makedirs(self.save_path, exist_ok=True)
                torch.save(self.net.state_dict(), os.path.join(
                    self.save_path, self.save_model_name + str(epoch)))

    def finalize(self):
        """scripts/train_base.py:253-287 without the plots: final weights,
        the per-epoch statistics as csv, results.json and - for a learnable
        simulator - its state_dict as `dynamics_model`.  Rank 0 writes."""
        if not parallel.is_main():
            return
        import json
        os.makedirs(self.save_path, exist_ok=True)
        torch.save(self.net.state_dict(),
                   os.path.join(self.save_path, self.save_model_name))
        res = self.results_dict
        for key, fname in (("mean_success", "mean_successes.csv"),
                           ("std_success", "std_success.csv"),
                           ("loss", "loss.csv"),
                           ("mean_divergence_full", "mean_divergence_full.csv"),
                           ("std_divergence_full", "std_divergence_full.csv"),
                           ("mean_divergence", "mean_divergence.csv"),
                           ("std_divergence", "std_divergence.csv")):
            np.savetxt(os.path.join(self.save_path, fname),
                       np.asarray([float(v) for v in res[key]]), delimiter=",")

        def plain(v):
            if isinstance(v, torch.Tensor):
                return v.detach().cpu().tolist()
            if isinstance(v, (np.floating, np.integer)):
                return v.item()
            if isinstance(v, np.ndarray):
                return v.tolist()
            return v
        with open(os.path.join(self.save_path, "results.json"), "w") as f:
            json.dump({k: [plain(x) for x in v] for k, v in res.items()}, f)
        if isinstance(self.train_dynamics, torch.nn.Module):
            torch.save(self.train_dynamics.state_dict(),
                       os.path.join(self.save_path, "dynamics_model"))
        print("finished and saved.")

    def _speed_curriculum(self, epoch, track):
        """The speed ladder of run_control (scripts/train_base.py:300-313):
        once the last five evaluations all flew longer than a full reference
        at the current speed - or after 100 epochs at it - the references get
        0.1 faster (up to 0.4) and the divergence threshold restarts at 0.1."""
        cfg = self.config
        full_length = 1000 / (cfg["speed_factor"] / cfg.get("delta_t", self.delta_t))
        track["successes"].append(self.results_dict["mean_success"][-1])
        recent = track["successes"][-5:]
        mastered = len(track["successes"]) > 5 and min(recent) > full_length
        if (mastered or epoch - track["since"] > 100) and cfg["speed_factor"] < 0.4:
            cfg["speed_factor"] += 0.1
            cfg["thresh_div"] = 0.1
            track["successes"] = []
            track["since"] = epoch + 1
            self.current_score = 0 if self.suc_up_down == 1 else np.inf

    def run_control(self, config, sampling_based_finetune=False, curriculum=0):
        """scripts/train_base.py:289-332.  `curriculum` (off by default here;
        the reference's default is on) needs the closed-loop statistics of
        evaluate_model, i.e. a trainer that provides the hook."""
        track = {"successes": [], "since": 0}
        if curriculum:
            self.config["speed_factor"] = 0.2
        try:
            for epoch in range(config["nr_epochs"]):
                evaluated = self.evaluate_model(epoch)
                if evaluated is None:          # hook not provided:
                    self.sample_new_data(epoch)  # it resamples itself
                elif curriculum:
                    self._speed_curriculum(epoch, track)
                print(f"\nEpoch {epoch}")
                self.run_epoch(train="controller", epoch=epoch)
                if sampling_based_finetune:
                    self.results_dict["samples_in_d2"].append(
                        getattr(self.state_data, "eval_counter", 0))
        except KeyboardInterrupt:
            pass
        self.finalize()

    def run_dynamics(self, config):
        """scripts/train_base.py:334-375: fit the learnable
        train dynamics for the first `train_dyn_for_epochs` epochs (every
        `train_dyn_every`-th), then train the controller through it."""
        until = config.get("train_dyn_for_epochs", 10)
        every = config.get("train_dyn_every", 1)
        try:
            for epoch in range(config["nr_epochs"]):
                if self.evaluate_model(epoch) is None:
                    self.sample_new_data(epoch)
                fit = epoch <= until and epoch % every == 0
                print(f"\nEpoch {epoch}")
                self.run_epoch(train="dynamics" if fit else "controller",
                               epoch=epoch)
                self.results_dict["samples_in_d2"].append(
                    self.count_finetune_data)
                if epoch == until:   # the controller phase starts from scratch
                    self.current_score = 0 if self.suc_up_down == 1 else np.inf
        except KeyboardInterrupt:
            pass
        self.finalize()
