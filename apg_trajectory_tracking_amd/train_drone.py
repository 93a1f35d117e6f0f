"""Drop-in for scripts/train_drone.py:27-257 (`TrainDrone`) restricted to the
APG hot path: `train_controller_model` (:175-203, concurrent k-step unroll)
and `train_recurrent_model` (:113-173, autoregressive / LSTM unroll).

concurrent : one fused HIP launch does H x dynamics + quad_mpc_loss + the
             analytic adjoint (functional.quad_rollout_loss) in place of ~600
             eager ops + autograd.
recurrent  : the policy runs inside the loop, so each step is
             state_preprocessing (HIP) -> policy (PyTorch) -> dynamics (HIP),
             all differentiable; quad_mpc_loss (HIP) closes the graph.
             Semantics pinned in SURVEY.md §8a A4: the reference window is
             COPIED before the current position is subtracted (as shipped,
             the reference writes through a view, which breaks autograd on
             torch 2.x and subtracts cumulatively).
"""
import json
import os

import torch

from . import functional as F
from .dataset import SyntheticQuadDataset, state_preprocessing
from .drone_loss import quad_mpc_loss
from .models.hutter_model import Net
from .models.rnn import LSTM_NEW
from .train_base import TrainBase, _PlannedStep


class TrainDrone(TrainBase):

    def __init__(self, train_dynamics, eval_dynamics, config):
        self.config = config
        super().__init__(train_dynamics, eval_dynamics, **config)
        if self.sample_in not in ("eval_env", "train_env", "real_flightmare"):
            raise ValueError(
                "sample in must be one of eval_env, train_env, real_flightmare"
            )
        self.hidden_generator = None   # seeds LSTM (h0, c0) draws if set
        # LSTM mode: run the policy INSIDE the rollout kernel (K7) when the
        # network is the reference architecture LSTM_NEW(15, 10, 9, 4, conv=1)
        self.fused_policy = True
        self.fused_learnt = True    # controller phase through LearntDynamics
        # True: the fused steps are called (index=None) on the SAME resident
        # tensors step after step - their plane-layout copies are then kept
        # instead of being rebuilt per step (functional._StaticPlanes)
        self.static_shard = False

    def initialize_model(self, base_model=None, modified_params={},
                         state_data=None, device=None, seed=0,
                         base_model_name="model_quad"):
        """Policy + dataset + optimizer (scripts/train_drone.py:51-112).
        `base_model`: a module, or - as in the reference - the directory of a
        trained model (`<dir>/model_quad`, here a state_dict checkpoint, see
        checkpoint.py).  `state_data` defaults to the seeded synthetic
        polynomial set (the reference samples `data/traj_data_1`, which
        upstream does not ship).  The run's parameters are written to
        `<save_path>/config.json` like the reference does."""
        device = torch.device(device or "cuda")
        if isinstance(base_model, (str, os.PathLike)):
            from .checkpoint import load_policy
            base_model = load_policy(os.path.join(base_model, base_model_name),
                                     conv=True)
        if state_data is None:
            state_data = SyntheticQuadDataset(
                self.epoch_size, self.horizon, self.delta_t,
                ref_length=self.ref_length, seed=seed, device=device,
                self_play=self.self_play)
        self.state_data = state_data
        in_state_size = self.state_data.normed_states.size()[1]
        if base_model is not None:
            self.net = base_model
        else:
            net_class = LSTM_NEW if self.train_mode == "LSTM" else Net
            self.net = net_class(
                in_state_size, self.horizon, self.ref_dim,
                self.actions_out_dim, conv=1)
        self.net.to(device)
        if isinstance(self.train_dynamics, torch.nn.Module):
            self.train_dynamics.to(device)     # learnable simulator (N3)
        self.config["ref_length"] = self.ref_length
        self.config["dt"] = self.delta_t
        self.config["take_every_x"] = self.self_play_every_x
        self.config["thresh_div"] = self.thresh_div_start
        self.config["thresh_stable"] = self.thresh_stable_start
        self.config["mean"] = torch.as_tensor(self.state_data.mean).tolist()
        self.config["std"] = torch.as_tensor(self.state_data.std).tolist()
        self.config["modified_params"] = {
            k: (v.tolist() if hasattr(v, "tolist") else v)
            for k, v in modified_params.items()}
        from . import parallel
        if parallel.is_main():       # one writer under torch.distributed
            os.makedirs(self.save_path, exist_ok=True)
            with open(os.path.join(self.save_path, "config.json"), "w") as f:
                json.dump(self.config, f, default=str)
        self.init_optimizer()

    def recurrent_forward_as_shipped(self, current_state, in_ref_states, ref_states):
        """SURVEY.md §8a A4 `legacy_inplace_ref`: the forward half of
        train_recurrent_model exactly as scripts/train_drone.py:134-165 ships it -
        the reference window shifted IN PLACE through a view of the batch, every
        step again (:138-142) - in one fused launch; returns (loss, states
        [B,H,12], actions [B,H,4]).  Forward only: the reference itself cannot
        back-propagate through that write (torch >= 1.5 refuses), which is why
        training uses the copied window; the batch tensor is left as it was."""
        n = self.net
        hc = (None, None)
        if self.train_mode == "LSTM":
            n.reset_hidden_state(current_state.size()[0])
            hc = (n.hidden_state.to(current_state.device), n.cell_state.to(current_state.device))
        states, actions = F.quad_recurrent_forward_inplace_ref(
            n, current_state, in_ref_states, self.delta_t, self.train_dynamics.params, *hc)
        loss = quad_mpc_loss(states, ref_states[:, :self.horizon], actions, printout=0)
        return loss, states, actions

    def train_recurrent_model(
        self, in_state, current_state, in_ref_states, ref_states, index=None,
        prepared=None, slot=0
    ):
        """`index` (fused paths only): the tensors are the whole data set, the
        batch is rows `index` (gathered inside the kernels' layout change).
        `prepared` (fused paths only): functional.quad_recurrent_prepare's
        result for this batch - the layout change already ran (one batch ahead,
        TrainBase._pipelined_epoch; `slot` = which of its two buffer sets) and
        the four tensors are unused."""
        self.optimizer_controller.zero_grad()
        fused = self.fused_policy and (
            self._fusable() if self.train_mode == "LSTM" else self._fusable_mlp())
        # autoregressive, one process: the reverse sweep's second stage applies
        # the optimizer's update (as the concurrent step's does)
        update = None
        if fused and self.train_mode != "LSTM":
            nb = (prepared[2].shape[-1] if prepared is not None
                  else current_state.size()[0] if index is None else index.numel())
            update = self._in_kernel_update(0 < nb <= F._MAX_FUSED_AR_BATCH)
        elif fused:
            # LSTM (round 6): the launch behind the weight products applies the
            # update, packs the next step's tables, reduces the loss
            n = self.net
            update = self._in_kernel_update(
                self.resident_tables, names=F._LSTM_PARAMS,
                tensors=(n.conv_ref.weight, n.conv_ref.bias, n.lstm.weight_ih,
                         n.lstm.weight_hh, n.lstm.bias_ih, n.lstm.bias_hh,
                         n.fc_out.weight, n.fc_out.bias))
        stepped = update is not None
        lstm_kw = dict(update=update, resident_tables=bool(self.resident_tables))
        if prepared is not None:
            if not fused:
                raise ValueError("prepared batches need the fused policy path")
            batch_size = prepared[2].shape[-1]
            kw = dict(prepared=prepared)
            lstm_eager = self.train_mode == "LSTM" and self.hidden_generator is not None
            key = (self.train_mode.lower(), batch_size, "slot", slot)
            if self.train_mode == "LSTM":
                def compute():
                    self.net.reset_hidden_state(
                        batch_size, generator=self.hidden_generator)
                    return F.quad_lstm_rollout_grads(
                        self.net, None, None, None, self.delta_t,
                        self.train_dynamics.params, self.net.hidden_state,
                        self.net.cell_state, **kw, **lstm_kw)
            else:
                def compute():
                    return F.quad_mlp_rollout_grads(
                        self.net, None, None, None, self.delta_t,
                        self.train_dynamics.params, update=update, **kw)
            if lstm_eager:
                return self._step_direct(*compute(), stepped=stepped)
            return self._graphed(key, (), self._direct_parts(compute, stepped),
                                 volatile=tuple(prepared))
        batch_size = current_state.size()[0] if index is None else index.numel()
        static = index is None and self.static_shard
        tensors = (current_state, in_ref_states, ref_states)
        held = (self._graph_index(index)
                if index is not None and fused
                and not (self.train_mode == "LSTM"
                         and self.hidden_generator is not None) else None)
        if held is not None:
            index = held      # the captured gather reads this buffer
        if self.train_mode == "LSTM":
            if self.fused_policy and self._fusable():
                def compute():
                    self.net.reset_hidden_state(
                        batch_size, generator=self.hidden_generator)
                    return F.quad_lstm_rollout_grads(
                        self.net, current_state, in_ref_states, ref_states,
                        self.delta_t, self.train_dynamics.params,
                        self.net.hidden_state, self.net.cell_state, index=index,
                        static_inputs=self.static_shard, **lstm_kw)
                # (a private hidden-state generator is not registered with the
                # graph: those runs step eagerly)
                if held is not None:
                    return self._graphed(("lstm", batch_size), tensors,
                                         self._direct_parts(compute, stepped),
                                         volatile=(held,))
                if static and self.hidden_generator is None:
                    return self._graphed("lstm", tensors,
                                         self._direct_parts(compute, stepped))
                return self._step_direct(*compute(), stepped=stepped)
            self.net.reset_hidden_state(
                batch_size, generator=self.hidden_generator)
        elif self.fused_policy and self._fusable_mlp():
            def compute():
                return F.quad_mlp_rollout_grads(
                    self.net, current_state, in_ref_states, ref_states,
                    self.delta_t, self.train_dynamics.params, index=index,
                    static_inputs=self.static_shard, update=update)
            if held is not None:
                return self._graphed(("autoregressive", batch_size), tensors,
                                     self._direct_parts(compute, stepped), volatile=(held,))
            if static:
                return self._graphed("autoregressive", tensors,
                                     self._direct_parts(compute, stepped))
            return self._step_direct(*compute(), stepped=stepped)
        if index is not None:     # per-step path: materialise the batch
            current_state, in_ref_states, ref_states = (
                t.index_select(0, index) for t in
                (current_state, in_ref_states, ref_states))
        states, actions = [], []
        for k in range(self.horizon):
            rel = in_ref_states[:, k:k + self.horizon].clone()
            rel[:, :, :3] = rel[:, :, :3] - current_state[:, None, :3]
            in_state = state_preprocessing(current_state)
            action = torch.sigmoid(self.net(in_state, rel))
            actions.append(action)
            current_state = self.train_dynamics(
                current_state, action, dt=self.delta_t)
            states.append(current_state)
        intermediate_states = torch.stack(states, dim=1)
        action_seq = torch.stack(actions, dim=1)
        loss = quad_mpc_loss(
            intermediate_states, ref_states[:, :self.horizon], action_seq)
        return self._step(loss)

    def prefetch_plan(self):
        """run_epoch's hook for the fused modes: (prepare, step) - `prepare(
        index, out)` is the batch's layout change + row gather (issued one
        batch ahead on a side stream, TrainBase._pipelined_epoch), `step(
        prepared, slot)` the optimizer step on its result.  None: no fused
        path for this trainer / network."""
        if not (torch.cuda.is_available() and self.trainloader is not None
                and self.trainloader.tensors[0].is_cuda):
            return None
        normed, states, in_ref, ref = self.trainloader.tensors
        if self.train_mode == "concurrent":
            if not self.train_concurrent_fused(None, None, None, None, probe=True):
                return None
            return (lambda index, out: F.quad_concurrent_prepare(
                        normed, states, in_ref, ref, index=index, out=out),
                    # (takes the loop's events: ApgStepEvents)
                    lambda prepared, slot, events=None: self.train_concurrent_fused(
                        None, None, None, None, prepared=prepared, slot=slot,
                        events=events))
        if not self.recurrent_indexed_ok():
            return None
        return (lambda index, out: F.quad_recurrent_prepare(
                    states, in_ref, ref, index=index, out=out),
                # (recurrent steps: the inputs are read by the products as well;
                # the event stays where the loop recorded it - before the step)
                lambda prepared, slot: self.train_recurrent_model(
                    None, None, None, None, prepared=prepared, slot=slot))

    # the concurrent step names its batch by row numbers and the forward kernel
    # reads the data set's rows itself (no gather pass)
    rows_in_kernel = True

    @staticmethod
    def _rows_ok(normed, states, in_ref, ref, index):
        ok = lambda t: (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                        and t.numel() * 4 < (1 << 32) - 64)
        # (every shape the rows plan insists on: a mismatch falls back to the
        # gather path instead of raising from the plan - ADVICE r5)
        N = states.shape[0] if states.dim() == 2 else -1
        return (all(ok(t) for t in (normed, states, in_ref, ref))
                and index.is_cuda and index.dtype == torch.int64 and index.is_contiguous()
                and states.dim() == 2 and states.shape[1] == 12
                and normed.shape[0] == N and in_ref.shape[0] == N and ref.shape[0] == N
                and normed.dim() == 2 and normed.shape[1] == 15 and in_ref.dim() == 3
                and in_ref.shape[1] >= 10 and in_ref.shape[2] == 9 and ref.dim() == 3
                and ref.shape[1] >= 10 and ref.shape[2] in (9, 6))

    def concurrent_rows_ok(self):
        ld = self.trainloader
        if not (self.rows_in_kernel and torch.cuda.is_available() and ld is not None
                and self.train_concurrent_fused(None, None, None, None, probe=True)
                and self._in_kernel_update() is not None
                and self._plannable()):
            return False
        probe = torch.empty(0, dtype=torch.int64, device=ld.tensors[0].device)
        return self._rows_ok(*ld.tensors, probe)

    def recurrent_indexed_ok(self):
        """run_epoch may hand index batches to train_recurrent_model."""
        return self.fused_policy and (
            self._fusable() if self.train_mode == "LSTM" else self._fusable_mlp())

    def _fusable(self):
        n = self.net
        return (isinstance(n, LSTM_NEW) and n.conv and self.horizon == 10
                and self.analytic_train_dynamics()
                and n.lstm.weight_ih.shape == (32, 175)
                and n.conv_ref.weight.shape == (20, 9, 3)
                and n.fc_out.weight.shape == (4, 8))

    def train_concurrent_fused(
        self, in_state, current_state, in_ref_states, ref_states, index=None,
        probe=False, prepared=None, slot=0, events=None
    ):
        """scripts/train_base.py:198-204 + scripts/train_drone.py:175-203 with
        the policy inside the kernels (apg_quad_mlp_concurrent_train_step).
        `prepared`: functional.quad_concurrent_prepare's result for this batch
        (see train_recurrent_model)."""
        n = self.net
        ok = (self.fused_policy and isinstance(n, Net) and n.conv
                and self.horizon == 10 and self.analytic_train_dynamics()
                and n.states_in.weight.shape == (64, 15)
                and n.conv_ref.weight.shape == (20, 9, 3)
                and n.fc1.weight.shape == (64, 224)
                and n.fc_out.weight.shape == (40, 64))
        if probe:
            return ok
        if not ok:
            return None
        # one process: the optimizer's update happens inside the step's second
        # stage (no separate SGD launch); more ranks: after the all-reduce
        update = self._in_kernel_update()
        stepped = update is not None
        planned = stepped and self._plannable()
        dyn = self.train_dynamics
        tensors12 = (F.mlp_param_objects(n) or None) if planned else None
        if planned and prepared is not None:
            # (a slot of the pipelined epoch: its buffers are refilled by the
            # loop's gather, the plan reads them by address)
            return self._planned(
                ("concurrent", prepared[1].shape[-1], "slot", slot), (),
                lambda: _PlannedStep(F.QuadConcurrentStepPlan(
                    n, prepared, self.delta_t, dyn.params, update=update), n),
                volatile=tuple(prepared), events=events, params=tensors12)
        if planned and index is not None and self.rows_in_kernel and self._rows_ok(
                in_state, current_state, in_ref_states, ref_states, index):
            # the forward kernel reads the batch's rows through the index itself
            # (apg_quad_mlp_concurrent_train_step_rows): no gather, no copy of
            # the index - the plan takes its address per launch
            B = index.numel()
            src = (in_state, current_state, in_ref_states, ref_states)
            # the kernel reads THESE tensors' rows: the operand range of the
            # in-kernel policy is checked on them (cached by in-place version:
            # nothing per step) - a direct call with an index gets the check the
            # gather path's entry points make
            F._guard_policy_inputs("fused concurrent step (rows)", normed=in_state,
                                   in_ref=in_ref_states)
            return self._planned(
                ("concurrent", B, "rows"), src,
                lambda: _PlannedStep(F.QuadConcurrentStepPlan(
                    n, None, self.delta_t, dyn.params, update=update,
                    rows=src + (B,)), n),
                params=tensors12, index=index)
        if planned and index is not None:
            held = self._graph_index(index)       # persistent copy of the batch rows
            B = held.numel()

            def build():
                out = F.quad_concurrent_prepare(in_state, current_state, in_ref_states,
                                                ref_states, index=held)
                gather = lambda: F.quad_concurrent_prepare(
                    in_state, current_state, in_ref_states, ref_states, index=held,
                    out=out)
                return _PlannedStep(F.QuadConcurrentStepPlan(
                    n, out, self.delta_t, dyn.params, update=update), n, before=gather)
            return self._planned(("concurrent", B),
                                 (in_state, current_state, in_ref_states, ref_states),
                                 build, volatile=(held,), params=tensors12)
        if planned and self.static_shard and in_state is not None:
            src = (in_state, current_state, in_ref_states, ref_states)

            def build():
                hit = F._STATIC_PLANES.lookup("concurrent", src)
                if hit is None:
                    hit = F._STATIC_PLANES.store(
                        "concurrent", src, F.quad_concurrent_prepare(*src))
                return _PlannedStep(F.QuadConcurrentStepPlan(
                    n, hit, self.delta_t, dyn.params, update=update), n)
            return self._planned("concurrent", src, build, params=tensors12)
        if prepared is not None:
            def compute():
                return F.quad_concurrent_policy_grads(
                    n, None, None, None, None, self.delta_t,
                    self.train_dynamics.params, prepared=prepared,
                    events=events, update=update)
            return self._graphed(("concurrent", prepared[1].shape[-1], "slot", slot),
                                 (), self._direct_parts(compute, stepped),
                                 volatile=tuple(prepared))
        tensors = (in_state, current_state, in_ref_states, ref_states)
        held = None if index is None else self._graph_index(index)
        if held is not None:
            index = held      # the captured gather reads this buffer

        def compute():
            return F.quad_concurrent_policy_grads(
                n, in_state, current_state, in_ref_states, ref_states, self.delta_t,
                self.train_dynamics.params, index=index,
                static_inputs=self.static_shard, update=update)
        if held is not None:
            return self._graphed(("concurrent", held.numel()), tensors,
                                 self._direct_parts(compute, stepped), volatile=(held,))
        if index is None and self.static_shard:
            return self._graphed("concurrent", tensors,
                                 self._direct_parts(compute, stepped))
        return self._step_direct(*compute(), stepped=stepped)

    # ------------------------------------------- packed (row-layout) path --
    # scripts/train_base.py:198-209 + scripts/train_drone.py:175-203 for ANY
    # PyTorch policy, on the tensors the fastest rollout kernel reads
    # (quad_rollout_rows_kernel, APG_LAYOUT_PACKED): the data set keeps its
    # state0 / reference rows in that layout (SyntheticQuadDataset.packed),
    # the policy's action sequence is produced as [H, B, 4] rows - by
    # `forward_packed` when the network has one (the head GEMM batched over
    # the horizon, no transpose), otherwise by one transposing copy of the
    # network's [B, 4H] output - and dL/dactions returns to autograd in the
    # same layout.
    def packed_path_ok(self):
        return (self.train_mode == "concurrent" and self.horizon in (5, 10)
                and self.analytic_train_dynamics()
                and hasattr(self.state_data, "packed")
                and self.ref_length == self.horizon)

    def policy_action_rows(self, in_state, in_ref_states):
        """sigmoid(policy) as action rows [H, B, action_dim]."""
        net = self.net
        if hasattr(net, "forward_packed"):
            return torch.sigmoid(net.forward_packed(in_state, in_ref_states))
        plan = torch.sigmoid(net(in_state, in_ref_states))
        return plan.view(-1, self.horizon, self.action_dim).transpose(
            0, 1).contiguous()

    def train_controller_packed(self, in_state, in_ref_states, state0_rows,
                                ref_rows):
        """One optimizer step of the concurrent mode on packed tensors:
        state0_rows [3, B, 4], ref_rows [H, B, 6] = [pos, vel]; policy inputs
        in the reference's layout.  Same arithmetic as run_epoch's concurrent
        body followed by train_controller_model."""
        def forward_loss():
            action_rows = self.policy_action_rows(in_state, in_ref_states)
            return F.quad_rollout_loss(
                state0_rows, action_rows, ref_rows, self.delta_t,
                self.train_dynamics.params, layout="packed")
        if self.static_shard:
            # resident tensors: forward, autograd's backward and the update are
            # captured once and replayed (graph_steps; the policy is arbitrary
            # PyTorch code, so ~40 small launches per step otherwise)
            return self._graphed("packed", (in_state, in_ref_states, state0_rows,
                                            ref_rows),
                                 self._autograd_parts(forward_loss)).detach()
        self.optimizer_controller.zero_grad()
        return self._step(forward_loss()).detach()

    def _fusable_learnt(self):
        from .dynamics.quad_dynamics_trained import LearntDynamics
        d = self.train_dynamics
        return (isinstance(d, LearntDynamics)
                and self.horizon <= 48
                and tuple(d.linear_state_1.weight.shape) == (64, 16)
                and tuple(d.linear_state_2.weight.shape) == (12, 64)
                and d.linear_at.is_cuda)

    def _fusable_mlp(self):
        n = self.net
        return (isinstance(n, Net) and n.conv and self.horizon == 10
                and self.train_mode == "autoregressive"
                and self.analytic_train_dynamics()
                and n.states_in.weight.shape == (64, 15)
                and n.conv_ref.weight.shape == (20, 9, 3)
                and n.fc1.weight.shape == (64, 224)
                and n.fc_out.weight.shape == (4, 64))

    def evaluate_model(self, epoch):
        """scripts/train_drone.py:205-238: closed-loop evaluation on "rand"
        references (all runs in one launch, evaluate_drone.QuadEvaluator),
        resampling, divergence-threshold curriculum, checkpoint, statistics."""
        from .evaluate_drone import QuadEvaluator
        n = self.net
        # the environment flown in: `sample_in` (scripts/train_drone.py:39-49);
        # after train_dynamics() that is the LEARNT simulator, residual
        # network included (:44-45) - the closed-loop kernels step through it
        # (csrc/learnt_residual.h), so self-play states, the score and the
        # threshold ladder come from the same model as in the reference
        from .dynamics.quad_dynamics_trained import LearntDynamics
        env = (self.eval_dynamics if self.sample_in == "eval_env"
               else self.train_dynamics)
        if not (isinstance(n, (Net, LSTM_NEW)) and n.conv and self.horizon == 10
                and hasattr(env, "params")
                and (isinstance(env, LearntDynamics)
                     or not isinstance(env, torch.nn.Module))):
            return None          # no fused evaluator for this architecture
        self.config.setdefault("thresh_div", self.thresh_div_start)
        self.config.setdefault("thresh_stable", self.thresh_stable_start)
        evaluator = QuadEvaluator(n, env, **{
            k: v for k, v in self.config.items()
            if k in ("ref_length", "dt", "speed_factor", "train_mode")})
        # self play (NetworkWrapper.predict_actions, take_every_x): visited
        # states go into the data set's self-play slots (concurrent mode: the
        # reference windows have the data set's length only there)
        self_play = (self.state_data
                     if getattr(self.state_data, "num_self_play", 0) > 0
                     and self.ref_length == self.horizon else None)
        with torch.no_grad():
            (suc_mean, suc_std, div_full_mean, div_full_std, div_mean,
             div_std) = evaluator.run_eval(
                "rand", nr_test=self.config.get("nr_test", 10),
                max_steps=self.config.get("max_steps", 251),
                thresh_div=self.config["thresh_div"],
                thresh_stable=self.config["thresh_stable"],
                dataset=self_play, take_every_x=self.self_play_every_x)
        self.sample_new_data(epoch)
        if epoch % 5 == 0 and self.config["thresh_div"] < self.thresh_div_end:
            self.config["thresh_div"] += .05
        self.save_model(epoch, suc_mean, suc_std)
        for key, val in (("mean_divergence_full", div_full_mean),
                         ("std_divergence_full", div_full_std),
                         ("mean_divergence", div_mean), ("std_divergence", div_std),
                         ("mean_success", suc_mean), ("std_success", suc_std),
                         ("thresh_div", self.config["thresh_div"])):
            self.results_dict[key].append(val)
        return suc_mean, suc_std

    def train_controller_model(
        self, current_state, action_seq, in_ref_states, ref_states
    ):
        self.optimizer_controller.zero_grad()
        if self.fused_learnt and self._fusable_learnt():
            # learnt simulator, frozen in this phase: the whole unroll through
            # LearntDynamics.forward + loss + backward to the actions in ONE
            # kernel (apg_quad_learnt_rollout_fwd_bwd); the simulator's own
            # parameters get no gradient here (optimizer_controller does not
            # own them, scripts/train_base.py:140-143)
            loss = F.quad_learnt_rollout_loss(
                self.train_dynamics, current_state, action_seq, ref_states,
                self.delta_t)
            return self._step(loss)
        if not self.analytic_train_dynamics():
            # learnt simulator: unroll through its own forward (:185-191)
            states = []
            for k in range(action_seq.size()[1]):
                current_state = self.train_dynamics(
                    current_state, action_seq[:, k], dt=self.delta_t)
                states.append(current_state)
            loss = quad_mpc_loss(torch.stack(states, dim=1), ref_states, action_seq)
            return self._step(loss)
        loss = F.quad_rollout_loss(
            current_state, action_seq, ref_states, self.delta_t,
            self.train_dynamics.params)
        return self._step(loss)


def train_control(base_model, config, device=None):
    """scripts/train_drone.py:241-257."""
    from .dynamics.quad_dynamics_flightmare import FlightmareDynamics
    modified_params = config["modified_params"]
    train_dynamics = FlightmareDynamics(modified_params=modified_params)
    eval_dynamics = FlightmareDynamics(modified_params=modified_params)
    config["sample_in"] = "train_env"
    trainer = TrainDrone(train_dynamics, eval_dynamics, config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_control(config)
    return trainer


def train_dynamics(base_model, config, device=None):
    """scripts/train_drone.py:260-278: fit LearntDynamics
    to the (modified) evaluation simulator, then train the controller through
    the learnt one."""
    from .dynamics.quad_dynamics_flightmare import FlightmareDynamics
    from .dynamics.quad_dynamics_trained import LearntDynamics
    modified_params = config["modified_params"]
    config["sample_in"] = "train_env"
    trainer = TrainDrone(LearntDynamics(), FlightmareDynamics(modified_params),
                         config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_dynamics(config)
    return trainer


def train_sampling_finetune(base_model, config, device=None):
    """scripts/train_drone.py:281-300: train in the nominal simulator on
    states visited in the modified one (self play samples from `eval_env`)."""
    from .dynamics.quad_dynamics_flightmare import FlightmareDynamics
    modified_params = config["modified_params"]
    config["sample_in"] = "eval_env"
    trainer = TrainDrone(FlightmareDynamics(),
                         FlightmareDynamics(modified_params=modified_params),
                         config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_control(config, sampling_based_finetune=True)
    return trainer
