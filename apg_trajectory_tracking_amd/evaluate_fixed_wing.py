"""Batched closed-loop evaluation of a fixed-wing controller - the role of
scripts/evaluate_fixed_wing.py:19-178 (`FixedWingEvaluator.fly_to_point`,
`run_eval`) with every test flight flown in parallel by one kernel launch
(apg_wing_mlp_closed_loop) instead of a Python loop of batch-1 policy /
dynamics calls per time step.  Outside SURVEY.md §8 (VERDICT r2 "what's
missing" #5): the caller of the fixed-wing hot path on the evaluation side.

Same names, arguments and statistics as the reference evaluator:
  * `fly_to_point(target_points, max_steps, return_traj)` returns the flown
    trajectory (state + action rows) or (div_target, div_to_linear) - for one
    flight ([n, 3] targets) exactly the reference's arrays, for a batch
    ([B, n, 3]) a list of them;
  * `run_eval(nr_test, ...)` draws its targets from numpy's global stream in
    the reference's order (np.random.rand(2) per flight, :143), so that
    `np.random.seed` makes both evaluators fly the same flights, and returns
    (mean, std) of the per-flight mean target error, or the per-flight means
    with return_dists.
There is no renderer and no MPC baseline here (out of scope)."""
import numpy as np
import torch

from . import functional as F


class FixedWingNetWrapper:
    """neural_control/controllers/network_wrapper.py:71-98: the policy, the
    data set whose mean / std / dt / horizon define its inputs and the
    self-play cadence (every take_every_x-th policy call, counted through all
    flights, hands its state and target to the data set)."""

    def __init__(self, model, dataset, horizon=1, take_every_x=1000, **kwargs):
        self.net = model
        self.dataset = dataset
        self.horizon = horizon
        self.action_dim = 4
        self.action_counter = 0
        self.take_every_x = take_every_x


class FixedWingEvaluator:

    def __init__(self, controller, env, dt=0.01, horizon=1, render=0,
                 thresh_div=10, thresh_stable=0.8, test_time=0, **kwargs):
        """controller: a FixedWingNetWrapper; env: the FixedWingDynamics the
        reference's SimpleWingEnv steps with (or an object with `.dynamics`);
        its `.params` are used - or, after train_dynamics() (scripts/
        train_fixed_wing.py:42-43), a LearntFixedWingDynamics module: the kernel
        then steps through its physics on the module's current parameters plus
        its residual network."""
        from .dynamics.fixed_wing_dynamics import LearntFixedWingDynamics
        if render:
            raise ValueError("there is no renderer on the GPU path")
        self.controller = controller
        self.dynamics = getattr(env, "dynamics", env)
        self.learnt = (self.dynamics
                       if isinstance(self.dynamics, LearntFixedWingDynamics) else None)
        if self.learnt is None and isinstance(self.dynamics, torch.nn.Module):
            raise TypeError(
                f"no closed-loop kernel steps through {type(self.dynamics).__name__}")
        self.dt = dt
        self.horizon = horizon
        self.thresh_div = thresh_div
        self.thresh_stable = thresh_stable
        self.des_speed = 11.5
        self.test_time = test_time

    # ------------------------------------------------------------- one launch
    def _closed_loop(self, targets, max_steps, want_trajectory):
        c = self.controller
        d = c.dataset
        dev = next(c.net.parameters()).device
        targets = torch.as_tensor(np.asarray(targets), dtype=torch.float32)
        return F.wing_mlp_closed_loop(
            c.net, targets.to(dev), self.dt, self.dynamics.params,
            d.mean.tolist(), d.std.tolist(), data_dt=d.dt,
            data_horizon=d.horizon, max_steps=max_steps,
            thresh_div=self.thresh_div, thresh_stable=self.thresh_stable,
            test_time=int(bool(self.test_time)),
            want_trajectory=want_trajectory, learnt=self.learnt)

    def _div_target(self, out, max_steps):
        """fly_to_point's div_target list per flight: what each step appended
        (a passed target first, then a divergence), and the entry of a flight
        that used up max_steps (:126-128)."""
        steps = out["steps"].cpu().numpy()
        ev = torch.stack((out["div_pass"], out["div_fail"]), 2)   # [T,B,2]
        ev = ev.permute(1, 0, 2).reshape(len(steps), -1).cpu().numpy()
        lists = []
        for i, n in enumerate(steps):
            row = ev[i, :2 * n]
            row = row[row >= 0].astype(np.float64).tolist()
            if n == max_steps:
                row.append(float(self.thresh_div))
            lists.append(np.array(row))
        return lists

    def _self_play(self, out):
        """FixedWingNetWrapper.predict_actions (:81-87): call number
        action_counter + 1, counted through the flights in order, goes to the
        data set when it is a multiple of take_every_x."""
        c = self.controller
        steps = out["steps"].to(torch.int64)
        total = int(steps.sum())
        data = c.dataset
        if getattr(data, "num_self_play", 0) > 0 and total > 0 and "seen" in out:
            dev = steps.device
            T = out["seen"].shape[0]
            first = c.action_counter + torch.cumsum(steps, 0) - steps
            k = torch.arange(T, device=dev)[:, None]
            pick = ((first[None] + k + 1) % c.take_every_x == 0) & (k < steps[None])
            ii, kk = torch.nonzero(pick.t(), as_tuple=True)       # flight-major
            if kk.numel():
                seen = out["seen"][kk, :, ii]                     # [n,15]
                data.add_eval_data(seen[:, :12], seen[:, 12:])
        c.action_counter += total

    def _fly(self, targets, max_steps, return_traj=False):
        """One launch for all flights + the self play that goes with it."""
        # the per-step rows (31 floats x max_steps per flight) only when they
        # are returned or feed the self play
        collecting = getattr(self.controller.dataset, "num_self_play", 0) > 0
        out = self._closed_loop(targets, max_steps, return_traj or collecting)
        self._self_play(out)
        return out

    def fly_to_point(self, target_points, max_steps=1000, do_avg_act=0,
                     return_traj=False):
        targets = np.asarray(target_points, dtype=np.float32)
        single = targets.ndim == 2
        if single:
            targets = targets[None]
        out = self._fly(targets, max_steps, return_traj)
        steps = out["steps"].cpu().numpy()
        if return_traj:
            drone = out["drone"].permute(2, 0, 1).cpu().numpy()
            trajs = [drone[i, :n] for i, n in enumerate(steps)]
            return trajs[0] if single else trajs
        div_target = self._div_target(out, max_steps)
        lin = out["div_linear"].t().cpu().numpy().astype(np.float64)
        div_linear = [lin[i, :n] for i, n in enumerate(steps)]
        if single:
            return div_target[0], div_linear[0]
        return div_target, div_linear

    def run_eval(self, nr_test, return_dists=False, x_dist=50, x_std=5,
                 printout=True, max_steps=1000):
        """scripts/evaluate_fixed_wing.py:133-178, all flights in one launch."""
        yz = (np.random.rand(nr_test, 2) - .5) * 2 * x_std
        targets = np.concatenate(
            (np.full((nr_test, 1, 1), float(x_dist)), yz[:, None]), 2)
        out = self._fly(targets.astype(np.float32), max_steps)
        # per-flight mean of fly_to_point's div_target list, without building
        # the lists: the entries of the steps flown, plus thresh_div for a
        # flight that used up max_steps (:126-128)
        steps = out["steps"].to(torch.int64)
        ev = torch.stack((out["div_pass"], out["div_fail"]), 2).double()  # [T,B,2]
        flown = torch.arange(ev.shape[0], device=ev.device)[:, None] < steps[None]
        valid = (ev >= 0) & flown[:, :, None]
        total = torch.where(valid, ev, torch.zeros_like(ev)).sum((0, 2))
        count = valid.sum((0, 2))
        cut = steps == max_steps
        total = total + cut * float(self.thresh_div)
        count = count + cut
        mean_div_target = (total / count).cpu().numpy()
        not_div_time = steps.cpu().numpy()
        mean_err, std_err = np.mean(mean_div_target), np.std(mean_div_target)
        if printout:
            print("Time not diverged: %3.2f (%3.2f)"
                  % (np.mean(not_div_time), np.std(not_div_time)))
            print("Average error (target): %3.2f (%3.2f)" % (mean_err, std_err))
        if return_dists:
            return mean_div_target
        return mean_err, std_err
