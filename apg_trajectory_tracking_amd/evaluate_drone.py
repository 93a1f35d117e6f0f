"""Batched closed-loop evaluation of a quadrotor controller - the role of
scripts/evaluate_drone.py:30-300 (`QuadEvaluator.follow_trajectory`,
`run_eval`) for the "rand" reference, with every test trajectory tracked in
parallel by one kernel launch (apg_quad_mlp_closed_loop) instead of a Python
loop of batch-1 policy / dynamics calls per time step.

Same names and statistics as the reference evaluator:
  * `follow_trajectory` returns (reference_trajectory, drone_trajectory,
    divergences, actions) - here with a leading batch axis / per-trajectory
    lists;
  * `run_eval` returns (mean(stable), std(stable), mean / std of the tracking
    error of the complete runs, mean / std of the tracking error), where
    stable = number of steps with divergence < thresh_div and "complete" is
    measured against the length of the last run, as in the reference
    (:266-299).
There is no renderer and no MPC baseline here (out of scope)."""
import numpy as np
import torch

from . import functional as F
from . import synthetic


class QuadEvaluator:

    def __init__(self, controller, environment, ref_length=10, dt=0.1,
                 test_time=0, speed_factor=.6, train_mode="concurrent",
                 trajectory_length=501, **kwargs):
        """controller: the policy (hutter_model.Net or rnn.LSTM_NEW, conv
        branch) or an object with a `.net`; environment: the dynamics the
        reference's eval_env steps with - FlightmareDynamics (its `.params` are
        used) or, after train_dynamics() swapped it in (scripts/train_drone.py:
        44-45), a LearntDynamics module: the kernel then steps through the
        action transform, the analytic step and the residual network."""
        from .dynamics.quad_dynamics_trained import LearntDynamics
        self.net = getattr(controller, "net", controller)
        self.dynamics = getattr(environment, "dynamics", environment)
        self.learnt = (self.dynamics if isinstance(self.dynamics, LearntDynamics)
                       else None)
        if self.learnt is None and isinstance(self.dynamics, torch.nn.Module):
            raise TypeError(
                f"no closed-loop kernel steps through {type(self.dynamics).__name__}")
        if ref_length != 10:
            raise ValueError("the fused evaluator is built for horizon 10")
        self.horizon = ref_length
        self.dt = dt
        self.test_time = test_time
        self.speed_factor = speed_factor
        self.train_mode = train_mode
        self.trajectory_length = trajectory_length
        self.hidden = None     # (h0, c0) [B,8] for an LSTM controller

    def _closed_loop(self, traj, **kw):
        """One launch for the whole batch; MLP or LSTM controller."""
        if hasattr(self.net, "lstm"):
            B, dev = traj.shape[0], traj.device
            if self.hidden is None or self.hidden[0].shape[0] != B:
                # reset_hidden_state (rnn.py:30-33): standard normal draws
                self.hidden = (torch.randn(B, 8, device=dev),
                               torch.randn(B, 8, device=dev))
            return F.quad_lstm_closed_loop(
                self.net, traj, self.dt, self.dynamics.params, self.hidden[0],
                self.hidden[1], learnt=self.learnt, **kw)
        return F.quad_mlp_closed_loop(self.net, traj, self.dt,
                                      self.dynamics.params, learnt=self.learnt, **kw)

    def reference_batch(self, nr_test, seed=None):
        """Counterpart of `Random.__init__` (random_traj.py:28-35): nr_test
        trajectories, lifted by 3 m.  Like the reference (unseeded numpy
        draws per run) every call gives NEW trajectories: `seed=None` takes
        the seed from torch's global generator, so `torch.manual_seed` makes
        an evaluation reproducible."""
        if seed is None:
            seed = int(torch.randint(2**31 - 1, (1,)))
        traj = synthetic.quad_eval_trajectories(
            nr_test, self.trajectory_length, self.dt, seed=seed,
            speed=self.speed_factor / .6)
        traj[:, :, 2] += 3
        return traj

    def follow_trajectory(self, traj_type="rand", max_nr_steps=200,
                          thresh_stable=.4, thresh_div=3, trajectories=None,
                          nr_test=1, seed=None, **traj_args):
        if traj_type != "rand":
            raise ValueError("only the 'rand' reference is evaluated on the GPU")
        dev = next(self.net.parameters()).device
        traj = (self.reference_batch(nr_test, seed) if trajectories is None
                else trajectories).to(dev)
        out = self._closed_loop(
            traj, max_steps=max_nr_steps, thresh_div=thresh_div,
            thresh_stable=thresh_stable, test_time=self.test_time,
            want_trajectory=True)
        steps = out["steps"].cpu().numpy()
        T = out["div"].shape[0]
        # projected reference: row min(k + 1, L - H) of the trajectory
        L = traj.shape[1]
        idx = torch.clamp(torch.arange(1, T + 1, device=dev), max=L - self.horizon)
        ref_tr = traj[:, idx, :3]
        drone = out["drone"].permute(2, 0, 1)
        divs = out["div"].t()
        acts = out["actions"].permute(2, 0, 1)
        return ([ref_tr[i, :n] for i, n in enumerate(steps)],
                [drone[i, :n + 1] for i, n in enumerate(steps)],
                [divs[i, :n] for i, n in enumerate(steps)],
                [acts[i, :n] for i, n in enumerate(steps)])

    def add_self_play_data(self, dataset, traj, out, steps, take_every_x):
        """The (state, window) pairs of every take_every_x-th policy call."""
        dev = traj.device
        T, L, H = out["div"].shape[0], traj.shape[1], self.horizon
        first = torch.cumsum(steps, 0) - steps               # calls before run i
        k = torch.arange(T, device=dev)[:, None]             # [T,1]
        pick = ((first[None] + k + 1) % take_every_x == 0) & (k < steps[None])
        kk, ii = torch.nonzero(pick.t(), as_tuple=True)[::-1]  # run-major order
        if kk.numel() == 0:
            return 0
        states = out["start_states"][kk, :, ii]              # [n,12]
        ws = torch.clamp(kk + 1, max=L - H)                  # get_ref_traj window
        rows = ws[:, None] + torch.arange(H, device=dev)[None]
        windows = traj[ii[:, None], rows]                    # [n,H,9]
        return dataset.add_eval_data(states, windows)

    def run_eval(self, reference="rand", nr_test=10, max_steps=251,
                 thresh_div=1, thresh_stable=1, return_dict=False,
                 trajectories=None, dataset=None, take_every_x=1000, seed=None,
                 **kwargs):
        """scripts/evaluate_drone.py:237-300, all runs in one launch.
        `dataset` (optional, with `add_eval_data`): self play - every
        take_every_x-th policy call, counted through the runs in order like
        NetworkWrapper.action_counter (network_wrapper.py:47-71), hands its
        (state, reference window) to the data set."""
        if nr_test == 0:
            return 0, 0
        if reference != "rand":
            raise ValueError("only the 'rand' reference is evaluated on the GPU")
        dev = next(self.net.parameters()).device
        traj = (self.reference_batch(nr_test, seed) if trajectories is None
                else trajectories).to(dev)
        with torch.no_grad():
            out = self._closed_loop(
                traj, max_steps=max_steps, thresh_div=thresh_div,
                thresh_stable=thresh_stable, test_time=self.test_time,
                want_trajectory=dataset is not None)
        steps = out["steps"].to(torch.int64)
        T = out["div"].shape[0]
        if dataset is not None:
            self.add_self_play_data(dataset, traj, out, steps, take_every_x)
        valid = torch.arange(T, device=dev)[:, None] < steps[None]
        divs = torch.where(valid, out["div"], torch.zeros_like(out["div"]))
        div = (divs.sum(0) / steps.clamp(min=1)).cpu().numpy().astype(np.float64)
        stable = (valid & (out["div"] < thresh_div)).sum(0).cpu().numpy()
        max_steps_stable = int(steps[-1])      # len(reference_traj) of the last run
        ratio_stable = np.sum(stable == max_steps_stable) / len(stable)
        full = div[stable == max_steps_stable]
        if return_dict:
            return {"avg_tracking_error": np.mean(full),
                    "std_tracking_error": np.std(full),
                    "ratio_stable": ratio_stable}
        return (np.mean(stable), np.std(stable), np.mean(full), np.std(full),
                np.mean(div), np.std(div))
