"""Drop-in for scripts/train_fixed_wing.py:20-197 (`TrainFixedWing`):
`train_controller_model` (:90-116) as one fused HIP launch (H x fixed-wing
dynamics at dt = delta_t_train + fixed_wing_mpc_loss + analytic adjoint) and
`evaluate_model` (:142-197) on the batched closed-loop evaluator
(evaluate_fixed_wing.FixedWingEvaluator: all test flights in one launch),
with self play into the data set and the threshold curriculum."""
import json
import os

import torch

from . import functional as F
from .dataset import SyntheticWingDataset
from .drone_loss import fixed_wing_mpc_loss
from .models.hutter_model import Net
from .train_base import TrainBase


class TrainFixedWing(TrainBase):

    def __init__(self, train_dynamics, eval_dynamics, config):
        self.config = config
        super().__init__(train_dynamics, eval_dynamics, **config)
        if self.train_mode != "concurrent":
            raise ValueError(
                "autoregressive / LSTM training is only implemented "
                "for the Quadrotor! Use concurrent as train mode"
            )
        if self.sample_in not in ("eval_env", "train_env"):
            raise ValueError("sample in must be one of eval_env, train_env")

    def initialize_model(self, base_model=None, modified_params={},
                         state_data=None, device=None, seed=0,
                         base_model_name="model_wing"):
        """Policy + data set + optimizer (scripts/train_fixed_wing.py:46-88).
        `base_model`: a module, or - as in the reference - the directory of a
        trained model (`<dir>/model_wing`, here a state_dict checkpoint, see
        checkpoint.py; its config.json supplies mean / std, :56-63).  The run's
        parameters are written to `<save_path>/config.json` like the reference
        does."""
        device = torch.device(device or "cuda")
        if isinstance(base_model, (str, os.PathLike)):
            from .checkpoint import load_policy
            path = os.path.join(base_model, "config.json")
            if not os.path.exists(path):
                path = os.path.join(base_model, "param_dict.json")
            with open(path) as f:
                previous = json.load(f)
            self.config["mean"] = previous["mean"]
            self.config["std"] = previous["std"]
            base_model = load_policy(os.path.join(base_model, base_model_name),
                                     system="wing")
        if isinstance(self.train_dynamics, torch.nn.Module):
            self.train_dynamics.to(device)     # learnable simulator
        if base_model is not None:
            self.net = base_model
        else:
            # state without position (12 - 3) in, H x 4 actions out (:68-76)
            self.net = Net(
                self.state_size - self.ref_dim, 1, self.ref_dim,
                self.action_dim * self.horizon, conv=False)
        self.net.to(device)
        if state_data is None:
            # epoch_size sampled states + int(self_play * epoch_size) slots for
            # the states the evaluation flights visit (WingDataset(epoch_size,
            # **config), :80; DroneDataset.__init__, dataset.py:48-56)
            state_data = SyntheticWingDataset(
                self.epoch_size, self.horizon, self.delta_t_train, seed=seed,
                device=device, self_play=self.self_play,
                mean=self.config.get("mean"), std=self.config.get("std"))
        self.state_data = state_data
        self.config["mean"] = self.state_data.mean.tolist()
        self.config["std"] = self.state_data.std.tolist()
        self.config["thresh_div"] = self.thresh_div_start
        self.config["dt"] = self.delta_t
        self.config["take_every_x"] = self.self_play_every_x
        self.config["thresh_stable"] = self.thresh_stable_start
        from . import parallel
        if parallel.is_main():       # one writer under torch.distributed
            os.makedirs(self.save_path, exist_ok=True)
            with open(os.path.join(self.save_path, "config.json"), "w") as f:
                json.dump(self.config, f, default=str)
        self.init_optimizer()

    # flights per launch while the self-play slots are first filled (the
    # reference flies them five at a time, :156-160; here a launch is one batch)
    self_play_flights = 64

    def evaluate_model(self, epoch):
        """scripts/train_fixed_wing.py:142-197: flights with self play (at
        epoch 0 until `self_play` states were collected), two test_time
        flights for the score, resampling, the divergence / stability
        threshold ladders, checkpoint, statistics."""
        from .evaluate_fixed_wing import FixedWingEvaluator, FixedWingNetWrapper
        from .dynamics.fixed_wing_dynamics import LearntFixedWingDynamics
        n = self.net
        env = (self.eval_dynamics if self.sample_in == "eval_env"
               else self.train_dynamics)
        # (after train_dynamics() `env` is the LEARNT simulator, residual network
        # included, scripts/train_fixed_wing.py:42-43: the closed-loop kernel
        # steps through it, csrc/learnt_residual.h)
        if not (isinstance(n, Net) and not n.conv and hasattr(env, "params")
                and (isinstance(env, LearntFixedWingDynamics)
                     or not isinstance(env, torch.nn.Module))
                and hasattr(self.state_data, "add_eval_data")
                and n.fc1.weight.shape == (64, 128)):
            return None          # no fused evaluator for this architecture
        keys = ("dt", "horizon", "thresh_div", "thresh_stable", "take_every_x")
        cfg = {k: self.config[k] for k in keys if k in self.config}
        controller = FixedWingNetWrapper(n, self.state_data, **cfg)
        evaluator = FixedWingEvaluator(controller, env, **cfg)
        data = self.state_data
        with torch.no_grad():
            if epoch == 0 and data.num_self_play > 0:
                goal = data.eval_counter + int(self.config.get(
                    "self_play", data.num_self_play))
                while data.eval_counter < goal:
                    before = data.eval_counter
                    evaluator.run_eval(nr_test=self.self_play_flights,
                                       printout=False)
                    if data.eval_counter == before:
                        break          # nothing is being collected
            evaluator.run_eval(nr_test=10, printout=False)
            evaluator_test = FixedWingEvaluator(
                controller, env, **dict(cfg, test_time=True))
            suc_mean, suc_std = evaluator_test.run_eval(nr_test=2, printout=False)
        self.sample_new_data(epoch)
        if epoch % 5 == 0 and self.config["thresh_div"] < self.thresh_div_end:
            self.config["thresh_div"] += .2
        if epoch % 5 == 0 and self.config["thresh_stable"] < self.thresh_stable_end:
            self.config["thresh_stable"] += .05
        self.save_model(epoch, suc_mean, suc_std)
        self.results_dict["mean_success"].append(suc_mean)
        self.results_dict["std_success"].append(suc_std)
        self.results_dict["thresh_div"].append(self.config["thresh_div"])
        return suc_mean, suc_std

    fused_policy = True   # policy on the matrix cores around the fused rollout

    def train_concurrent_fused(
        self, in_state, current_state, in_ref_states, ref_states, index=None,
        probe=False
    ):
        """scripts/train_base.py:198-204 + scripts/train_fixed_wing.py:90-116
        with the policy inside HIP kernels (functional.wing_concurrent_policy_grads)."""
        n = self.net
        if not (self.fused_policy and isinstance(n, Net) and not n.conv
                and self.horizon in (10, 20) and self.analytic_train_dynamics()
                and n.states_in.weight.shape == (64, 9)
                and n.ref_in.weight.shape == (64, 3)
                and n.fc1.weight.shape == (64, 128)
                and n.fc_out.weight.shape == (4 * self.horizon, 64)):
            return False if probe else None
        if probe:
            return True
        loss, grads, flat = F.wing_concurrent_policy_grads(
            n, in_state, in_ref_states, current_state, ref_states,
            self.delta_t_train, self.train_dynamics.params, index=index)
        return self._step_direct(loss, grads, flat)

    def train_controller_model(
        self, current_state, action_seq, in_ref_state, ref_states
    ):
        self.optimizer_controller.zero_grad()
        if not self.analytic_train_dynamics():
            # learnt simulator (LearntFixedWingDynamics): unroll through its
            # own forward, step by step, as the reference does (:94-106)
            states = []
            for k in range(action_seq.size()[1]):
                current_state = self.train_dynamics(
                    current_state, action_seq[:, k], dt=self.delta_t_train)
                states.append(current_state)
            loss = fixed_wing_mpc_loss(
                torch.stack(states, dim=1), ref_states, action_seq)
            return self._step(loss)
        loss = F.wing_rollout_loss(
            current_state, action_seq, ref_states, self.delta_t_train,
            self.train_dynamics.params)
        return self._step(loss)


def train_control(base_model, config, device=None):
    """scripts/train_fixed_wing.py:200-215: train a controller from scratch or
    from `base_model` (a policy module); self play samples come from the
    training environment."""
    from .dynamics.fixed_wing_dynamics import FixedWingDynamics
    modified_params = config["modified_params"]
    train_dynamics = FixedWingDynamics(modified_params)
    eval_dynamics = FixedWingDynamics(modified_params)
    config["sample_in"] = "train_env"
    trainer = TrainFixedWing(train_dynamics, eval_dynamics, config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_control(config, curriculum=0)
    return trainer


def train_dynamics(base_model, config, device=None):
    """scripts/train_fixed_wing.py:218-241: fit LearntFixedWingDynamics to the
    (modified) evaluation simulator, then train the controller through the
    learnt one; thresholds high so that the tracking error is reliable."""
    from .dynamics.fixed_wing_dynamics import (
        FixedWingDynamics, LearntFixedWingDynamics)
    modified_params = config["modified_params"]
    config["sample_in"] = "train_env"
    config["thresh_div_start"] = 20
    config["thresh_stable_start"] = 1.5
    learnt = LearntFixedWingDynamics()
    if device is not None or torch.cuda.is_available():
        learnt = learnt.to(torch.device(device or "cuda"))
    trainer = TrainFixedWing(
        learnt, FixedWingDynamics(modified_params=modified_params), config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_dynamics(config)
    return trainer


def train_sampling_finetune(base_model, config, device=None):
    """scripts/train_fixed_wing.py:244-262: train in the nominal simulator on
    states visited in the modified one (self play samples from `eval_env`)."""
    from .dynamics.fixed_wing_dynamics import FixedWingDynamics
    modified_params = config["modified_params"]
    config["sample_in"] = "eval_env"
    trainer = TrainFixedWing(
        FixedWingDynamics(), FixedWingDynamics(modified_params=modified_params),
        config)
    trainer.initialize_model(base_model, modified_params=modified_params,
                             device=device)
    trainer.run_control(config, sampling_based_finetune=True)
    return trainer
