"""Drop-in for scripts/train_fixed_wing.py:20-197 (`TrainFixedWing`):
`train_controller_model` (:90-116) as one fused HIP launch (H x fixed-wing
dynamics at dt = delta_t_train + fixed_wing_mpc_loss + analytic adjoint) and
`evaluate_model` (:142-197) on the batched closed-loop evaluator
(evaluate_fixed_wing.FixedWingEvaluator: all test flights in one launch),
with self play into the data set and the threshold curriculum."""
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
                         state_data=None, device=None, seed=0):
        device = torch.device(device or "cuda")
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
        n = self.net
        env = (self.eval_dynamics if self.sample_in == "eval_env"
               else self.train_dynamics)
        if isinstance(env, torch.nn.Module):
            env = self.eval_dynamics
        if not (isinstance(n, Net) and not n.conv and hasattr(env, "params")
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
                and self.horizon == 20 and self.analytic_train_dynamics()
                and n.states_in.weight.shape == (64, 9)
                and n.ref_in.weight.shape == (64, 3)
                and n.fc1.weight.shape == (64, 128)
                and n.fc_out.weight.shape == (80, 64)):
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
