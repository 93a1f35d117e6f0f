"""Drop-in for scripts/train_fixed_wing.py:20-116 (`TrainFixedWing`)
restricted to the APG hot path: `train_controller_model` (:90-116) as one
fused HIP launch (H x fixed-wing dynamics at dt = delta_t_train +
fixed_wing_mpc_loss + analytic adjoint)."""
import torch

from . import functional as F
from .dataset import SyntheticWingDataset
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
            n = int(self.epoch_size * (1 + self.self_play))
            state_data = SyntheticWingDataset(
                n, self.horizon, self.delta_t_train, seed=seed, device=device)
        self.state_data = state_data
        self.config["mean"] = self.state_data.mean.tolist()
        self.config["std"] = self.state_data.std.tolist()
        self.config["dt"] = self.delta_t
        self.init_optimizer()

    fused_policy = True   # policy on the matrix cores around the fused rollout

    def train_concurrent_fused(
        self, in_state, current_state, in_ref_states, ref_states, index=None,
        probe=False
    ):
        """scripts/train_base.py:198-204 + scripts/train_fixed_wing.py:90-116
        with the policy inside HIP kernels (functional.wing_concurrent_policy_grads)."""
        n = self.net
        if not (self.fused_policy and isinstance(n, Net) and not n.conv
                and self.horizon == 20 and hasattr(self.train_dynamics, "params")
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
        loss = F.wing_rollout_loss(
            current_state, action_seq, ref_states, self.delta_t_train,
            self.train_dynamics.params)
        return self._step(loss)
