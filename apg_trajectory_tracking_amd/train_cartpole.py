"""Drop-in for scripts/train_cartpole.py:30-165 (`TrainCartpole`) restricted
to the APG hot path: `run_epoch` (:118-165) with `make_reference` (:103-110).
The controller branch is one fused HIP launch (make_reference + H x cartpole
dynamics + cartpole_loss_mpc + adjoint).  Quirks kept: the policy ends in
tanh with NO sigmoid (:127-130), `simple_model.Net` zeroes column 0 of its
input in place, run_epoch has no `epoch` argument and divides by the last
batch index (:163)."""
import torch

from . import functional as F
from .dataset import SyntheticCartpoleDataset
from .models.simple_model import Net
from .train_base import TrainBase


class TrainCartpole(TrainBase):

    def __init__(self, train_dynamics, eval_dynamics, config,
                 train_image_dyn=0, train_seq_dyn=0, swingup=0):
        self.swingup = swingup
        self.config = config
        super().__init__(train_dynamics, eval_dynamics, **self.config)
        if self.sample_in not in ("eval_env", "train_env"):
            raise ValueError("sample in must be one of eval_env, train_env")
        if train_image_dyn or train_seq_dyn:
            raise NotImplementedError(
                "image / sequence dynamics are outside the APG hot path")
        if self.train_mode != "concurrent":
            raise ValueError(
                "autoregressive / LSTM training is only implemented "
                "for the Quadrotor! Use concurrent as train mode"
            )

    def initialize_model(self, base_model=None, state_data=None, device=None,
                         seed=0):
        device = torch.device(device or "cuda")
        self.net = base_model if base_model is not None else Net(
            self.state_size, self.horizon * self.action_dim)
        self.net.to(device)
        if state_data is None:
            state_data = SyntheticCartpoleDataset(
                int(self.config.get("sample_data", 1000)), seed=seed,
                device=device)
        self.state_data = state_data
        self.init_optimizer()

    def dataset_tensors(self):
        return (self.state_data.states, self.state_data.labels)

    def make_reference(self, current_state):
        """ref_k = s0 * (1 - k/(H-1)) for k < H-1, last row zero (:103-110)."""
        ref_states = torch.zeros(
            current_state.size()[0], self.horizon, self.state_size,
            device=current_state.device)
        for k in range(self.horizon - 1):
            ref_states[:, k] = (
                current_state * (1 - 1 / (self.horizon - 1) * k))
        return ref_states

    def run_epoch(self, train="controller"):
        if train != "controller":
            raise NotImplementedError(
                "learnt-dynamics training is outside the APG hot path")
        self.results_dict["trained"].append(train)
        running_loss = None
        i = -1
        for i, data in enumerate(self.trainloader, 0):
            in_state, current_state = data
            # the policy zeroes column 0 of its input in place: hand it a
            # private copy, as DataLoader collation does in the reference
            actions = self.net(in_state.clone())  # tanh output, no sigmoid
            action_seq = torch.reshape(
                actions, (-1, self.horizon, self.action_dim))
            self.optimizer_controller.zero_grad()
            loss = F.cartpole_rollout_loss(
                current_state, action_seq, self.delta_t,
                self.train_dynamics.params)
            loss = self._step(loss).detach()
            running_loss = loss if running_loss is None else running_loss + loss
        epoch_loss = float(running_loss.item()) / i
        self.results_dict["loss_" + train].append(epoch_loss)
        print(f"Loss ({train}): {round(epoch_loss, 2)}")
        return epoch_loss
