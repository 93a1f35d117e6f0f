"""Mirror of neural_control/models/simple_model.py:9-28 (cartpole `Net`):
five tanh layers 4-32-64-64-32-out.  Quirk kept: forward() zeroes column 0 of
its INPUT in place (simple_model.py:21) - the cart position is hidden from
the policy and the caller's tensor is modified."""
import torch
import torch.nn as nn

from ..nn import Linear


class Net(nn.Module):

    def __init__(self, in_size, out_size):
        super().__init__()
        self.fc0 = Linear(in_size, 32)
        self.fc1 = Linear(32, 64)
        self.fc2 = Linear(64, 64)
        self.fc3 = Linear(64, 32)
        self.fc_out = Linear(32, out_size)

    def forward(self, x):
        x[:, 0] *= 0
        for layer in (self.fc0, self.fc1, self.fc2, self.fc3, self.fc_out):
            x = torch.tanh(layer(x))
        return x
