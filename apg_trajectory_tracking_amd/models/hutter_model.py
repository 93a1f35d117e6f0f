"""Mirror of neural_control/models/hutter_model.py:6-49 (`Net`): state branch
Linear(state_dim, 64) + reference branch conv1d(ref_dim -> 20, k = 3) (or a
Linear over the flattened window), three 64-wide tanh layers, linear head."""
import torch
import torch.nn as nn

from ..nn import Linear

from .common import (CONV_CHANNELS, CONV_KERNEL, DENSE_WIDTH, encode_window,
                     window_feature_count)


class Net(nn.Module):

    def __init__(self, state_dim, horizon, ref_dim, nr_actions_predict,
                 conv=True):
        super().__init__()
        W = DENSE_WIDTH
        self.horizon, self.conv = horizon, conv
        self.reshape_len = window_feature_count(horizon, conv)
        # registration order = the reference's state_dict order
        self.states_in = Linear(state_dim, W)
        self.conv_ref = nn.Conv1d(ref_dim, CONV_CHANNELS, kernel_size=CONV_KERNEL)
        self.ref_in = Linear(horizon * ref_dim, W)     # conv=False branch
        self.fc1 = Linear(W + self.reshape_len, W)
        self.fc2, self.fc3 = Linear(W, W), Linear(W, W)
        self.fc_out = Linear(W, nr_actions_predict)

    def trunk(self, state, ref):
        """Everything up to (not including) the output layer: [B,64]."""
        x = torch.cat((torch.tanh(self.states_in(state)),
                       encode_window(self, ref)), dim=1)
        for layer in (self.fc1, self.fc2, self.fc3):
            x = torch.tanh(layer(x))
        return x

    def forward(self, state, ref):
        """state [B,state_dim], ref [B,horizon,ref_dim] -> [B,nr_actions]."""
        return self.fc_out(self.trunk(state, ref))

    def forward_soa(self, state, ref):
        """Same network, output transposed: [nr_actions, B].  The head GEMM is
        evaluated as W h^T + b, so the concurrent-mode action sequence comes
        out of rocBLAS already in the device-native SoA layout
        ([H][4][B] after a free reshape) that the fused rollout kernel reads
        fully coalesced - no transpose pass in either direction."""
        h = self.trunk(state, ref)
        return torch.addmm(self.fc_out.bias[:, None], self.fc_out.weight, h.t())

    def forward_packed(self, state, ref):
        """Same network, output as action ROWS: [H, B, nr_actions // H] - the
        `APG_LAYOUT_PACKED` action tensor of the fused rollout (include/apg.h).
        Plain head GEMM + ONE transposing copy (10 MB at B = 65 536, ~15 us;
        its backward is the same copy).  Evaluating the head with the horizon
        as the batch of the GEMM (`baddbmm`, no copy) was measured and dropped:
        rocBLAS runs the ten [B x 64] x [64 x 4] products and their weight
        gradients 250 us slower than the one wide GEMM
        (profiles/r03_packed_step_timeline.txt)."""
        H = self.horizon
        out = self.forward(state, ref)
        return out.view(out.shape[0], H, -1).transpose(0, 1).contiguous()
