"""Mirror of neural_control/models/hutter_model.py:6-49 (`Net`): state branch
Linear(state_dim, 64) + reference branch conv1d(ref_dim -> 20, k = 3) (or a
Linear over the flattened window), three 64-wide tanh layers, linear head."""
import torch
import torch.nn as nn


class Net(nn.Module):

    def __init__(self, state_dim, horizon, ref_dim, nr_actions_predict,
                 conv=True):
        super().__init__()
        self.horizon = horizon
        self.conv = conv
        self.reshape_len = 20 * (horizon - 2) if conv else 64
        self.states_in = nn.Linear(state_dim, 64)
        self.conv_ref = nn.Conv1d(ref_dim, 20, kernel_size=3)
        self.ref_in = nn.Linear(horizon * ref_dim, 64)  # used when conv=False
        self.fc1 = nn.Linear(64 + self.reshape_len, 64)
        self.fc2 = nn.Linear(64, 64)
        self.fc3 = nn.Linear(64, 64)
        self.fc_out = nn.Linear(64, nr_actions_predict)

    def trunk(self, state, ref):
        """Everything up to (not including) the output layer: [B,64]."""
        s = torch.tanh(self.states_in(state))
        if self.conv:
            r = torch.relu(self.conv_ref(ref.transpose(1, 2)))
            r = r.reshape(-1, self.reshape_len)
        else:
            r = torch.tanh(self.ref_in(ref))
        x = torch.cat((s, r), dim=1)
        x = torch.tanh(self.fc1(x))
        x = torch.tanh(self.fc2(x))
        return torch.tanh(self.fc3(x))

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
