"""Mirror of neural_control/models/rnn.py:8-51 (`LSTM_NEW`): the reference
window goes through conv1d(ref_dim -> 20, k = 3), is concatenated with the
state features and fed to an LSTMCell with 8 hidden units; a linear head maps
the hidden state to one action.  The hidden / cell state is carried across
the steps of one unroll and re-drawn from N(0,1) by reset_hidden_state()
(rnn.py:30-33) - pass `generator=` for reproducible draws."""
import torch
import torch.nn as nn


class LSTM_NEW(nn.Module):

    def __init__(self, state_dim, horizon, ref_dim, nr_actions_predict,
                 conv=True):
        super().__init__()
        self.state_dim = state_dim
        self.ref_dim = ref_dim
        self.horizon = horizon
        self.conv = conv
        self.reshape_len = 20 * (horizon - 2) if conv else 64
        self.conv_ref = nn.Conv1d(ref_dim, 20, kernel_size=3)
        self.ref_in = nn.Linear(horizon * ref_dim, 64)
        self.fc_out = nn.Linear(8, nr_actions_predict)
        self.lstm = nn.LSTMCell(state_dim + self.reshape_len, 8)
        self.hidden_state = None
        self.cell_state = None
        self.reset_hidden_state(1)

    def reset_hidden_state(self, batch_size=1, generator=None):
        dev = self.fc_out.weight.device
        if generator is None:
            self.hidden_state = torch.randn(batch_size, 8, device=dev)
            self.cell_state = torch.randn(batch_size, 8, device=dev)
        else:  # draw on the generator's device, then move
            gdev = generator.device
            self.hidden_state = torch.randn(
                batch_size, 8, generator=generator, device=gdev).to(dev)
            self.cell_state = torch.randn(
                batch_size, 8, generator=generator, device=gdev).to(dev)

    def forward(self, state, ref):
        if self.conv:
            r = torch.relu(self.conv_ref(ref.transpose(1, 2)))
            r = r.reshape(-1, self.reshape_len)
        else:
            r = torch.tanh(self.ref_in(ref))
        x = torch.cat((state, r), dim=1)
        self.hidden_state, self.cell_state = self.lstm(
            x, (self.hidden_state, self.cell_state))
        return self.fc_out(self.hidden_state)
