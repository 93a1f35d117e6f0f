"""Mirror of neural_control/models/rnn.py:8-51 (`LSTM_NEW`): the reference
window goes through conv1d(ref_dim -> 20, k = 3), is concatenated with the
state features and fed to an LSTMCell with 8 hidden units; a linear head maps
the hidden state to one action.  The hidden / cell state is carried across
the steps of one unroll and re-drawn from N(0,1) by reset_hidden_state()
(rnn.py:30-33) - pass `generator=` for reproducible draws."""
import torch
import torch.nn as nn

from ..nn import Linear

from .common import (CONV_CHANNELS, CONV_KERNEL, DENSE_WIDTH, encode_window,
                     window_feature_count)

HIDDEN = 8     # LSTM units


class LSTM_NEW(nn.Module):

    def __init__(self, state_dim, horizon, ref_dim, nr_actions_predict,
                 conv=True):
        super().__init__()
        self.state_dim, self.ref_dim = state_dim, ref_dim
        self.horizon, self.conv = horizon, conv
        self.reshape_len = window_feature_count(horizon, conv)
        # registration order = the reference's state_dict order
        self.conv_ref = nn.Conv1d(ref_dim, CONV_CHANNELS, kernel_size=CONV_KERNEL)
        self.ref_in = Linear(horizon * ref_dim, DENSE_WIDTH)
        self.fc_out = Linear(HIDDEN, nr_actions_predict)
        self.lstm = nn.LSTMCell(state_dim + self.reshape_len, HIDDEN)
        self.hidden_state = self.cell_state = None
        self.hidden_pool, self._pool = 16, None    # (device draws: see reset_hidden_state)
        self.reset_hidden_state(1)

    def reset_hidden_state(self, batch_size=1, generator=None):
        """Fresh (h, c) ~ N(0, 1), [batch_size, 8] each, h drawn first."""
        dev = self.fc_out.weight.device
        # with a generator: draw on ITS device, then move
        where = dev if generator is None else generator.device
        if torch.device(where).type == "cpu":
            # the reference's two draws from a CPU stream, value for value
            draw = lambda: torch.randn(batch_size, HIDDEN, generator=generator,
                                       device=where).to(dev)
            self.hidden_state = draw()
            self.cell_state = draw()
        else:
            # device stream (no draw-for-draw counterpart): one launch for both,
            # drawn in the plane layout the fused kernels read ([8][B]; the
            # [B, 8] tensors are transposed views, iid entries either way) -
            # no layout change per step.  Round 6: `hidden_pool` batches' worth
            # are drawn by ONE launch and handed out step by step (the draw was
            # 7 us of a 450 us step, every step); not under stream capture,
            # where a replay must draw for itself.
            K = int(self.hidden_pool)
            if K > 1 and not (torch.device(where).type == "cuda"
                              and torch.cuda.is_current_stream_capturing()):
                key = (batch_size, torch.device(where), torch.device(dev), id(generator))
                pool = self._pool
                if pool is None or pool[0] != key or pool[2] >= K:
                    pool = self._pool = [key, torch.randn(
                        K, 2, HIDDEN, batch_size, generator=generator, device=where).to(dev), 0]
                both = pool[1][pool[2]]
                pool[2] += 1
            else:
                both = torch.randn(2, HIDDEN, batch_size, generator=generator,
                                   device=where).to(dev)
            self.hidden_state, self.cell_state = both[0].t(), both[1].t()

    def forward(self, state, ref):
        x = torch.cat((state, encode_window(self, ref)), dim=1)
        carry = self.lstm(x, (self.hidden_state.contiguous(),
                              self.cell_state.contiguous()))
        self.hidden_state, self.cell_state = carry
        return self.fc_out(self.hidden_state)
