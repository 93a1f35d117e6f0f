"""Pieces shared by the policy classes.  The parameter NAMES of the policies
(`conv_ref`, `ref_in`, `fc_out`, ...) are part of the checkpoint format shared
with the reference, so the modules themselves stay attributes of the policy
classes; only the arithmetic lives here."""
import torch

CONV_CHANNELS = 20     # conv1d(ref_dim -> 20, kernel 3) over the window
CONV_KERNEL = 3
DENSE_WIDTH = 64


def window_feature_count(horizon, conv):
    """Length of the encoded reference window."""
    return CONV_CHANNELS * (horizon - (CONV_KERNEL - 1)) if conv else DENSE_WIDTH


def encode_window(policy, ref):
    """ref [B, horizon, ref_dim] -> [B, window_feature_count]: relu(conv1d)
    over the window positions, channel-major flattened, or tanh(Linear) of
    the flattened window for conv=False policies."""
    if policy.conv:
        z = policy.conv_ref(ref.transpose(1, 2))
        return torch.relu(z).reshape(z.shape[0], -1)
    return torch.tanh(policy.ref_in(ref))
