"""Checkpoint interchange with the reference.

The reference saves WHOLE-MODULE pickles (`torch.save(self.net, path)`,
scripts/train_base.py:233-259), which can only be unpickled with the
reference's own module tree importable.  The portable form is the
`state_dict`: parameter names and shapes of the package's model classes are
identical to the reference's, so

    net = build_policy(system, state_dict)     # dims inferred from shapes
    torch.save(net.state_dict(), path)         # what this package writes

round-trips between the two code bases.  `reference_pickle_to_state_dict`
performs the one-off conversion where the reference IS importable."""
import torch

from .models.hutter_model import Net
from .models.rnn import LSTM_NEW
from .models.simple_model import Net as CartpoleNet


def reference_pickle_to_state_dict(path):
    """Unpickle a reference `model_<system>` file (needs `neural_control` on
    sys.path) and return its state_dict as CPU tensors."""
    net = torch.load(path, map_location="cpu", weights_only=False)
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}


def build_policy(system, state_dict, conv=True):
    """Instantiate the matching policy class and load `state_dict`
    (name -> tensor or numpy array).  Dimensions are inferred from the
    parameter shapes (`conv` only disambiguates LSTM_NEW, whose shapes do not
    tell the two reference branches apart)."""
    sd = {k: torch.as_tensor(v) for k, v in state_dict.items()}
    if system == "cartpole":
        net = CartpoleNet(sd["fc0.weight"].shape[1], sd["fc_out.weight"].shape[0])
    elif "lstm.weight_ih" in sd:
        ref_dim = sd["conv_ref.weight"].shape[1]
        horizon = sd["ref_in.weight"].shape[1] // ref_dim
        reshape_len = 20 * (horizon - 2) if conv else 64
        state_dim = sd["lstm.weight_ih"].shape[1] - reshape_len
        net = LSTM_NEW(state_dim, horizon, ref_dim, sd["fc_out.weight"].shape[0],
                       conv=conv)
    else:
        state_dim = sd["states_in.weight"].shape[1]
        ref_dim = sd["conv_ref.weight"].shape[1]
        horizon = sd["ref_in.weight"].shape[1] // ref_dim
        # fc1 takes 64 + 20 * (horizon - 2) inputs with the conv branch, 128 without
        conv = sd["fc1.weight"].shape[1] == 64 + 20 * (horizon - 2) and horizon > 2
        net = Net(state_dim, horizon, ref_dim, sd["fc_out.weight"].shape[0],
                  conv=conv)
    net.load_state_dict(sd)
    return net


def load_policy(path, system="quad", conv=True):
    """Policy from a checkpoint FILE holding a state_dict (what this package
    writes; a reference pickle converted by reference_pickle_to_state_dict).
    A reference whole-module pickle is refused with a pointer to the
    converter rather than unpickled blindly."""
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        raise ValueError(
            f"{path} is not a state_dict checkpoint ({type(e).__name__}); "
            "convert reference pickles with "
            "checkpoint.reference_pickle_to_state_dict where the reference "
            "is importable") from e
    if not isinstance(sd, dict):
        raise ValueError(f"{path}: expected a state_dict, got {type(sd).__name__}")
    return build_policy(system, sd, conv=conv)
