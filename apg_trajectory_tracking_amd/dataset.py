"""Device-side counterparts of the pieces of neural_control/dataset.py that
sit on the hot path: `state_preprocessing` (dataset.py:207-220, called inside
the autoregressive / LSTM loop at scripts/train_drone.py:144) and whole-tensor
training sets that replace the per-sample DataLoader collate
(scripts/train_base.py:132-137) - see `TensorBatches`."""
import torch

from . import functional as F


def state_preprocessing(drone_states):
    """state [B,12] -> policy features [B,15] =
    [v_world(3), world_to_body[:, :, :2] flattened (6), v_body(3), omega(3)];
    one HIP kernel forward (apg_quad_features_fwd), one backward."""
    return F.quad_features(drone_states)


class TensorBatches:
    """Iterates shuffled minibatches of a tuple of device tensors that share
    their first dimension.  Equivalent of DataLoader(dataset, batch_size,
    shuffle=True, num_workers=0) over a Dataset whose __getitem__ returns
    `tuple(t[i] for t in tensors)` (scripts/train_base.py:132-137,
    neural_control/dataset.py:125-132) - but one index_select per tensor per
    batch instead of O(B) Python collate calls."""

    def __init__(self, tensors, batch_size, shuffle=True, generator=None):
        n = tensors[0].shape[0]
        if any(t.shape[0] != n for t in tensors):
            raise ValueError("all tensors must share the first dimension")
        self.tensors = tuple(tensors)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator

    def __len__(self):
        n = self.tensors[0].shape[0]
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.tensors[0].shape[0]
        dev = self.tensors[0].device
        if self.shuffle:
            perm = torch.randperm(n, generator=self.generator).to(dev)
        for lo in range(0, n, self.batch_size):
            if self.shuffle:
                idx = perm[lo:lo + self.batch_size]
                yield tuple(t.index_select(0, idx) for t in self.tensors)
            else:
                yield tuple(t[lo:lo + self.batch_size] for t in self.tensors)
