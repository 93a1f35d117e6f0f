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


# ---------------------------------------------------------------------------
# Whole-tensor, device-resident training sets.  They expose the attributes the
# trainers read from the reference datasets (normed_states, states,
# in_ref_states, ref_states, mean, std, num_sampled_states, resample_data) but
# are filled from the seeded synthetic generators of synthetic.py, because the
# reference's sampled data (`data/traj_data_1`, gym environments) is outside
# the hot path and not shipped upstream.
# ---------------------------------------------------------------------------
class SyntheticQuadDataset:
    """Counterpart of QuadDataset (neural_control/dataset.py:135-204):
    `normed_states` are the 15 policy features (state_preprocessing),
    `in_ref_states` = [rel. pos, vel, vel - v_drone], `ref_states` the 9-column
    reference rows the loss reads."""

    def __init__(self, num_states, horizon, dt, ref_length=None, seed=0,
                 device="cuda"):
        self.num_sampled_states = int(num_states)
        self.horizon, self.dt = horizon, dt
        self.ref_length = ref_length or horizon
        self.device = torch.device(device)
        self.seed = seed
        self._epoch = 0
        self._fill()

    def _fill(self):
        from . import synthetic
        d = synthetic.quad_polynomial_batch(
            self.num_sampled_states, self.horizon, self.dt,
            seed=self.seed + self._epoch, ref_length=self.ref_length)
        self.states = d["state0"].to(self.device)
        self.ref_states = d["ref"].to(self.device)
        self.in_ref_states = d["in_ref"].to(self.device)
        with torch.no_grad():
            self.normed_states = state_preprocessing(self.states)
        self.mean = torch.zeros(12)
        self.std = torch.ones(12)

    def resample_data(self):
        self._epoch += 1
        self._fill()

    def __len__(self):
        return self.states.shape[0]

    def __getitem__(self, index):
        return (self.normed_states[index], self.states[index],
                self.in_ref_states[index], self.ref_states[index])


class SyntheticWingDataset:
    """Counterpart of WingDataset (neural_control/dataset.py:261-350):
    normed_states = ((state - mean) / std)[:, 3:], in_ref = last linear
    reference point relative to the aircraft, ref_states = the [N,H,3] linear
    reference of _compute_target_pos; fixed mean / std of set_fixed_mean
    (:284-299)."""

    MEAN = [0.0, 0.0, 0.0, 11.525899887084961, -0.00016766408225521445,
            0.16617104411125183, 0.007394296582788229, 0.018172707409,
            0.020353179425001144, -0.0005361468647606671, 0.01662314310669899,
            0.004487641621381044]
    STD = [16.626325607299805, 0.8449159860610962, 0.8879243731498718,
           0.6243225932121277, 0.28072822093963623, 0.29176747798,
           0.04499124363064766, 0.10370047390460968, 0.049977313727,
           0.06449887901544571, 0.27508440613746643, 0.05634994804859]

    def __init__(self, num_states, horizon, dt, seed=0, device="cuda"):
        self.num_sampled_states = int(num_states)
        self.horizon, self.dt = horizon, dt
        self.device = torch.device(device)
        self.seed = seed
        self._epoch = 0
        self.mean = torch.tensor(self.MEAN)
        self.std = torch.tensor(self.STD)
        self._fill()

    def _fill(self):
        from . import synthetic
        d = synthetic.wing_batch(self.num_sampled_states, self.horizon,
                                 self.dt, seed=self.seed + self._epoch)
        states = d["state0"]
        self.normed_states = (
            ((states - self.mean) / self.std)[:, 3:]).to(self.device)
        self.states = states.to(self.device)
        self.ref_states = d["ref"].to(self.device)
        self.in_ref_states = (d["ref"][:, -1] - states[:, :3]).to(self.device)

    def resample_data(self):
        self._epoch += 1
        self._fill()

    def __len__(self):
        return self.states.shape[0]

    def __getitem__(self, index):
        return (self.normed_states[index], self.states[index],
                self.in_ref_states[index], self.ref_states[index])


class SyntheticCartpoleDataset:
    """Counterpart of CartpoleDataset (neural_control/dataset.py:223-258):
    `states` (policy input) and `labels` (simulation start state) hold the
    same values."""

    def __init__(self, num_states=1000, seed=0, device="cuda"):
        from . import synthetic
        d = synthetic.cartpole_batch(int(num_states), 1, seed=seed)
        self.labels = d["state0"].to(torch.device(device))
        self.states = self.labels.clone()
        self.num_sampled_states = int(num_states)

    def __len__(self):
        return self.states.shape[0]

    def __getitem__(self, index):
        return self.states[index], self.labels[index]
