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
    batch instead of O(B) Python collate calls.

    Data parallel (`shard=(rank, world)`, world > 1): `batch_size` stays the
    GLOBAL minibatch; every rank holds the same data set, draws the SAME
    permutation (a generator seeded with `shard_seed`, advanced once per
    epoch on every rank) and takes its contiguous slice of each global batch
    (parallel.shard_range) - so all ranks see the same number of batches and,
    PROVIDED every rank holds the same data set (same seed / a broadcast: the
    loader does not check), the union of their slices is the single-process
    batch.  A ragged last batch with fewer rows than ranks would leave some
    ranks an EMPTY slice (kernels and the all-reduce expect >= 1 row each): it
    is dropped on every rank alike (at most world - 1 samples per epoch; the
    single-process loader keeps it, like the reference's drop_last=False)."""

    def __init__(self, tensors, batch_size, shuffle=True, generator=None,
                 shard=None, shard_seed=0):
        n = tensors[0].shape[0]
        if any(t.shape[0] != n for t in tensors):
            raise ValueError("all tensors must share the first dimension")
        self.tensors = tuple(tensors)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator
        self.shard = shard if shard and shard[1] > 1 else None
        if self.shard and self.generator is None:
            self.generator = torch.Generator().manual_seed(int(shard_seed))

    def _slice(self, idx):
        """This rank's part of one global batch of indices."""
        if self.shard is None:
            return idx
        from .parallel import shard_range
        lo, hi = shard_range(idx.shape[0], *self.shard)
        return idx[lo:hi]

    def _too_small(self, rows):
        """A global batch with fewer rows than ranks (see the class docstring)."""
        return self.shard is not None and rows < self.shard[1]

    def __len__(self):
        n = self.tensors[0].shape[0]
        full, tail = divmod(n, self.batch_size)
        return full + (1 if tail and not self._too_small(tail) else 0)

    def epoch_order(self):
        """This epoch's sample order (one draw of the permutation, exactly what
        iter_indices() would draw) - the one prefetch_order() drew ahead, if it
        still fits."""
        n = self.tensors[0].shape[0]
        dev = self.tensors[0].device
        ahead = self.__dict__.pop("_order_ahead", None)
        if (ahead is not None and self.shuffle and ahead[0].numel() == n
                and ahead[0].device == dev):
            main = torch.cuda.current_stream(dev)
            main.wait_event(ahead[1])
            ahead[0].record_stream(main)
            return ahead[0]
        return self._permutation(n, dev) if self.shuffle else torch.arange(n, device=dev)

    def prefetch_order(self, stream):
        """Draw the NEXT epoch's permutation now, on `stream` (a side stream): a
        device permutation of 2 M rows is ~0.25 ms of sort kernels, which
        otherwise stand between an epoch's loss read-back and the next epoch's
        first step.  The draws come from the same generator in the same order."""
        dev = self.tensors[0].device
        if not (self.shuffle and self.generator is None and dev.type == "cuda"):
            return
        with torch.cuda.stream(stream):
            order = self._permutation(self.tensors[0].shape[0], dev)
            self._order_ahead = (order, stream.record_event())

    def iter_indices(self, order=None):
        """The same batches as __iter__, as index tensors (int64, on the data's
        device) into `self.tensors`: lets a consumer fold the row gather into
        its own first pass over the data (functional.to_soa(index=...)).
        `order`: an epoch_order() drawn (or kept in a persistent buffer) by the
        caller; the batches are then views of it."""
        n = self.tensors[0].shape[0]
        if order is None:
            order = self.epoch_order()
        for lo in range(0, n, self.batch_size):
            if self._too_small(min(self.batch_size, n - lo)):
                continue
            yield self._slice(order[lo:lo + self.batch_size])

    def _permutation(self, n, dev):
        # drawn on the device unless a (CPU) generator pins the order: a host
        # permutation of 5e5 indices + its upload costs more than a whole
        # fused training step
        if self.generator is None:
            return torch.randperm(n, device=dev)
        return torch.randperm(n, generator=self.generator).to(dev)

    def __iter__(self):
        n = self.tensors[0].shape[0]
        dev = self.tensors[0].device
        if self.shuffle:
            perm = self._permutation(n, dev)
        for lo in range(0, n, self.batch_size):
            if self._too_small(min(self.batch_size, n - lo)):
                continue
            if self.shuffle:
                idx = self._slice(perm[lo:lo + self.batch_size])
                yield tuple(t.index_select(0, idx) for t in self.tensors)
            elif self.shard is not None:
                idx = self._slice(torch.arange(
                    lo, min(lo + self.batch_size, n), device=dev))
                yield tuple(t[idx[0]:idx[-1] + 1] if idx.numel() else t[:0]
                            for t in self.tensors)
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
    """Counterpart of QuadDataset / DroneDataset (neural_control/dataset.py:
    46-204): `normed_states` are the 15 policy features (state_preprocessing),
    `in_ref_states` = [rel. pos, vel, vel - v_drone], `ref_states` the 9-column
    reference rows the loss reads.  As in the reference the set has
    `num_sampled_states` sampled entries, refreshed by `resample_data`, followed
    by `num_self_play = int(self_play * num_states)` slots that the closed-loop
    evaluation overwrites cyclically with the states it visited
    (`add_eval_data`, the batched form of get_and_add_eval_data :103-119)."""

    _version = 0     # bumped by every in-place change of the tensors
    _packed = None   # (version, state0 rows, reference rows), see packed()

    def __init__(self, num_states, horizon, dt, ref_length=None, seed=0,
                 device="cuda", self_play=0.0):
        self.num_sampled_states = int(num_states)
        self.num_self_play = int(self_play * num_states)
        self.total_dataset_size = self.num_sampled_states + self.num_self_play
        self.horizon, self.dt = horizon, dt
        self.ref_length = ref_length or horizon
        self.device = torch.device(device)
        self.seed = seed
        self._epoch = 0
        self.eval_counter = 0
        self.mean = torch.zeros(12)
        self.std = torch.ones(12)
        (self.normed_states, self.states, self.in_ref_states,
         self.ref_states) = self._sample(self.total_dataset_size)

    def _sample(self, n):
        from . import synthetic
        d = synthetic.quad_polynomial_batch(
            n, self.horizon, self.dt, seed=self.seed + self._epoch,
            ref_length=self.ref_length)
        states = d["state0"].to(self.device)
        with torch.no_grad():
            normed = state_preprocessing(states)
        return (normed, states, d["in_ref"].to(self.device),
                d["ref"].to(self.device))

    def resample_data(self):
        """:87-101 - only the sampled part is renewed."""
        self._epoch += 1
        n = self.num_sampled_states
        for dst, src in zip((self.normed_states, self.states,
                             self.in_ref_states, self.ref_states), self._sample(n)):
            dst[:n] = src
        self._version += 1

    def prepare_data(self, states, ref_states):
        """QuadDataset.prepare_data (:155-204) on device tensors: states [n,12],
        ref_states [n,R,9] (position, euler, velocity rows) -> (policy
        features, states with the position zeroed, policy reference input,
        reference rows relative to the drone position)."""
        states = states.to(self.device, torch.float32).clone()
        ref = ref_states.to(self.device, torch.float32).clone()
        pos, vel = states[:, None, :3].clone(), states[:, None, 6:9].clone()
        ref[:, :, :3] -= pos
        states[:, :3] = 0
        with torch.no_grad():
            normed = state_preprocessing(states)
        in_ref = torch.cat((ref[:, :, :3], ref[:, :, 6:9], ref[:, :, 6:9] - vel), 2)
        return normed, states, in_ref, ref

    def get_eval_index(self):
        if self.num_self_play > 0:
            return self.eval_counter % self.num_self_play + self.num_sampled_states

    def add_eval_data(self, states, ref_states):
        """Overwrite the next self-play slots with n visited (state, reference
        window) pairs, in order, wrapping around like repeated
        get_and_add_eval_data(..., add_to_dataset=True) calls."""
        n = states.shape[0]
        if self.num_self_play == 0 or n == 0:
            return 0
        if ref_states.shape[1] != self.ref_length:
            raise ValueError("reference window length != data set ref_length")
        prepared = self.prepare_data(states, ref_states)
        # only the last num_self_play entries survive a wrap-around
        keep = min(n, self.num_self_play)
        idx = ((self.eval_counter + torch.arange(n - keep, n, device=self.device))
               % self.num_self_play + self.num_sampled_states)
        for dst, src in zip((self.normed_states, self.states,
                             self.in_ref_states, self.ref_states), prepared):
            dst[idx] = src[n - keep:]
        self.eval_counter += n
        self._version += 1
        return n

    def packed(self):
        """The simulation inputs of the whole set as ROWS (`APG_LAYOUT_PACKED`,
        include/apg.h): state0 [3, N, 4] and the reference's [pos, vel]
        columns [R, N, 6] - what the fused quadrotor rollout reads with one
        16-byte access per lane.  Built once per version of the data
        (resample_data / add_eval_data invalidate it); a minibatch is
        `index_select(1, index)` of these, the identity batch is the tensors
        themselves."""
        if self._packed is None or self._packed[0] != self._version:
            from . import synthetic
            r = self.ref_states
            ref6 = torch.cat((r[:, :, :3], r[:, :, 6:9]), 2)
            self._packed = (self._version,
                            synthetic.to_packed_state(self.states),
                            synthetic.to_packed_seq(ref6))
        return self._packed[1], self._packed[2]

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
    (:284-299) unless given.  As in the reference the set has
    `num_sampled_states` sampled entries, refreshed by `resample_data`,
    followed by `num_self_play = int(self_play * num_states)` slots that the
    closed-loop evaluation overwrites cyclically with the states it visited
    (`add_eval_data`, the batched get_and_add_eval_data :103-119)."""

    MEAN = [0.0, 0.0, 0.0, 11.525899887084961, -0.00016766408225521445,
            0.16617104411125183, 0.007394296582788229, 0.018172707409,
            0.020353179425001144, -0.0005361468647606671, 0.01662314310669899,
            0.004487641621381044]
    STD = [16.626325607299805, 0.8449159860610962, 0.8879243731498718,
           0.6243225932121277, 0.28072822093963623, 0.29176747798,
           0.04499124363064766, 0.10370047390460968, 0.049977313727,
           0.06449887901544571, 0.27508440613746643, 0.05634994804859]

    def __init__(self, num_states, horizon, dt, seed=0, device="cuda",
                 self_play=0.0, mean=None, std=None):
        self.num_sampled_states = int(num_states)
        self.num_self_play = int(self_play * num_states)
        self.total_dataset_size = self.num_sampled_states + self.num_self_play
        self.horizon, self.dt = horizon, dt
        self.device = torch.device(device)
        self.seed = seed
        self._epoch = 0
        self.eval_counter = 0
        self.mean = torch.tensor(self.MEAN if mean is None else mean).float()
        self.std = torch.tensor(self.STD if std is None else std).float()
        self._fill(self.total_dataset_size)

    def _fill(self, n):
        from . import synthetic
        d = synthetic.wing_batch(n, self.horizon, self.dt,
                                 seed=self.seed + self._epoch)
        states = d["state0"]
        fresh = dict(
            normed_states=((states - self.mean) / self.std)[:, 3:],
            states=states, ref_states=d["ref"],
            in_ref_states=d["ref"][:, -1] - states[:, :3])
        for name, t in fresh.items():
            if hasattr(self, name):    # renew IN PLACE: the trainer's loader
                getattr(self, name)[:n].copy_(t)   # holds these very tensors
            else:
                setattr(self, name, t.to(self.device))

    def resample_data(self):
        """:87-101 - only the sampled part is renewed."""
        self._epoch += 1
        self._fill(self.num_sampled_states)

    def _compute_target_pos(self, current_state, ref_vector):
        """:309-320: points 12 * dt * (i + 1) along ref_vector, the products in
        the reference's order (float32 tensor x Python double x int)."""
        steps = torch.arange(1, self.horizon + 1, device=current_state.device,
                             dtype=torch.float32)
        step_vec = ref_vector * (12 * self.dt)
        return current_state[:, None, :3] + step_vec[:, None, :] * steps[None, :, None]

    def prepare_data(self, states, ref_states):
        """WingDataset.prepare_data (:322-350) on device tensors: states [n,12],
        ref_states [n,3] (target points) -> (normed_states, states, in_ref,
        linear reference [n,H,3])."""
        states = states.to(self.device, torch.float32)
        target = ref_states.to(self.device, torch.float32)
        mean, std = self.mean.to(self.device), self.std.to(self.device)
        normed = ((states - mean) / std)[:, 3:]
        rel = target - states[:, :3]
        norm = torch.sqrt(torch.sum(rel**2, dim=1))
        nvec = (rel.t() / norm).t()
        ref = self._compute_target_pos(states, nvec)
        return normed, states, ref[:, -1] - states[:, :3], ref

    def get_eval_index(self):
        if self.num_self_play > 0:
            return self.eval_counter % self.num_self_play + self.num_sampled_states

    def add_eval_data(self, states, targets):
        """Overwrite the next self-play slots with n visited (state, target)
        pairs, in order, wrapping around like repeated
        get_and_add_eval_data(..., add_to_dataset=True) calls."""
        n = states.shape[0]
        if self.num_self_play == 0 or n == 0:
            return 0
        prepared = self.prepare_data(states, targets)
        keep = min(n, self.num_self_play)   # only these survive a wrap-around
        idx = ((self.eval_counter + torch.arange(n - keep, n, device=self.device))
               % self.num_self_play + self.num_sampled_states)
        for dst, src in zip((self.normed_states, self.states,
                             self.in_ref_states, self.ref_states), prepared):
            dst[idx] = src[n - keep:]
        self.eval_counter += n
        return n

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
