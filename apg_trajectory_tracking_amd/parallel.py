"""Multi-GPU data parallelism for the APG trainers: one process per GPU,
trajectories sharded by contiguous batch ranges, policy replicated, ONE
all-reduce(sum) of the flattened policy gradient (+1 slot for the loss) per
optimizer step - `backend="nccl"` is RCCL over xGMI on MI355X.

The reference is single-process; because its losses are SUMS over the batch
(neural_control/drone_loss.py:22-34) a plain sum-reduce reproduces the
single-device gradient of the concatenated batch exactly (up to fp32
summation order), with no rescaling.  The message is tiny (32 729 floats =
131 KB for the quadrotor policy): latency-bound, so everything goes in one
bucket and one collective.
"""
import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def group_live():
    """A default process group exists (its watchdog thread runs; collectives
    are real calls into RCCL / gloo - also with a world of one)."""
    return dist.is_available() and dist.is_initialized()


_suspended = False


class collectives_suspended:
    """Context: the steps' gradient all-reduce (`reduce_sum`) is skipped while
    everything else about an N-rank step stays as it is (split graphs, the
    flat message, the optimizer behind the slot).  Measurement only - `bench.py`
    times the N-rank step with and without its collective on the same ranks
    (`parallel_efficiency`); the ranks' weights drift apart inside it, so the
    caller re-broadcasts (or discards the trainer) afterwards."""

    def __enter__(self):
        global _suspended
        self._was, _suspended = _suspended, True
        return self

    def __exit__(self, *exc):
        global _suspended
        _suspended = self._was
        return False


def reduce_sum(msg, group=None):
    """THE collective of a training step: in-place all-reduce(sum) of the flat
    gradient (+ loss slot) message over RCCL (`nccl`) or gloo.  Issued whenever
    a process group is live - also for a world of one."""
    if msg is not None and group_live() and not _suspended:
        dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)


def any_rank(flag):
    """`flag` OR-ed over the ranks (one process: `flag`).  A host decision all
    ranks must take alike - re-capturing a step graph, whose warm-up steps issue
    collectives - costs one small all-reduce and a read-back where it is asked."""
    if world_size() <= 1:
        return bool(flag)
    import torch
    dev = ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item())


def shard_range(n, r=None, world=None):
    """Contiguous [lo, hi) slice of n items owned by rank r (the first
    n % world ranks get one extra)."""
    r = rank() if r is None else r
    world = world_size() if world is None else world
    base, extra = divmod(n, world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def is_main():
    """True on the rank that writes checkpoints / logs (rank 0)."""
    return rank() == 0


@torch.no_grad()
def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers (one
    flat fp32 message).  No-op for a single process."""
    if world_size() == 1 or module is None:
        return
    tensors = [p.data for p in module.parameters()] + list(module.buffers())
    tensors = [t for t in tensors if t.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


class GradAllReducer:
    """Flat bucket over the parameters that receive gradients.

    sync(loss) packs every `.grad` and the scalar loss into one contiguous
    fp32 buffer, all-reduces it (sum) and unpacks in place; returns the
    summed loss.  With world_size == 1 it is a no-op."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self._bucket = None

    def _ensure_bucket(self, like):
        n = sum(p.numel() for p in self.params) + 1
        if (self._bucket is None or self._bucket.numel() != n
                or self._bucket.device != like.device):
            self._bucket = torch.zeros(n, dtype=torch.float32, device=like.device)
        return self._bucket

    @torch.no_grad()
    def pack(self, loss):
        """Gradients + loss into the bucket (the all-reduce message)."""
        b = self._ensure_bucket(loss)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                b[off:off + n].zero_()
            else:
                b[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        b[off] = loss.detach().reshape(())
        return b

    @torch.no_grad()
    def unpack(self):
        """The (reduced) bucket back into the gradients; returns the loss."""
        b, off = self._bucket, 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(b[off:off + n].view_as(p.grad))
            off += n
        return b[off].clone()

    @torch.no_grad()
    def sync(self, loss):
        if world_size() == 1:
            return loss
        b = self.pack(loss)
        reduce_sum(b, group=self.group)
        return self.unpack()
