"""Seeded synthetic inputs for the APG rollout (replaces the reference's
offline `data/traj_data_1` set, which is git-ignored upstream and absent).

Shapes and column meaning follow the reference's dataset 4-tuple
(neural_control/dataset.py:155-204) and reference-row layout
[pos(3) relative, euler*sf(3), vel(3)]
(neural_control/trajectory/generate_trajectory.py:566-605); the distributions
are the ones fixed in SURVEY.md §8(d).  Everything is drawn from a CPU
`torch.Generator` so that the GPU path, the oracle and every rank of a
multi-GPU job see bit-identical tensors for a given (seed, rank).

Tensors are returned in the reference (AoS, row-major) layout; use
`to_soa_*` helpers for the device SoA layout the fused kernels read.
"""
import math
import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def quad_polynomial_batch(batch, horizon, dt, seed=0, ref_length=None):
    """Random degree-5 polynomial reference trajectories + matching drone
    start states + random (sigmoid-normal) action sequences.

    Returns dict with
      state0   [B,12]  [p(3)=0, rpy(3), v(3), omega(3)]
      ref      [B,R,9] [pos rel., 0,0,0, vel]       (R = ref_length or H)
      in_ref   [B,R,9] [pos rel., vel, vel - v_drone] (policy input,
                        neural_control/dataset.py:194-201)
      actions  [B,H,4] in (0,1)
      coeffs   [B,3,5] polynomial coefficients c_1..c_5 per axis
    """
    g = _gen(seed)
    R = horizon if ref_length is None else ref_length
    B = batch
    scale = torch.tensor(
        [1.5] + [1.0 / math.factorial(i) for i in range(2, 6)]
    )
    coeffs = torch.randn(B, 3, 5, generator=g) * scale
    t = (torch.arange(1, R + 1, dtype=torch.float32) * dt)  # t_{k+1}
    powers = torch.stack([t**i for i in range(1, 6)], 0)         # [5,R]
    dpowers = torch.stack([i * t**(i - 1) for i in range(1, 6)], 0)
    pos = torch.einsum("bai,ir->bra", coeffs, powers)            # [B,R,3]
    vel = torch.einsum("bai,ir->bra", coeffs, dpowers)
    ref = torch.zeros(B, R, 9)
    ref[:, :, 0:3] = pos
    ref[:, :, 6:9] = vel
    state0 = torch.zeros(B, 12)
    state0[:, 3:6] = torch.rand(B, 3, generator=g) * 0.4 - 0.2
    state0[:, 6:9] = coeffs[:, :, 0] + 0.3 * torch.randn(B, 3, generator=g)
    state0[:, 9:12] = 0.1 * torch.randn(B, 3, generator=g)
    actions = torch.sigmoid(torch.randn(B, horizon, 4, generator=g))
    in_ref = torch.cat(
        (pos, vel, vel - state0[:, None, 6:9]), dim=2
    )
    return dict(
        state0=state0, ref=ref, in_ref=in_ref, actions=actions, coeffs=coeffs
    )


def wing_batch(batch, horizon, dt, seed=0):
    """Fixed-wing start states, targets and the linear reference of
    WingDataset._compute_target_pos (neural_control/dataset.py:309-320).

    Returns dict with state0 [B,12] ([pos NED, uvw, euler, pqr]),
    target [B,3], ref [B,H,3] (linear reference), actions [B,H,4].
    """
    g = _gen(seed)
    B = batch
    state0 = torch.zeros(B, 12)
    state0[:, 3] = 11.5 + 0.5 * torch.randn(B, generator=g)
    state0[:, 4:6] = 0.3 * torch.randn(B, 2, generator=g)
    state0[:, 6:9] = 0.05 * torch.randn(B, 3, generator=g)
    state0[:, 9:12] = 0.05 * torch.randn(B, 3, generator=g)
    target = torch.empty(B, 3)
    target[:, 0] = 50.0
    target[:, 1:] = torch.rand(B, 2, generator=g) * 10 - 5
    rel = target - state0[:, :3]
    nvec = rel / torch.sqrt(torch.sum(rel**2, dim=1, keepdim=True))
    steps = torch.arange(1, horizon + 1, dtype=torch.float32)
    ref = state0[:, None, :3] + nvec[:, None, :] * (12 * dt) * steps[None, :, None]
    actions = torch.sigmoid(torch.randn(B, horizon, 4, generator=g))
    return dict(state0=state0, target=target, ref=ref, actions=actions)


def cartpole_batch(batch, horizon, seed=0):
    """state0 ~ U(-1,1)^4 * [2.4, 1.5, pi, 1.5]; actions ~ tanh(N(0,1))
    (the cartpole policy ends in tanh, scripts/train_cartpole.py:127-130)."""
    g = _gen(seed)
    span = torch.tensor([2.4, 1.5, math.pi, 1.5])
    state0 = (torch.rand(batch, 4, generator=g) * 2 - 1) * span
    actions = torch.tanh(torch.randn(batch, horizon, 1, generator=g))
    return dict(state0=state0, actions=actions)


# ---- layout helpers: reference AoS <-> device SoA (batch fastest) ---------
def to_soa_state(x):
    """[B,S] -> [S,B] contiguous."""
    return x.t().contiguous()


def to_soa_seq(x):
    """[B,H,C] -> [H,C,B] contiguous."""
    return x.permute(1, 2, 0).contiguous()


def from_soa_state(x):
    return x.t().contiguous()


def from_soa_seq(x):
    """[H,C,B] -> [B,H,C] contiguous."""
    return x.permute(2, 0, 1).contiguous()


# APG_LAYOUT_PACKED (include/apg.h): rows of one trajectory's floats, batch as
# the next-faster dimension - state [S/4][B][4], sequences [H][B][C]
def to_packed_state(x):
    """[B, S] -> [S/4, B, 4]"""
    B, S = x.shape
    return x.reshape(B, S // 4, 4).permute(1, 0, 2).contiguous()


def to_packed_seq(x):
    """[B, H, C] -> [H, B, C]"""
    return x.permute(1, 0, 2).contiguous()


def from_packed_state(x):
    """[S/4, B, 4] -> [B, S]"""
    G, B, W = x.shape
    return x.permute(1, 0, 2).reshape(B, G * W).contiguous()


def from_packed_seq(x):
    """[H, B, C] -> [B, H, C]; states [H, 3, B, 4] -> [B, H, 12]"""
    if x.dim() == 4:
        H, G, B, W = x.shape
        return x.permute(2, 0, 1, 3).reshape(B, H, G * W).contiguous()
    return x.permute(1, 0, 2).contiguous()


def quad_eval_trajectories(batch, length, dt, seed=42, speed=1.0):
    """Long smooth reference trajectories for the closed-loop evaluation, in the
    row format of `load_prepare_trajectory` (neural_control/trajectory/
    generate_trajectory.py:566-605): [position(3), euler(3), velocity(3)] per
    time step, [B, L, 9].  The reference reads min-snap trajectories from
    data/traj_data_1 (not shipped); here each axis is a sum of three sinusoids
    with random amplitude / frequency / phase (bounded, smooth, velocity = the
    exact derivative), starting at the origin; euler columns are zero."""
    g = _gen(seed)
    B, L = batch, length
    t = torch.arange(L, dtype=torch.float32) * dt
    amp = torch.rand(B, 3, 3, generator=g) * torch.tensor([1.5, 0.6, 0.2]) + 0.05
    freq = (torch.rand(B, 3, 3, generator=g) * torch.tensor([0.5, 0.8, 1.0])
            + torch.tensor([0.2, 0.7, 1.5])) * speed
    phase = torch.rand(B, 3, 3, generator=g) * (2 * math.pi)
    arg = freq[..., None] * t + phase[..., None]                  # [B,3,3,L]
    pos = (amp[..., None] * torch.sin(arg)).sum(2)                # [B,3,L]
    vel = (amp[..., None] * freq[..., None] * torch.cos(arg)).sum(2)
    pos = pos - pos[:, :, :1]
    traj = torch.zeros(B, L, 9)
    traj[:, :, 0:3] = pos.transpose(1, 2)
    traj[:, :, 6:9] = vel.transpose(1, 2)
    return traj
