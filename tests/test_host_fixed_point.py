"""The numerics contract of the trajectory-major reverse kernel's accumulators
(csrc/mlp_concurrent.hip, mlp_concurrent_bwd_tm_kernel), modelled on the host: operands
scaled into [-1, 1] by the workgroup's exponents, split into two fp16 terms,
three products per term pair accumulated in fp32 per wave (32 trajectories),
rounded to 32-bit fixed point (unit 2^-22, folded into the operand scales as
2^11 x 2^11) and added over the eight waves.
What the model pins: the integer sum does not depend on the order of the waves
(bit-reproducibility), it cannot overflow for operands at their bounds, and
the result is as close to float64 as an fp32 sum.  (The kernel itself is
compared with float64 autograd on the GPU: tests/test_gpu_in_sweep.py.)"""
import numpy as np
import pytest

FIX, PRE = 22, 11


def _exp(a):
    """e with a < 2^e (frexp's exponent), 0 for 0 - the kernel's bits_exp."""
    return int(np.frexp(np.float32(a))[1]) if a > 0 else 0


def _split(v):
    h = v.astype(np.float16)
    lo = (v - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float32), lo.astype(np.float32)


def _wave_block(delta_s, x_s):
    """One wave's 32 x 32 block over its 32 trajectories, as the matrix
    instructions form it: W_l x_h + W_h x_l + W_h x_h into an fp32 accumulator,
    16 trajectories (one instruction's k-slots) at a time."""
    dh, dl = _split(delta_s)
    xh, xl = _split(x_s)
    acc = np.zeros((delta_s.shape[0], x_s.shape[0]), np.float32)
    for k0 in (0, 16):
        sl = slice(k0, k0 + 16)
        for a, b in ((dl, xh), (dh, xl), (dh, xh)):
            acc = (acc + (a[:, sl].astype(np.float64) @ b[:, sl].T.astype(np.float64))
                   ).astype(np.float32)
    return acc


def _workgroup_block(delta, x, order):
    """delta [32, 256], x [32, 256] (rows x the workgroup's trajectories) ->
    (fixed-point sums, scale exponent)."""
    e = _exp(np.abs(delta).max())
    f = max(0, _exp(np.abs(x).max()))
    total = np.zeros((32, 32), np.int64)
    for w in order:
        sl = slice(32 * w, 32 * w + 32)
        # (the unit is folded into the operand scales, 2^11 each: the block
        # element leaves the matrix pipe in accumulator units)
        acc = _wave_block(np.ldexp(delta[:, sl], PRE - e).astype(np.float32),
                          np.ldexp(x[:, sl], FIX - PRE - f).astype(np.float32))
        q = np.rint(acc.astype(np.float64)).astype(np.int64)
        assert np.abs(q).max() < 2 ** 28
        total += q
    assert np.abs(total).max() < 2 ** 31          # fits the 32-bit accumulator
    return total, e + f


@pytest.mark.parametrize("seed,dscale,xscale", [(0, 1.0, 1.0), (1, 3e-7, 1.0),
                                                (2, 4e4, 37.0), (3, 1.0, 900.0)])
def test_fixed_point_block_is_order_independent_and_fp32_accurate(seed, dscale, xscale):
    rng = np.random.default_rng(seed)
    delta = (rng.normal(size=(32, 256)) * dscale).astype(np.float32)
    delta[:, 17] *= 40.0                  # one trajectory dominates the exponent
    x = np.tanh(rng.normal(size=(32, 256))).astype(np.float32) * np.float32(xscale)
    t0, e0 = _workgroup_block(delta, x, range(8))
    t1, e1 = _workgroup_block(delta, x, [5, 2, 7, 0, 3, 6, 1, 4])
    assert e0 == e1 and np.array_equal(t0, t1)          # any order: the same bits
    got = np.ldexp(t0.astype(np.float64), e0 - FIX)
    want = delta.astype(np.float64) @ x.T.astype(np.float64)
    f32 = np.zeros((32, 32), np.float32)
    for n in range(256):                                 # a plain fp32 sum as yardstick
        f32 += np.outer(delta[:, n], x[:, n]).astype(np.float32)
    scale = np.abs(want).max()
    err_fix = np.abs(got - want).max() / scale
    err_f32 = np.abs(f32 - want).max() / scale
    assert err_fix < 4e-6, (err_fix, err_f32)     # (an outlier trajectory x 40 sets the unit)
    assert err_fix < max(8 * err_f32, 5e-7), (err_fix, err_f32)


def test_fixed_point_block_cannot_overflow_at_the_operand_bounds():
    """Every operand at its bound with equal signs: 256 products of 1 x 1 per
    element = 2^8 in units of 2^-22: 2^30, inside int32."""
    delta = np.full((32, 256), 0.999, np.float32)
    x = np.full((32, 256), -1.0, np.float32)
    total, e = _workgroup_block(delta, x, range(8))
    assert np.abs(total).max() < 2 ** 31 and total.max() < 0
    got = np.ldexp(total.astype(np.float64), e - FIX)
    assert np.allclose(got, -0.999 * 256, rtol=1e-6)


def test_two_limb_addition_recovers_what_the_coarse_unit_drops():
    """The conv / states_in blocks (exponent = a loose bound): hi = rint(v 2^f),
    lo = rint((v 2^f - hi) 2^f); hi 2^-f + lo 2^-2f is v to 2^-(2f+1)."""
    f = 19
    rng = np.random.default_rng(4)
    v = (rng.normal(size=4096) * 1e-3).astype(np.float32)     # far below the bound
    s = np.ldexp(v.astype(np.float64), f)
    hi = np.rint(s)
    lo = np.rint(np.ldexp(s - hi, f))
    one = np.ldexp(hi, -f)
    two = np.ldexp(hi, -f) + np.ldexp(lo, -2 * f)
    assert np.abs(two - v).max() <= 2.0 ** (-2 * f - 1) * 1.0001
    assert np.abs(one - v).max() > 100 * np.abs(two - v).max()
    assert np.abs(lo).max() <= 2 ** (f - 1)
