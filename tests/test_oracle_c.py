"""Pin the C oracle (oracle/apg_oracle.c, matrix-form restatement with a
hand-written reverse sweep) against the golden vectors from the reference
and against the PyTorch-eager oracle; fp64 build cross-checks the fp32 one."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import c_oracle as co

MOD = {"translational_drag": [.1, .2, .3], "rotational_drag": [.01, .02, .03],
       "mass": 1.0}


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
@pytest.mark.parametrize("dt", [0.05, 0.1])
def test_c_step_and_vjp(tag, mp, dt):
    g = load_golden("quad_step.npz")
    key = f"{tag}_dt{int(round(dt*100)):03d}"
    nxt = co.quad_step(g["state"], g["action"], dt, mp)
    assert rel_err(nxt, g[key + "_next"]) < 2e-6
    for i, c in enumerate(g["cot"]):
        gs, ga = co.quad_step_vjp(g["state"], g["action"], dt, c, mp)
        assert rel_err(gs, g[key + "_gstate"][i]) < 1e-5
        assert rel_err(ga, g[key + "_gaction"][i]) < 1e-5


def test_c_known_answer():
    g = load_golden("quad_step.npz")
    nxt = co.quad_step(g["ka_state"], g["ka_action"], 0.05)
    assert rel_err(nxt, g["ka_next"]) < 2e-6


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
def test_c_rollout(tag, mp):
    g = load_golden("quad_rollout.npz")
    st, loss, ga, gs = co.quad_rollout_fwd_bwd(
        g["state0"], g["actions"], g["ref"], float(g["dt"]), mp)
    assert rel_err(st, g[tag + "_states"]) < 1e-5
    assert abs(loss - g[tag + "_loss"]) / g[tag + "_loss"] < 1e-5
    assert rel_err(ga, g[tag + "_gactions"]) < 1e-5
    assert rel_err(gs, g[tag + "_gstate0"]) < 1e-5
    # fp64 build of the same source agrees with fp32 to fp32 rounding
    st64, loss64, ga64, gs64 = co.quad_rollout_fwd_bwd(
        g["state0"], g["actions"], g["ref"], float(g["dt"]), mp,
        dtype=np.float64)
    assert rel_err(ga, ga64) < 2e-5 and rel_err(gs, gs64) < 2e-5


def test_c_vjp_matches_finite_differences_fp64():
    rng = np.random.default_rng(0)
    s = rng.normal(size=(4, 12)) * 0.5
    a = rng.uniform(size=(4, 4))
    c = rng.normal(size=(4, 12))
    gs, ga = co.quad_step_vjp(s, a, 0.1, c, MOD, dtype=np.float64)
    eps = 1e-6
    for j in range(12):
        d = np.zeros_like(s); d[:, j] = eps
        fd = ((co.quad_step(s + d, a, 0.1, MOD, np.float64)
               - co.quad_step(s - d, a, 0.1, MOD, np.float64)) / (2 * eps) * c).sum(1)
        np.testing.assert_allclose(gs[:, j], fd, rtol=1e-6, atol=1e-8)
    for j in range(4):
        d = np.zeros_like(a); d[:, j] = eps
        fd = ((co.quad_step(s, a + d, 0.1, MOD, np.float64)
               - co.quad_step(s, a - d, 0.1, MOD, np.float64)) / (2 * eps) * c).sum(1)
        np.testing.assert_allclose(ga[:, j], fd, rtol=1e-6, atol=1e-8)
