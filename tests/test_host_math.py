"""The SHIPPED per-trajectory arithmetic (csrc/quad_math.h, csrc/wing_math.h,
sincos_fast of csrc/apg_device.h - what the quadrotor / fixed-wing kernels
execute per lane) compiled for the HOST and pinned to the golden vectors
recorded from the reference: single steps + VJPs (G1, G5), the rollout
compositions with gradients (G2, G5), state_preprocessing + VJP (G7), the
reference's 1 001-step wing trace, and the accuracy claim of the branch-free
sin-cos.  No GPU needed: the same source with the CPU's fma / rint / sqrt."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden, rel_err

HERE = os.path.join(REPO, "tests", "host_math")
OUT = os.path.join(HERE, "_build")
CSRC = os.path.join(REPO, "apg_trajectory_tracking_amd", "csrc")
MOD = {"translational_drag": [.1, .2, .3], "rotational_drag": [.01, .02, .03],
       "mass": 1.0}


def _host_lib(name, headers):
    """hipcc --cuda-host-only build of tests/host_math/<name>.hip."""
    src = os.path.join(HERE, name + ".hip")
    lib = os.path.join(OUT, f"lib{name}.so")
    deps = [src] + [os.path.join(CSRC, h) for h in headers]
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(lib):
        pytest.skip("no hipcc to compile the host harness")
    os.makedirs(OUT, exist_ok=True)
    if (not os.path.exists(lib)
            or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps)):
        subprocess.run(
            ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared",
             "--cuda-host-only", "-I", os.path.join(REPO, "include"), "-I", CSRC,
             "-o", lib, src], check=True, stdout=subprocess.DEVNULL,
            stderr=subprocess.DEVNULL)
    return ctypes.CDLL(lib)


@pytest.fixture(scope="module")
def hm():
    lib = _host_lib("quad_math_host", ["quad_math.h", "apg_device.h"])
    lib.hm_quad_rollout.restype = ctypes.c_double
    return lib


@pytest.fixture(scope="module")
def hw():
    lib = _host_lib("wing_math_host", ["wing_math.h", "apg_device.h"])
    lib.hm_wing_rollout.restype = ctypes.c_double
    return lib


def _params(mp):
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    return FlightmareDynamics(modified_params=mp).params


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def test_sincos_fast_accuracy(hm):
    """Max error 1.6 ulp for |x| <= 1e5 (apg_device.h)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-4, 4, 400000), rng.uniform(-1e5, 1e5, 400000),
                        rng.normal(size=200000) * 1e-3]).astype(np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    hm.hm_sincos(_p(x), x.size, _p(s), _p(c))
    xs = x.astype(np.float64)
    for got, want in ((s, np.sin(xs)), (c, np.cos(xs))):
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        assert (np.abs(got - want) / ulp).max() < 1.7
    bad = np.array([np.nan, np.inf, -np.inf], np.float32)
    s3, c3 = np.empty_like(bad), np.empty_like(bad)
    hm.hm_sincos(_p(bad), 3, _p(s3), _p(c3))
    assert np.isnan(s3).all() and np.isnan(c3).all()


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
@pytest.mark.parametrize("dt", [0.05, 0.1])
def test_shipped_step_and_vjp(hm, tag, mp, dt):
    g = load_golden("quad_step.npz")
    key = f"{tag}_dt{int(round(dt*100)):03d}"
    s, a = _f(g["state"]), _f(g["action"])
    par = _params(mp)
    for i, c in enumerate(g["cot"]):
        nxt, gs, ga = np.empty_like(s), np.empty_like(s), np.empty_like(a)
        hm.hm_quad_step(_p(s), _p(a), ctypes.c_float(dt), ctypes.byref(par),
                        s.shape[0], _p(_f(c)), _p(nxt), _p(gs), _p(ga))
        assert rel_err(nxt, g[key + "_next"]) < 2e-6
        assert rel_err(gs, g[key + "_gstate"][i]) < 1e-5
        assert rel_err(ga, g[key + "_gaction"][i]) < 1e-5
    ka_s, ka_a = _f(g["ka_state"]), _f(g["ka_action"])   # the reference's own vector
    nxt = np.empty_like(ka_s)
    hm.hm_quad_step(_p(ka_s), _p(ka_a), ctypes.c_float(0.05),
                    ctypes.byref(_params({})), ka_s.shape[0], None, _p(nxt), None, None)
    assert rel_err(nxt, g["ka_next"]) < 2e-6


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
def test_shipped_rollout_composition(hm, tag, mp):
    from apg_trajectory_tracking_amd import functional as F
    g = load_golden("quad_rollout.npz")
    s0, act, ref = _f(g["state0"]), _f(g["actions"]), _f(g["ref"])
    B, H = act.shape[:2]
    st = np.empty((B, H, 12), np.float32)
    ga, gs = np.empty_like(act), np.empty_like(s0)
    w = F.quad_loss_weights()
    loss = hm.hm_quad_rollout(
        _p(s0), _p(act), _p(ref), ref.shape[2], ctypes.c_float(float(g["dt"])),
        ctypes.byref(_params(mp)), ctypes.byref(w), B, H, _p(st), _p(ga), _p(gs))
    assert rel_err(st, g[tag + "_states"]) < 1e-5
    assert abs(loss - g[tag + "_loss"]) / g[tag + "_loss"] < 1e-5
    assert rel_err(ga, g[tag + "_gactions"]) < 1e-5
    assert rel_err(gs, g[tag + "_gstate0"]) < 1e-5


def test_shipped_features_and_vjp(hm):
    g = load_golden("features.npz")
    s, cot = _f(g["state"]), _f(g["cot"])
    feat = np.empty((s.shape[0], 15), np.float32)
    gs = np.empty_like(s)
    hm.hm_quad_features(_p(s), s.shape[0], _p(cot), _p(feat), _p(gs))
    assert rel_err(feat, g["feat"]) < 2e-6
    assert rel_err(gs, g["gstate"]) < 1e-5


# ------------------------------------------------------------- fixed wing
def _wing_params(mp):
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    return FixedWingDynamics(modified_params=mp).params


def _wing_step(hw, s, a, dt, mp, steps=1, cot=None):
    s, a = _f(s), _f(a)
    nxt = np.empty_like(s)
    gs, ga = (np.empty_like(s), np.empty_like(a)) if cot is not None else (None, None)
    hw.hm_wing_step(_p(s), _p(a), ctypes.c_float(dt), ctypes.byref(_wing_params(mp)),
                    s.shape[0], int(steps), _p(None if cot is None else _f(cot)), _p(nxt),
                    _p(gs), _p(ga))
    return nxt, gs, ga


def test_shipped_wing_known_answers(hw):
    g = load_golden("wing.npz")
    nxt, _, _ = _wing_step(hw, g["ka_state"], g["ka_action"], 0.05, {})
    assert rel_err(nxt, g["ka_next"]) < 2e-6
    # tests/run_wing_sim.py: 1001-step open-loop trace, rows recorded on the way
    state = np.zeros((1, 12), np.float32)
    state[0, 3] = 11.5
    rows = list(g["sim_rows"])
    got, done = [], 0
    for r in rows:
        state, _, _ = _wing_step(hw, state, g["sim_action"], 1 / 100, {}, steps=r - done)
        done = r
        got.append(state[0].copy())
    assert rel_err(np.stack(got), g["sim_states"]) < 2e-4   # 1000 chained fp32 steps


@pytest.mark.parametrize("tag,mp", [
    ("def", {}), ("mod", {"mass": 1.4, "I_xz": -0.01, "CL0": 0.3, "rho": 1.0})
])
def test_shipped_wing_step_and_vjp(hw, tag, mp):
    g = load_golden("wing.npz")
    for i, c in enumerate(g["step_cot"]):
        nxt, gs, ga = _wing_step(hw, g["step_state"], g["step_action"], 0.05, mp, cot=c)
        assert rel_err(nxt, g[f"step_{tag}_next"]) < 2e-6
        assert rel_err(gs, g[f"step_{tag}_gstate"][i]) < 1e-5
        assert rel_err(ga, g[f"step_{tag}_gaction"][i]) < 1e-5


@pytest.mark.parametrize("H", [20, 10])
def test_shipped_wing_rollout_composition(hw, H):
    from apg_trajectory_tracking_amd import functional as F
    g = load_golden("wing.npz")
    p = f"h{H}_"
    s0, act, ref = _f(g[p + "state0"]), _f(g[p + "actions"]), _f(g[p + "ref"])
    B = s0.shape[0]
    st = np.empty((B, H, 12), np.float32)
    ga, gs = np.empty_like(act), np.empty_like(s0)
    w = F.wing_loss_weights()
    loss = hw.hm_wing_rollout(_p(s0), _p(act), _p(ref), ctypes.c_float(0.05),
                              ctypes.byref(_wing_params({})), ctypes.byref(w), B, H,
                              _p(st), _p(ga), _p(gs))
    assert rel_err(st, g[p + "states"]) < 1e-5
    assert abs(loss - g[p + "loss"]) / g[p + "loss"] < 1e-5
    assert rel_err(ga, g[p + "gactions"]) < 2e-5
    assert rel_err(gs, g[p + "gstate0"]) < 2e-5


# --------------------------------------------------------------- cart-pole
@pytest.fixture(scope="module")
def hc():
    lib = _host_lib("cartpole_math_host", ["cartpole_math.h", "apg_device.h"])
    lib.hm_cart_rollout.restype = ctypes.c_double
    return lib


def test_shipped_cartpole(hc):
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import CartpoleDynamics
    g = load_golden("cartpole.npz")
    par = CartpoleDynamics().params
    s, a = _f(g["ka_state"]), _f(g["ka_action"])
    nxt = np.empty_like(s)
    hc.hm_cart_step(_p(s), _p(a), ctypes.c_float(0.02), ctypes.byref(par), 1, None,
                    _p(nxt), None, None)
    assert rel_err(nxt, g["ka_next"]) < 2e-6
    np.testing.assert_allclose(nxt[0], [0.5260, 1.4057, 0.1080, 0.7744], atol=5e-5)
    # step + VJP
    s, a, cot = _f(g["state0"]), _f(g["actions"][:, 0]), _f(g["step_cot"])
    nxt, gs, ga = np.empty_like(s), np.empty_like(s), np.empty_like(a)
    hc.hm_cart_step(_p(s), _p(a), ctypes.c_float(0.02), ctypes.byref(par), s.shape[0],
                    _p(cot), _p(nxt), _p(gs), _p(ga))
    assert rel_err(nxt, g["step_next"]) < 2e-6
    assert rel_err(gs, g["step_gstate"]) < 1e-5
    assert rel_err(ga, g["step_gaction"]) < 1e-5
    # rollout with the reference derived from the start state
    act = _f(g["actions"])
    B, H = act.shape[:2]
    st = np.empty((B, H, 4), np.float32)
    gact, gs0 = np.empty_like(act), np.empty_like(s)
    loss = hc.hm_cart_rollout(_p(s), _p(act), ctypes.c_float(float(g["dt"])),
                              ctypes.byref(par), B, H, _p(st), _p(gact), _p(gs0))
    assert rel_err(st, g["states"]) < 1e-5
    assert abs(loss - g["loss"]) / g["loss"] < 1e-5
    assert rel_err(gact, g["gactions"]) < 1e-5
    assert rel_err(gs0, g["gstate0"]) < 1e-5


# ------------------------------------ float64 arbitration (VERDICT r3 #4a)
def test_shipped_quad_math_is_no_worse_than_fp32_reference_per_trajectory(hm):
    """The arithmetic the rollout kernels run per lane (closed form, contracted
    adjoint, branch-free sin-cos) against the float64 oracle, per trajectory,
    with the float32 oracle (the reference's op sequence) as the yardstick:
    20 000 synthetic trajectories, no escape hatch beyond 1e-4."""
    import torch
    from conftest import assert_no_worse_than_fp32
    from apg_trajectory_tracking_amd import functional as F, synthetic as sy
    from oracle import torch_port as tp
    B, H, dt = 20000, 10, 0.1
    d = sy.quad_polynomial_batch(B, H, dt, seed=3)
    st, _, ga, gs = tp.rollout_fwd_bwd(tp.QuadOracle(), tp.quad_mpc_loss,
                                       d["state0"], d["actions"], d["ref"], dt)
    d64 = {k: v.double() for k, v in d.items()}
    st64, _, ga64, gs64 = tp.rollout_fwd_bwd(
        tp.QuadOracle(dtype=torch.float64), tp.quad_mpc_loss, d64["state0"],
        d64["actions"], d64["ref"], dt)
    s0, a, r = (_f(d[k].numpy()) for k in ("state0", "actions", "ref"))
    hst = np.empty((B, H, 12), np.float32)
    hga, hgs = np.empty((B, H, 4), np.float32), np.empty((B, 12), np.float32)
    w = F.quad_loss_weights()
    hm.hm_quad_rollout(_p(s0), _p(a), _p(r), 9, ctypes.c_float(dt),
                       ctypes.byref(_params({})), ctypes.byref(w), B, H, _p(hst),
                       _p(hga), _p(hgs))
    assert_no_worse_than_fp32(hga, ga.numpy(), ga64.numpy(), "host quad dL/dactions")
    assert_no_worse_than_fp32(hgs, gs.numpy(), gs64.numpy(), "host quad dL/dstate0")
    assert_no_worse_than_fp32(hst, st.numpy(), st64.numpy(), "host quad states")


def test_shipped_wing_math_is_no_worse_than_fp32_reference_per_trajectory(hw):
    """The same for wing_math.h (polynomial atan / sin / cos, contracted
    adjoint) at H = 20: C oracle in float32 as the yardstick, in float64 as
    the arbiter."""
    from conftest import assert_no_worse_than_fp32
    from apg_trajectory_tracking_amd import functional as F, synthetic as sy
    from oracle import c_oracle as co
    B, H, dt = 20000, 20, 0.05
    d = sy.wing_batch(B, H, dt, seed=5)
    s0, a, r = (_f(d[k].numpy()) for k in ("state0", "actions", "ref"))
    c64 = co.wing_rollout_fwd_bwd(s0, a, r, dt, modified_params={}, dtype=np.float64)
    c32 = co.wing_rollout_fwd_bwd(s0, a, r, dt, modified_params={}, dtype=np.float32)
    hst = np.empty((B, H, 12), np.float32)
    hga, hgs = np.empty_like(a), np.empty_like(s0)
    w = F.wing_loss_weights()
    hw.hm_wing_rollout(_p(s0), _p(a), _p(r), ctypes.c_float(dt),
                       ctypes.byref(_wing_params({})), ctypes.byref(w), B, H,
                       _p(hst), _p(hga), _p(hgs))
    assert_no_worse_than_fp32(hst, c32[0], c64[0], "host wing states")
    assert_no_worse_than_fp32(hga, c32[2], c64[2], "host wing dL/dactions")
    assert_no_worse_than_fp32(hgs, c32[3], c64[3], "host wing dL/dstate0")
