"""The SHIPPED per-trajectory quadrotor arithmetic (csrc/quad_math.h and
sincos_fast of csrc/apg_device.h - what every quad kernel executes per lane)
compiled for the HOST and pinned to the golden vectors recorded from the
reference: single step + VJPs (G1), the rollout composition with gradients
(G2), state_preprocessing + VJP (G7), and the accuracy claim of the
branch-free sin-cos.  No GPU needed: the same source, the CPU's fma / rint."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden, rel_err

SRC = os.path.join(REPO, "tests", "host_math", "quad_math_host.hip")
OUT = os.path.join(REPO, "tests", "host_math", "_build")
LIB = os.path.join(OUT, "libquad_math_host.so")
CSRC = os.path.join(REPO, "apg_trajectory_tracking_amd", "csrc")
DEPS = [SRC, os.path.join(CSRC, "quad_math.h"), os.path.join(CSRC, "apg_device.h")]
MOD = {"translational_drag": [.1, .2, .3], "rotational_drag": [.01, .02, .03],
       "mass": 1.0}


@pytest.fixture(scope="module")
def hm():
    if not os.path.exists("/opt/rocm/bin/hipcc") and not os.path.exists(LIB):
        pytest.skip("no hipcc to compile the host harness")
    os.makedirs(OUT, exist_ok=True)
    if (not os.path.exists(LIB)
            or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in DEPS)):
        subprocess.run(
            ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared",
             "--cuda-host-only", "-I", os.path.join(REPO, "include"), "-I", CSRC,
             "-o", LIB, SRC], check=True, stdout=subprocess.DEVNULL,
            stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.hm_quad_rollout.restype = ctypes.c_double
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
