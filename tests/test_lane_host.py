"""The per-trajectory rollout of csrc/quad_lane_pk.h (packed-fp32 grouping of
the quadrotor step, loss and adjoint - the code the packed rollout kernel runs
per lane) compiled for the HOST and pinned against the golden vectors recorded
from the reference (G1 single step + VJPs, G2 rollout with gradients), so the
arithmetic is verified without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden, rel_err

SRC = os.path.join(REPO, "tests", "host_lane", "quad_lane_host.hip")
OUT = os.path.join(REPO, "tests", "host_lane", "_build")
LIB = os.path.join(OUT, "libquad_lane_host.so")
HDR = os.path.join(REPO, "apg_trajectory_tracking_amd", "csrc", "quad_lane_pk.h")
MOD = {"translational_drag": [.1, .2, .3], "rotational_drag": [.01, .02, .03],
       "mass": 1.0}


@pytest.fixture(scope="module")
def lane():
    os.makedirs(OUT, exist_ok=True)
    stale = (not os.path.exists(LIB) or
             os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)))
    if stale:
        subprocess.run(
            ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared",
             "--cuda-host-only", "-I", os.path.join(REPO, "include"), "-I",
             os.path.dirname(HDR), "-o", LIB, SRC], check=True,
            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(LIB)
    lib.quad_lane_rollout_host.restype = ctypes.c_double
    return lib


def _params(mp):
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    return FlightmareDynamics(modified_params=mp).params


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, np.float32)


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
@pytest.mark.parametrize("dt", [0.05, 0.1])
def test_lane_step_and_vjp(lane, tag, mp, dt):
    g = load_golden("quad_step.npz")
    key = f"{tag}_dt{int(round(dt*100)):03d}"
    s, a = _f(g["state"]), _f(g["action"])
    par = _params(mp)
    for i, c in enumerate(g["cot"]):
        nxt, gs, ga = np.empty_like(s), np.empty_like(s), np.empty_like(a)
        lane.quad_lane_step_host(_p(s), _p(a), ctypes.c_float(dt), ctypes.byref(par),
                                 s.shape[0], _p(_f(c)), _p(nxt), _p(gs), _p(ga))
        assert rel_err(nxt, g[key + "_next"]) < 2e-6
        assert rel_err(gs, g[key + "_gstate"][i]) < 1e-5
        assert rel_err(ga, g[key + "_gaction"][i]) < 1e-5


def test_lane_large_angles(lane):
    """The packed sin-cos keeps its accuracy far outside the first period."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(3)
    s = (rng.normal(size=(256, 12)) * 0.5).astype(np.float32)
    s[:, 3:6] = rng.uniform(-300, 300, size=(256, 3))
    a = rng.uniform(size=(256, 4)).astype(np.float32)
    nxt = np.empty_like(s)
    lane.quad_lane_step_host(_p(s), _p(a), ctypes.c_float(0.1),
                             ctypes.byref(_params({})), 256, None, _p(nxt), None, None)
    want = co.quad_step(s.astype(np.float64), a.astype(np.float64), 0.1,
                        dtype=np.float64)
    assert np.abs(nxt - want).max() < 2e-4      # |att| ~ 300: fp32 ulp is 3e-5


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
def test_lane_rollout(lane, tag, mp):
    from apg_trajectory_tracking_amd import functional as F
    g = load_golden("quad_rollout.npz")
    s0, act, ref = _f(g["state0"]), _f(g["actions"]), _f(g["ref"])
    B, H = act.shape[:2]
    st = np.empty((B, H, 12), np.float32)
    ga, gs = np.empty_like(act), np.empty_like(s0)
    w = F.quad_loss_weights()
    loss = lane.quad_lane_rollout_host(
        _p(s0), _p(act), _p(ref), ref.shape[2], ctypes.c_float(float(g["dt"])),
        ctypes.byref(_params(mp)), ctypes.byref(w), B, H, _p(st), _p(ga), _p(gs))
    assert rel_err(st, g[tag + "_states"]) < 1e-5
    assert abs(loss - g[tag + "_loss"]) / g[tag + "_loss"] < 1e-5
    assert rel_err(ga, g[tag + "_gactions"]) < 1e-5
    assert rel_err(gs, g[tag + "_gstate0"]) < 1e-5
