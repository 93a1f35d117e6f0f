"""ctypes loader for the C oracle (oracle/apg_oracle.c).  ORACLE = test
infrastructure: importable only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")


class OracleQuadCfg(ctypes.Structure):
    _fields_ = [("mass", ctypes.c_double), ("arm_length", ctypes.c_double),
                ("frame_inertia", ctypes.c_double * 3),
                ("gravity", ctypes.c_double * 3), ("kinv", ctypes.c_double * 3),
                ("rot_drag", ctypes.c_double * 3),
                ("trans_drag", ctypes.c_double * 3)]


DEFAULT = dict(mass=0.723, arm_length=0.31, frame_inertia=[4.5, 4.5, 7.0],
               gravity=[0, 0, -9.81], kinv_ang_vel_tau=[16.6, 16.6, 5.0],
               rotational_drag=[0, 0, 0], translational_drag=[0, 0, 0])


def make_cfg(modified_params=None):
    c = dict(DEFAULT)
    c.update(modified_params or {})
    return OracleQuadCfg(
        c["mass"], c["arm_length"], (ctypes.c_double * 3)(*c["frame_inertia"]),
        (ctypes.c_double * 3)(*c["gravity"]),
        (ctypes.c_double * 3)(*c["kinv_ang_vel_tau"]),
        (ctypes.c_double * 3)(*c["rotational_drag"]),
        (ctypes.c_double * 3)(*c["translational_drag"]))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = ctypes.CDLL(LIB)
        for sfx in ("f32", "f64"):
            getattr(_lib, f"oracle_quad_rollout_fwd_bwd_{sfx}").restype = ctypes.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def quad_step(state, action, dt, modified_params=None, dtype=np.float32):
    sfx = "f32" if dtype == np.float32 else "f64"
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    out = np.empty_like(s)
    cfg = make_cfg(modified_params)
    R = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    getattr(lib(), f"oracle_quad_step_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(out))
    return out


def quad_step_vjp(state, action, dt, gnext, modified_params=None,
                  dtype=np.float32):
    sfx = "f32" if dtype == np.float32 else "f64"
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    g = np.ascontiguousarray(gnext, dtype)
    gs, ga = np.empty_like(s), np.empty_like(a)
    cfg = make_cfg(modified_params)
    R = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    getattr(lib(), f"oracle_quad_step_vjp_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(g), _p(gs), _p(ga))
    return gs, ga


def quad_rollout_fwd_bwd(state0, actions, ref, dt, modified_params=None,
                         dtype=np.float32, want_states=True):
    """-> (states [B,H,12] or None, loss (float), gactions, gstate0)."""
    sfx = "f32" if dtype == np.float32 else "f64"
    s = np.ascontiguousarray(state0, dtype)
    a = np.ascontiguousarray(actions, dtype)
    r = np.ascontiguousarray(ref, dtype)
    B, H = a.shape[:2]
    states = np.empty((B, H, 12), dtype) if want_states else None
    ga, gs = np.empty_like(a), np.empty_like(s)
    cfg = make_cfg(modified_params)
    R = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    loss = getattr(lib(), f"oracle_quad_rollout_fwd_bwd_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), _p(r), R(dt), B, H, _p(states),
        _p(ga), _p(gs))
    return states, float(loss), ga, gs
