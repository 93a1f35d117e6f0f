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
        srcs = [os.path.join(HERE, f) for f in
                ("apg_oracle.c", "apg_oracle_wing.c", "Makefile")]
        if (not os.path.exists(LIB) or os.path.getmtime(LIB)
                < max(os.path.getmtime(f) for f in srcs if os.path.exists(f))):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _lib = ctypes.CDLL(LIB)
        for sfx in ("f32", "f64"):
            for sysname in ("quad", "wing", "cartpole"):
                getattr(_lib, f"oracle_{sysname}_rollout_fwd_bwd_{sfx}"
                        ).restype = ctypes.c_double
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


# ------------------------------------------------------------------ wing
WING_FIELDS = [
    "mass", "I_xx", "I_yy", "I_zz", "I_xz", "rho", "S", "c", "b", "g",
    "CL0", "CL_alpha", "CL_q", "CL_del_e", "CD0", "CD_alpha", "CD_q", "CD_del_e",
    "CY0", "CY_beta", "CY_p", "CY_r", "CY_del_a", "CY_del_r",
    "Cl0", "Cl_beta", "Cl_p", "Cl_r", "Cl_del_a", "Cl_del_r",
    "Cm0", "Cm_alpha", "Cm_q", "Cm_del_e",
    "Cn0", "Cn_beta", "Cn_p", "Cn_r", "Cn_del_a", "Cn_del_r", "epsilon"]

# neural_control/dynamics/config_fixed_wing.json
WING_DEFAULT = dict(
    mass=1.01, I_xx=0.04766, I_yy=0.05005, I_zz=0.09558, I_xz=-0.00105,
    rho=1.225, S=0.276, c=0.185, b=1.54, g=9.81,
    CL0=0.39, CL_alpha=4.5321, CL_q=0.318, CL_del_e=0.527,
    CD0=0.0765, CD_alpha=0.3346, CD_q=0.354, CD_del_e=0.004,
    CY0=0.0, CY_beta=-0.033, CY_p=-0.1, CY_r=0.039, CY_del_a=0.0, CY_del_r=0.225,
    Cl0=0.0, Cl_beta=-0.081, Cl_p=-0.529, Cl_r=0.159, Cl_del_a=-0.453,
    Cl_del_r=0.005,
    Cm0=0.02, Cm_alpha=-1.4037, Cm_q=-0.1324, Cm_del_e=-0.4236,
    Cn0=0.0, Cn_beta=0.189, Cn_p=-0.083, Cn_r=-0.948, Cn_del_a=-0.041,
    Cn_del_r=-0.077, epsilon=0.16534698176788384)


class OracleWingCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in WING_FIELDS]


def make_wing_cfg(modified_params=None):
    c = dict(WING_DEFAULT)
    c.update(modified_params or {})
    return OracleWingCfg(*[float(c[n]) for n in WING_FIELDS])


def _real(dtype):
    return ("f32", ctypes.c_float) if dtype == np.float32 else ("f64", ctypes.c_double)


def wing_step(state, action, dt, modified_params=None, dtype=np.float32):
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    out = np.empty_like(s)
    cfg = make_wing_cfg(modified_params)
    getattr(lib(), f"oracle_wing_step_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(out))
    return out


def wing_step_vjp(state, action, dt, gnext, modified_params=None,
                  dtype=np.float32):
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    g = np.ascontiguousarray(gnext, dtype)
    gs, ga = np.empty_like(s), np.empty_like(a)
    cfg = make_wing_cfg(modified_params)
    getattr(lib(), f"oracle_wing_step_vjp_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(g), _p(gs), _p(ga))
    return gs, ga


def wing_rollout_fwd_bwd(state0, actions, ref, dt, modified_params=None,
                         dtype=np.float32, want_states=True):
    """-> (states [B,H,12] or None, loss, gactions, gstate0); ref [B,H,3]."""
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state0, dtype)
    a = np.ascontiguousarray(actions, dtype)
    r = np.ascontiguousarray(ref, dtype)
    B, H = a.shape[:2]
    states = np.empty((B, H, 12), dtype) if want_states else None
    ga, gs = np.empty_like(a), np.empty_like(s)
    cfg = make_wing_cfg(modified_params)
    loss = getattr(lib(), f"oracle_wing_rollout_fwd_bwd_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), _p(r), R(dt), B, H, _p(states),
        _p(ga), _p(gs))
    return states, float(loss), ga, gs


# -------------------------------------------------------------- cartpole
class OracleCartpoleCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                ("masscart", "masspole", "length", "max_force_mag", "friction",
                 "gravity")]


# neural_control/dynamics/config_cartpole.json; friction forced to 0.5
# (cartpole_dynamics.py:34), gravity 9.81 (:18)
CARTPOLE_DEFAULT = dict(masscart=1.0, masspole=0.1, length=0.5,
                        max_force_mag=30.0)


def make_cartpole_cfg(modified_params=None):
    c = dict(CARTPOLE_DEFAULT)
    c.update(modified_params or {})
    return OracleCartpoleCfg(c["masscart"], c["masspole"], c["length"],
                             c["max_force_mag"], 0.5, 9.81)


def cartpole_step(state, action, dt, modified_params=None, dtype=np.float32):
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    out = np.empty_like(s)
    cfg = make_cartpole_cfg(modified_params)
    getattr(lib(), f"oracle_cartpole_step_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(out))
    return out


def cartpole_step_vjp(state, action, dt, gnext, modified_params=None,
                      dtype=np.float32):
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state, dtype)
    a = np.ascontiguousarray(action, dtype)
    g = np.ascontiguousarray(gnext, dtype)
    gs, ga = np.empty_like(s), np.empty_like(a)
    cfg = make_cartpole_cfg(modified_params)
    getattr(lib(), f"oracle_cartpole_step_vjp_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), s.shape[0], _p(g), _p(gs), _p(ga))
    return gs, ga


def cartpole_rollout_fwd_bwd(state0, actions, dt, modified_params=None,
                             dtype=np.float32, ref_grad=True):
    """-> (states [B,H,4], loss, gactions [B,H,1], gstate0); the reference is
    make_reference(state0) (ref_grad: its dependence on state0 is
    differentiated, as in the trainer's autograd graph)."""
    sfx, R = _real(dtype)
    s = np.ascontiguousarray(state0, dtype)
    a = np.ascontiguousarray(actions, dtype)
    B, H = a.shape[:2]
    states = np.empty((B, H, 4), dtype)
    ga, gs = np.empty_like(a), np.empty_like(s)
    cfg = make_cartpole_cfg(modified_params)
    loss = getattr(lib(), f"oracle_cartpole_rollout_fwd_bwd_{sfx}")(
        ctypes.byref(cfg), _p(s), _p(a), R(dt), B, H, int(bool(ref_grad)),
        _p(states), _p(ga), _p(gs))
    return states, float(loss), ga, gs
