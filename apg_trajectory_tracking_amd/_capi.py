"""ctypes binding of libapg_hip.so (the C ABI declared in include/apg.h).

There is deliberately NO fallback: if the shared library is missing or a
tensor is not on an MI355X, the call raises.  PyTorch is used only to own
device memory and to name the HIP stream work is enqueued on.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# APG_LIB: load an alternative build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("APG_LIB") or os.path.join(_HERE, "csrc", "libapg_hip.so")

LAYOUT_SOA = 0
LAYOUT_AOS = 1
LAYOUT_PACKED = 2   # rows [rows][B][C]: fused quad rollout only
MAX_HORIZON = 48
ROLLOUT_BLOCK = 64

_c_float_p = ctypes.c_void_p  # device pointers travel as integers


class ApgQuadParams(ctypes.Structure):
    _fields_ = [
        ("mass", ctypes.c_float),
        ("kinv", ctypes.c_float * 3),
        ("inertia", ctypes.c_float * 3),
        ("gravity", ctypes.c_float * 3),
        ("trans_drag", ctypes.c_float * 3),
        ("rot_drag", ctypes.c_float * 3),
    ]


class ApgQuadLossWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float)
                for n in ("pos", "vel", "av", "rates", "thrust")]


WING_PARAM_FIELDS = (
    "mass", "I_xx", "I_yy", "I_zz", "I_xz", "rho", "S", "c", "b", "g",
    "CL0", "CL_alpha", "CL_q", "CL_del_e",
    "CD0", "CD_alpha", "CD_q", "CD_del_e",
    "CY0", "CY_beta", "CY_p", "CY_r", "CY_del_a", "CY_del_r",
    "Cl0", "Cl_beta", "Cl_p", "Cl_r", "Cl_del_a", "Cl_del_r",
    "Cm0", "Cm_alpha", "Cm_q", "Cm_del_e",
    "Cn0", "Cn_beta", "Cn_p", "Cn_r", "Cn_del_a", "Cn_del_r",
    "epsilon",
)


class ApgWingParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in WING_PARAM_FIELDS]


class ApgWingLossWeights(ctypes.Structure):
    _fields_ = [("pos", ctypes.c_float), ("action", ctypes.c_float)]


class ApgDeferredLoss(ctypes.Structure):
    _fields_ = [("prev_partials", ctypes.c_void_p),
                ("prev_count", ctypes.c_int),
                ("prev_loss", ctypes.c_void_p)]


class ApgLearntResidual(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "linear_at", "w1", "b1", "w2", "b2")]


class ApgLstmPolicy(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "conv_w", "conv_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_out", "b_out")]


class ApgLstmPolicyGrads(ctypes.Structure):
    _fields_ = ApgLstmPolicy._fields_


class ApgLstmStepTail(ctypes.Structure):
    """include/apg.h: what apg_quad_lstm_step_tail does after the products."""
    _fields_ = [("grad", ApgLstmPolicyGrads), ("ih_hh", ctypes.c_void_p),
                ("conv_pos", ctypes.c_void_p), ("update", ctypes.c_int),
                ("lr", ctypes.c_double), ("momentum", ctypes.c_double),
                ("param", ApgLstmPolicyGrads), ("mom", ApgLstmPolicyGrads),
                ("tables_fwd", ctypes.c_void_p), ("tables_bwd", ctypes.c_void_p),
                ("loss_partials", ctypes.c_void_p), ("n_partials", ctypes.c_int),
                ("loss", ctypes.c_void_p), ("loss_sum", ctypes.c_void_p),
                ("applied", ctypes.c_int)]


class ApgMlpPolicy(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2",
        "w_3", "b_3", "w_out", "b_out")]


class ApgMlpPolicyGrads(ctypes.Structure):
    """Where apg_quad_mlp_concurrent_step puts each parameter's gradient."""
    _fields_ = ApgMlpPolicy._fields_


class ApgStepEvents(ctypes.Structure):
    """hipEvent_t handles a pipelined caller hands to the training step."""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "inputs_ready", "after_forward", "after_reverse")]


class ApgBatchRows(ctypes.Structure):
    """A minibatch named by row numbers into the data set's tensors
    (apg_quad_mlp_concurrent_train_step_rows)."""
    _fields_ = ([("index", ctypes.c_void_p)]
                + [(n, ctypes.c_void_p) for n in ("normed", "state0", "in_ref", "ref")]
                + [(n, ctypes.c_int) for n in (
                    "ld_normed", "ld_state0", "ld_in_ref", "ld_ref")]
                + [("n_rows", ctypes.c_longlong), ("running_loss", ctypes.c_void_p)])


class ApgMlpSgdUpdate(ctypes.Structure):
    """apg_quad_mlp_concurrent_train_step's optimizer part: momentum SGD on
    the policy's tensors and the optimizer's momentum buffers."""
    _fields_ = [("lr", ctypes.c_double), ("momentum", ctypes.c_double),
                ("param", ApgMlpPolicyGrads), ("momentum_buf", ApgMlpPolicyGrads),
                ("resident", ctypes.c_int)]


class ApgWingPolicy(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "w_s", "b_s", "w_r", "b_r", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3",
        "w_out", "b_out")]


class ApgGemmProblem(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p),
                ("bdesc", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("bias_out", ctypes.c_void_p), ("N", ctypes.c_longlong),
                ("M", ctypes.c_int), ("S", ctypes.c_int), ("J", ctypes.c_int),
                ("sdiv", ctypes.c_int), ("with_ones", ctypes.c_int),
                ("b_planes", ctypes.c_int), ("ldc", ctypes.c_int)]


class ApgSoaItem(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("index", ctypes.c_void_p),
                ("dst", ctypes.c_void_p), ("R", ctypes.c_int), ("ld", ctypes.c_int)]


class ApgCartpoleParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in (
        "masscart", "masspole", "length", "max_force_mag", "friction",
        "gravity")]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float

# name -> argtypes  (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "apg_quad_step_fwd": [_P, _P, _F, ctypes.POINTER(ApgQuadParams), _I, _I,
                          _P, _P],
    "apg_quad_step_bwd": [_P, _P, _F, ctypes.POINTER(ApgQuadParams), _I, _I,
                          _P, _P, _P, _P],
    "apg_quad_rollout_fwd_bwd": [
        _P, _P, _P, _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), _I, _I, _I, _P, _P, _P, _P, _P,
        ctypes.POINTER(ApgDeferredLoss), _P],
    "apg_quad_rollout_fwd": [_P, _P, _F, ctypes.POINTER(ApgQuadParams), _I,
                             _I, _I, _P, _P],
    "apg_quad_learnt_rollout_fwd_bwd": [
        _P, _P, _P, _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgLearntResidual), ctypes.POINTER(ApgQuadLossWeights),
        _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "apg_quad_loss_fwd_bwd": [
        _P, _P, _I, _P, ctypes.POINTER(ApgQuadLossWeights), _I, _I, _I, _P,
        _P, _P, _P, _P],
    "apg_quad_features_fwd": [_P, _I, _I, _P, _P],
    "apg_quad_features_bwd": [_P, _P, _I, _I, _P, _P],
    "apg_quad_lstm_rollout_fwd": [
        _P, _P, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgLstmPolicy), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P,
        _P],
    "apg_quad_lstm_workspace_floats": [],
    "apg_quad_lstm_loss_partials_count": [_I],
    "apg_quad_lstm_rollout_bwd": [
        _P, _P, _P, _P, _I, _P, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), ctypes.POINTER(ApgLstmPolicy),
        _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_cot_amax_floats": [_I],
    "apg_quad_lstm_tables_floats": [_I],
    "apg_quad_lstm_pack_tables": [ctypes.POINTER(ApgLstmPolicy), _P, _P, _P],
    "apg_quad_lstm_rollout_fwd_packed": [
        _P, _P, _P, _P, _F, ctypes.POINTER(ApgQuadParams), _P, _I, _I,
        _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_rollout_bwd_packed": [
        _P, _P, _P, _P, _I, _P, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), _P,
        _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_step_tail": [ctypes.POINTER(ApgLstmStepTail), _P],
    "apg_quad_lstm_rollout_fwd_rows": [
        ctypes.POINTER(ApgBatchRows), _P, _P, _F, ctypes.POINTER(ApgQuadParams), _P, _I, _I,
        _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_rollout_bwd_rows": [
        ctypes.POINTER(ApgBatchRows), _I, _P, _P, _P, _P, _P, _P, _F,
        ctypes.POINTER(ApgQuadParams), ctypes.POINTER(ApgQuadLossWeights), _P,
        _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_conv_wgrad_partials_floats": [_I],
    "apg_quad_lstm_conv_wgrad": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P],
    "apg_quad_lstm_gate_wgrad_partials_floats": [_I],
    "apg_quad_lstm_wgrads": [
        _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(ApgLstmPolicy), _P, _I, _I,
        _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(ApgLstmStepTail), _P],
    "apg_quad_lstm_gate_wgrad": [
        _P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(ApgLstmPolicy), _P, _I, _I,
        _P, _P, _P, _P, _P, _P],
    "apg_quad_mlp_rollout_fwd": [
        _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgMlpPolicy), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_mlp_rollout_fwd_inplace_ref": [
        _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgMlpPolicy), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_rollout_fwd_inplace_ref": [
        _P, _P, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgLstmPolicy), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P,
        _P],
    "apg_quad_mlp_workspace_floats": [],
    "apg_quad_mlp_loss_partials_count": [_I],
    "apg_quad_mlp_step_workspace_floats": [],
    "apg_quad_mlp_step_partials_floats": [_I],
    "apg_quad_mlp_concurrent_step": [
        _P, _P, _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), ctypes.POINTER(ApgMlpPolicy), _I, _I,
        _P, _P, _P, _P, _P, ctypes.POINTER(ApgMlpPolicyGrads), _P, _P, _P, _P, _P],
    "apg_quad_mlp_concurrent_train_step": [
        _P, _P, _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), ctypes.POINTER(ApgMlpPolicy), _I, _I,
        _P, _P, _P, _P, _P, ctypes.POINTER(ApgMlpPolicyGrads), _P, _P, _P,
        ctypes.POINTER(ApgMlpSgdUpdate), ctypes.POINTER(ApgStepEvents), _P],
    "apg_quad_mlp_concurrent_train_step_rows": [
        ctypes.POINTER(ApgBatchRows), _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), ctypes.POINTER(ApgMlpPolicy), _I, _I,
        _P, _P, _P, _P, _P, ctypes.POINTER(ApgMlpPolicyGrads), _P, _P, _P,
        ctypes.POINTER(ApgMlpSgdUpdate), ctypes.POINTER(ApgStepEvents), _P],
    "apg_quad_mlp_rollout_step_workspace_floats": [],
    "apg_quad_mlp_rollout_step_partials_floats": [_I],
    "apg_quad_mlp_rollout_train_step": [
        _P, _P, _P, _I, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgQuadLossWeights), ctypes.POINTER(ApgMlpPolicy), _I, _I,
        _P, _P, _P, _P, _P, _P, ctypes.POINTER(ApgMlpPolicyGrads), _P, _P, _P,
        ctypes.POINTER(ApgMlpSgdUpdate), _P],
    "apg_quad_mlp_closed_loop": [
        _P, _I, _F, ctypes.POINTER(ApgQuadParams), ctypes.POINTER(ApgMlpPolicy),
        _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_closed_loop": [
        _P, _I, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgLstmPolicy), _I, _I, _I, _F, _F, _I, _P, _P, _P, _P,
        _P, _P, _P],
    "apg_quad_mlp_closed_loop_env": [
        _P, _I, _F, ctypes.POINTER(ApgQuadParams), ctypes.POINTER(ApgLearntResidual),
        ctypes.POINTER(ApgMlpPolicy), _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "apg_quad_lstm_closed_loop_env": [
        _P, _I, _P, _P, _F, ctypes.POINTER(ApgQuadParams),
        ctypes.POINTER(ApgLearntResidual), ctypes.POINTER(ApgLstmPolicy), _I, _I, _I,
        _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "apg_planes_gemm_grouped": [ctypes.POINTER(ApgGemmProblem), _I, _P, _I, _P],
    "apg_to_soa": [_P, _P, _I, _I, _I, _P, _P],
    "apg_to_soa_multi": [ctypes.POINTER(ApgSoaItem), _I, _I, _P],
    "apg_wing_policy_workspace_floats": [],
    "apg_wing_policy_fwd": [_P, _P, ctypes.POINTER(ApgWingPolicy), _I, _I, _P, _P,
                            _P, _P, _P],
    "apg_wing_policy_bwd": [_P, _P, _P, _P, ctypes.POINTER(ApgWingPolicy), _I, _I,
                            _P, _P, _P, _P],
    "apg_linear_wgrad_workspace_floats": [_I, _I],
    "apg_linear_wgrad": [_P, _P, ctypes.c_longlong, _I, _I, _P, _P, _P, _P],
    "apg_wing_learnt_param_count": [],
    "apg_wing_learnt_workspace_floats": [_I],
    "apg_wing_learnt_step_fwd": [
        _P, _P, _F, ctypes.POINTER(ApgWingParams), ctypes.POINTER(ctypes.c_float),
        _I, _P, _P],
    "apg_wing_learnt_step_bwd": [
        _P, _P, _F, ctypes.POINTER(ApgWingParams), ctypes.POINTER(ctypes.c_float),
        _I, _P, _P, _P, _P, _P, _P],
    "apg_wing_mlp_closed_loop": [
        _P, _I, _P, _F, ctypes.POINTER(ApgWingParams),
        ctypes.POINTER(ApgWingPolicy), ctypes.POINTER(ctypes.c_float),
        ctypes.POINTER(ctypes.c_float), _F, _I, _I, _I, _F, _F, _I, _P, _P, _P,
        _P, _P, _P, _P, _P],
    "apg_wing_mlp_closed_loop_env": [
        _P, _I, _P, _F, ctypes.POINTER(ApgWingParams), ctypes.POINTER(ctypes.c_float),
        ctypes.POINTER(ApgLearntResidual),
        ctypes.POINTER(ApgWingPolicy), ctypes.POINTER(ctypes.c_float),
        ctypes.POINTER(ctypes.c_float), _F, _I, _I, _I, _F, _F, _I, _P, _P, _P,
        _P, _P, _P, _P, _P],
    "apg_planes_gemm_workspace_floats": [_I, _I, _I, _I],
    "apg_planes_gemm_default_wgs": [_I, _I, _I, _I],
    "apg_planes_gemm_multi_workspace_floats": [ctypes.POINTER(ApgGemmProblem), _I],
    "apg_planes_gemm_multi": [ctypes.POINTER(ApgGemmProblem), _I, _P, _P],
    "apg_planes_gemm": [_P, _I, _I, _P, _P, _I, _I, _I, _I,
                        ctypes.c_longlong, _P, _I, _P, _I, _P, _P],
    "apg_wing_step_fwd": [_P, _P, _F, ctypes.POINTER(ApgWingParams), _I, _I,
                          _P, _P],
    "apg_wing_step_bwd": [_P, _P, _F, ctypes.POINTER(ApgWingParams), _I, _I,
                          _P, _P, _P, _P],
    "apg_wing_rollout_fwd_bwd": [
        _P, _P, _P, _F, ctypes.POINTER(ApgWingParams),
        ctypes.POINTER(ApgWingLossWeights), _I, _I, _I, _P, _P, _P, _P, _P,
        ctypes.POINTER(ApgDeferredLoss), _P],
    "apg_wing_rollout_fwd": [_P, _P, _F, ctypes.POINTER(ApgWingParams), _I,
                             _I, _I, _P, _P],
    "apg_wing_set_two_per_lane": [_I],
    "apg_cartpole_step_fwd": [_P, _P, _F, ctypes.POINTER(ApgCartpoleParams),
                              _I, _I, _P, _P],
    "apg_cartpole_step_bwd": [_P, _P, _F, ctypes.POINTER(ApgCartpoleParams),
                              _I, _I, _P, _P, _P, _P],
    "apg_cartpole_rollout_fwd_bwd": [
        _P, _P, _F, ctypes.POINTER(ApgCartpoleParams), _I, _I, _I, _P, _P,
        _P, _P, _P, _P],
    "apg_cartpole_rollout_fwd": [_P, _P, _F, ctypes.POINTER(ApgCartpoleParams),
                                 _I, _I, _I, _P, _P],
    "apg_reduce_loss_partials": [_P, _I, _P, _P],
    "apg_loss_partials_count": [_I],
    "apg_stream_copy": [_P, _P, ctypes.c_longlong, _P],
    "apg_stream_copy_shape": [_P, _P, ctypes.c_longlong, _I, _P],
    "apg_stream_rows_probe": [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _I, _P],
    "apg_version": [],
    "apg_last_error_string": [],
}
_RESTYPES = {"apg_last_error_string": ctypes.c_char_p,
             "apg_planes_gemm_multi_workspace_floats": ctypes.c_longlong,
             "apg_quad_mlp_step_partials_floats": ctypes.c_longlong,
             "apg_quad_mlp_rollout_step_partials_floats": ctypes.c_longlong,
             "apg_linear_wgrad_workspace_floats": ctypes.c_longlong}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with "
                "`python -m apg_trajectory_tracking_amd.build` "
                "(there is no CPU / PyTorch fallback for the APG kernels)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a symbol is absent
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().apg_last_error_string().decode()
        exc = ValueError if code == -1 else RuntimeError
        raise exc(f"{what} failed ({code}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def require_device(*tensors):
    """Every tensor must be a contiguous fp32 tensor on one HIP device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "apg_trajectory_tracking_amd kernels run on an MI355X only: "
                f"got a tensor on {t.device} (no CPU fallback exists)")
        if t.dtype != torch.float32:
            raise TypeError(f"fp32 tensors required, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("contiguous tensors required")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("all tensors must live on the same device")
    return dev


def loss_partials_count(B):
    return 1 if B <= 0 else (B + ROLLOUT_BLOCK - 1) // ROLLOUT_BLOCK
