import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)"
    )
    # the CPU oracles are ~600 tiny eager ops per unroll: on a 256-CPU host the
    # default intra-op thread count makes them 10x SLOWER than 16 threads
    # (bench.py's thread sweep: 137 ms at 16 threads, 1 995 ms at 128)
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except ImportError:
        pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|) - the 'relative to tensor scale' error used for
    the 1e-4 fp32 parity bar of BASELINE.json:north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-30)
    return float(np.max(np.abs(a - b)) / denom)


def per_trajectory_err(got, want):
    """e_b = max|got_b - want_b| / max|want_b| for every trajectory b (leading
    axis); float64."""
    want = np.asarray(want, dtype=np.float64)
    B = want.shape[0]
    g, w = np.asarray(got, dtype=np.float64).reshape(B, -1), want.reshape(B, -1)
    return np.abs(g - w).max(1) / np.maximum(np.abs(w).max(1), 1e-30)


def assert_no_worse_than_fp32(dev, f32, f64, what, factor=3.0, worst_factor=4.0,
                              floor=1e-6, bar=1e-4, frac=1e-3):
    """VERDICT r3 #4a: the float64 oracle arbitrates.  Per-trajectory errors
    of the DEVICE result and of the float32 ORACLE (the reference's own
    arithmetic) are both taken against the float64 oracle.  On the worst
    `frac` of the trajectories (each side's own worst set: the errors are
    rounding noise, the same trajectory is not the worst for both)
      * the MEAN device error of that set is at most `factor` x the float32
        oracle's (+ `floor`, 8 ulp: below that both are pure rounding);
        measured on the MI355X: 2.1 x for the quadrotor rollouts (hardware
        sin / cos), 1.2-2.6 x for the fixed wing,
      * the single worst trajectory at most `worst_factor` x the oracle's worst
        (the maximum of ~100 000 noisy values: measured 1.0-2.8 x),
      * and EVERY trajectory meets north_star's 1e-4 by itself - there is no
        looser per-trajectory bound anywhere in the suite.
    Kernels with the policy inside run their layers as fp16-split products with
    a fast tanh: ~3 x the float32 noise on the states (5e-6 at worst), their
    callers pass factor=4.  Returns the statistics (pytest -s prints them)."""
    e_dev = np.sort(per_trajectory_err(dev, f64))
    e_f32 = np.sort(per_trajectory_err(f32, f64))
    n = max(1, int(len(e_dev) * frac))
    stats = dict(what=what, worst_dev=e_dev[-1], worst_f32=e_f32[-1],
                 tail_mean_dev=e_dev[-n:].mean(), tail_mean_f32=e_f32[-n:].mean(),
                 median_dev=np.median(e_dev), median_f32=np.median(e_f32), tail=n)
    print("fp64 arbiter:", {k: (float("%.3g" % v) if isinstance(v, float) else v)
                            for k, v in stats.items()})
    assert e_dev[-1] < bar, stats
    assert e_dev[-n:].mean() <= factor * e_f32[-n:].mean() + floor, stats
    assert e_dev[-1] <= worst_factor * e_f32[-1] + floor, stats
    return stats


def per_row_err(got, want):
    """e_r = max|got_r - want_r| / max|want_r| for every OUTPUT ROW r of a
    parameter (leading axis; a bias is one row per element); float64."""
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    if want.ndim == 1:
        want, got = want[:, None], got[:, None]
    R = want.shape[0]
    g, w = got.reshape(R, -1), want.reshape(R, -1)
    return np.abs(g - w).max(1), np.abs(w).max(1)


def param_row_stats(dev, f32, f64, factor=4.0, eps=1e-4):
    """{parameter: dict(worst error / row scale, worst error / tensor scale, worst
    and median err_dev / err_f32 over the rows)} - the numbers behind
    assert_param_rows_no_worse_than_fp32 (pytest -s prints them)."""
    out = {}
    for k, w in f64.items():
        ed, scale = per_row_err(dev[k], w)
        ef, _ = per_row_err(f32[k], w)
        tmax = np.abs(np.asarray(w)).max()
        ratio = ed / np.maximum(ef, 1e-300)
        # VERDICT r5 next #4c: the bound WITHOUT its tensor-scale floor
        # (factor err_f32 + eps |row|): how many rows miss it, by how much, and
        # the floor (in units of the tensor's largest entry) that would just hold
        bare = factor * ef + eps * scale
        over = np.maximum(ed - bare, 0.0)
        out[k] = dict(rel_row=float((ed / np.maximum(scale, 1e-300)).max()),
                      rel_tensor=float(ed.max() / tmax),
                      f32_rel_tensor=float(ef.max() / tmax),
                      ratio_max=float(ratio.max()), ratio_median=float(np.median(ratio)),
                      rows=int(len(ed)), rows_needing_floor=int((over > 0).sum()),
                      worst_over_bare_bound=float((ed / np.maximum(bare, 1e-300)).max()),
                      floor_needed_rel_tensor=float(over.max() / tmax))
    return out


def assert_param_rows_no_worse_than_fp32(dev, f32, f64, what, factor=4.0, eps=1e-4,
                                         tensor_eps=2e-6, bias_tensor_eps=1.5e-5,
                                         report_only=False):
    """VERDICT r4 next #2a: the float64 oracle arbitrates PARAMETER gradients
    per output row.  `dev`, `f32`, `f64`: {parameter name: gradient} of the
    kernels, of float32 autograd (the reference's arithmetic) and of float64
    autograd.  For every row r of every parameter (a bias: every element)
        err_dev(r) <= factor * err_f32(r) + eps * |row| + t_eps * |tensor|
    with err = max abs deviation over the row, |row| / |tensor| the largest
    float64 magnitude of the row / of the whole parameter; eps = north_star's
    1e-4, applied to every row's OWN scale; t_eps = 2e-6 for weight matrices,
    1.5e-5 for biases.
    What t_eps is: the kernels evaluate the policy with fp16-split products
    and a fast tanh - per TRAJECTORY 3-4 x float32's noise on states and
    cotangents (assert_no_worse_than_fp32, factor 4; ~1e-5 of a trajectory's
    own cotangent after ten steps of dynamics).  A gradient element is a sum
    of 65 536 (x 10) such terms of both signs: its noise is ~1e-5 x rms(term)
    x sqrt(N) in ABSOLUTE terms, whatever is left of the sum itself after
    cancellation.  A bias element is ONE such sum (measured: up to 7e-6 of the
    bias vector's largest entry, identically in the plane path with its exact
    float accumulation; float32 autograd, summing exactly rounded terms
    pairwise, stays at 6e-7); a weight row has 15-224 of them and its largest
    entry rarely cancels (measured <= 2.4e-6 of the tensor where a row misses
    1e-4 of itself).  The fixed-point accumulators' unit - set per workgroup
    and layer by the largest cotangent - shows up here as an error that does
    NOT shrink with the row: round 5 found exactly that in the concurrent
    kernel's head block and biases (x1e3 outlier per workgroup: 27 % of a small
    row's own scale, biases 40 x the plane path's) and removed it (per-row
    exponents, per-wave float bias sums); tests/test_gpu_round5.py runs the
    plane path beside the in-sweep path and prints both.
    A per-tensor max norm alone (conftest.rel_err) cannot see a row quantised
    at a unit set by another row's (or another trajectory's) magnitude; this
    can.  Returns param_row_stats."""
    stats = param_row_stats(dev, f32, f64, factor, eps)
    print("fp64 row arbiter:", what, {k: {a: float("%.2g" % b) for a, b in v.items()}
                                      for k, v in stats.items()})
    if report_only:
        return stats
    for k, w in f64.items():
        ed, scale = per_row_err(dev[k], w)
        ef, _ = per_row_err(f32[k], w)
        tmax = np.abs(np.asarray(w)).max()
        te = bias_tensor_eps if np.asarray(w).ndim == 1 else tensor_eps
        bound = factor * ef + eps * scale + te * tmax
        ratio = ed / np.maximum(bound, 1e-300)
        r = int(ratio.argmax())
        assert ratio.max() <= 1.0, (
            what, k, "row", r, dict(err_dev=ed[r], err_f32=ef[r], row_scale=scale[r],
                                    tensor_scale=float(tmax)))
    return stats


# ---- fixed-wing closed loop (G15): shared by the CPU and the GPU tests ------
def wing_loop_policy(device="cpu"):
    """The controller the reference ships (trained_models/wing), rebuilt from
    the state_dict recorded in checkpoints.npz (G9)."""
    import torch
    from apg_trajectory_tracking_amd.checkpoint import build_policy
    ck = load_golden("checkpoints.npz")
    sd = {k[len("wing.w."):]: torch.from_numpy(ck[k]) for k in ck.files
          if k.startswith("wing.w.")}
    return build_policy("wing", sd).to(device).eval()


def wing_loop_case(g, case):
    """(targets [B,n,3] float32, keyword settings, modified parameters) of a
    G15 case."""
    mp = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in g[f"{case}.modified"]}
    kw = dict(max_steps=int(g[f"{case}.max_steps"]),
              thresh_div=float(g[f"{case}.thresh_div"]),
              thresh_stable=float(g[f"{case}.thresh_stable"]),
              test_time=int(g[f"{case}.test_time"]))
    return g[f"{case}.targets"], kw, mp


def oracle_wing_closed_loop(net, targets, dt, params, mean, std, data_dt=0.05,
                            data_horizon=10, state0=None, max_steps=1000,
                            thresh_div=10.0, thresh_stable=0.8, test_time=0,
                            want_trajectory=False, modified_params=None, learnt=None):
    """The oracle's closed loop behind the signature and the output layout of
    functional.wing_mlp_closed_loop: lets the CPU suite run the evaluator's
    host logic without the kernel (tests only)."""
    import copy
    import torch
    from oracle import torch_port as tp
    out = tp.wing_closed_loop(
        copy.deepcopy(net).cpu(),
        learnt if learnt is not None else tp.WingOracle(modified_params=modified_params),
        targets.cpu(), dt, mean, std, data_dt, data_horizon, max_steps, thresh_div,
        thresh_stable, test_time, state0=None if state0 is None else state0.cpu())
    res = dict(div_linear=out["div_linear"].t().float(),
               div_pass=out["div_pass"].t().float(),
               div_fail=out["div_fail"].t().float(),
               steps=out["steps"].to(torch.int32))
    if want_trajectory:
        res.update(drone=out["traj"].permute(1, 2, 0).contiguous(),
                   seen=out["seen"].permute(1, 2, 0).contiguous())
    return res


@pytest.fixture(autouse=True)
def _launch_form_is_not_left_to_timing():
    """TrainBase measures at a mode's first capture whether graph replays or
    stream-order launches are faster ON THIS BOX and keeps the faster form
    (tests/test_gpu_round5.py tests that, setting `measure_launch_form` itself).
    Everywhere else a test that asks for `graph_steps` means the graph form: the
    default is pinned off, so that `trainer._graphs` does not depend on timing
    (seen once in round 6: the autoregressive step of a small batch chose stream
    order on one box and `test_static_shard_keeps_plane_copies...` failed)."""
    from apg_trajectory_tracking_amd.train_base import TrainBase
    before = TrainBase.MEASURE_LAUNCH_FORM
    TrainBase.MEASURE_LAUNCH_FORM = False
    yield
    TrainBase.MEASURE_LAUNCH_FORM = before
