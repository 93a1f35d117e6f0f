"""Parity of the HIP kernels (through the C ABI) against the golden vectors
generated from the reference and against the PyTorch-eager oracle.

Bar (BASELINE.json north_star): states and gradients within 1e-4 relative,
fp32.  `rel_err` = max|a-b| / max|b| over a tensor."""
import numpy as np
import pytest
import torch

from conftest import assert_no_worse_than_fp32, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4
MOD = {"translational_drag": [.1, .2, .3], "rotational_drag": [.01, .02, .03],
       "mass": 1.0}
WMOD = {"mass": 1.4, "I_xz": -0.01, "CL0": 0.3, "rho": 1.0}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "needs an MI355X"
    return torch.device("cuda:0")


@pytest.fixture
def wing_pk():
    """Selects the fixed-wing rollout kernel for plane-layout batches through
    the C ABI's hook (apg_wing_set_two_per_lane: 1 = two trajectories per lane
    whenever possible, 0 = never); the shipped choice (2: by batch size) is
    restored afterwards."""
    from apg_trajectory_tracking_amd import _capi

    def choose(mode):
        _capi.check(_capi.lib().apg_wing_set_two_per_lane(int(mode)),
                    "apg_wing_set_two_per_lane")
    yield choose
    choose(2)


def D(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def soa_state(x):
    return x.t().contiguous()


def soa_seq(x):
    return x.permute(1, 2, 0).contiguous()


def aos_seq(x):
    return x.permute(2, 0, 1).contiguous()


# ------------------------------------------------------------------ quad
def test_quad_known_answer(dev):
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_step.npz")
    dyn = FlightmareDynamics()
    nxt = dyn.simulate_quadrotor(D(g["ka_action"], dev), D(g["ka_state"], dev),
                                 0.05)
    assert rel_err(N(nxt), g["ka_next"]) < 1e-6
    # B = 1 eval-time call (neural_control/environments/drone_env.py:99)
    nxt = dyn(D(g["state"][:1], dev), D(g["action"][:1], dev), 0.1)
    assert rel_err(N(nxt), g["b1_next"]) < 1e-6


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
@pytest.mark.parametrize("dt", [0.05, 0.1])
def test_quad_step_and_vjp(dev, tag, mp, dt):
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_step.npz")
    dyn = FlightmareDynamics(modified_params=dict(mp))
    s = D(g["state"], dev).requires_grad_(True)
    a = D(g["action"], dev).requires_grad_(True)
    nxt = dyn(s, a, dt)
    key = f"{tag}_dt{int(round(dt*100)):03d}"
    assert rel_err(N(nxt), g[key + "_next"]) < 1e-6
    for i, c in enumerate(g["cot"]):
        gs, ga = torch.autograd.grad(nxt, (s, a), D(c, dev), retain_graph=True)
        assert rel_err(N(gs), g[key + "_gstate"][i]) < 1e-5
        assert rel_err(N(ga), g[key + "_gaction"][i]) < 1e-5


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
def test_quad_rollout_golden(dev, layout, tag, mp):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_rollout.npz")
    dyn = FlightmareDynamics(modified_params=dict(mp))
    s0, a, r = D(g["state0"], dev), D(g["actions"], dev), D(g["ref"], dev)
    if layout == "soa":
        s0, a, r = soa_state(s0), soa_seq(a), soa_seq(r)
    res = F.quad_rollout_fwd_bwd(s0, a, r, float(g["dt"]), dyn.params,
                                 layout=layout, want_states=True)
    st, ga, gs = res["states"], res["grad_actions"], res["grad_state0"]
    if layout == "soa":
        st, ga, gs = aos_seq(st), aos_seq(ga), gs.t()
    assert rel_err(N(st), g[tag + "_states"]) < 1e-5
    assert abs(res["loss"].item() - g[tag + "_loss"]) / g[tag + "_loss"] < 1e-5
    assert rel_err(N(ga), g[tag + "_gactions"]) < TOL
    assert rel_err(N(gs), g[tag + "_gstate0"]) < TOL
    # the no-grad unroll gives the same states
    st2 = F.quad_rollout_fwd(s0, a, float(g["dt"]), dyn.params, layout=layout)
    if layout == "soa":
        st2 = aos_seq(st2)
    assert rel_err(N(st2), g[tag + "_states"]) < 1e-5


def test_quad_rollout_h5_ragged_and_packed_ref(dev):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_rollout.npz")
    dyn = FlightmareDynamics()
    s0, a, r = D(g["h5_state0"], dev), D(g["h5_actions"], dev), D(g["h5_ref"], dev)
    for layout in ("aos", "soa"):
        for packed in (False, True):
            rr = torch.cat((r[:, :, 0:3], r[:, :, 6:9]), 2).contiguous() \
                if packed else r
            args = (s0, a, rr)
            if layout == "soa":
                args = (soa_state(s0), soa_seq(a), soa_seq(rr))
            res = F.quad_rollout_fwd_bwd(*args, float(g["h5_dt"]), dyn.params,
                                         layout=layout, want_states=True)
            st, ga, gs = res["states"], res["grad_actions"], res["grad_state0"]
            if layout == "soa":
                st, ga, gs = aos_seq(st), aos_seq(ga), gs.t()
            assert rel_err(N(st), g["h5_states"]) < 1e-5
            assert abs(res["loss"].item() - g["h5_loss"]) / g["h5_loss"] < 1e-5
            assert rel_err(N(ga), g["h5_gactions"]) < TOL
            assert rel_err(N(gs), g["h5_gstate0"]) < TOL


@pytest.mark.parametrize("H", [1, 7, 10, 23, 48])
def test_quad_rollout_vs_oracle_any_horizon(dev, H):
    """H = 10 / 5 run the register-resident kernel, the others the LDS one."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from oracle import torch_port as tp
    B, dt = 200, 0.05
    d = synthetic.quad_polynomial_batch(B, H, dt, seed=100 + H)
    st, loss, ga, gs = tp.rollout_fwd_bwd(
        tp.QuadOracle(MOD), tp.quad_mpc_loss, d["state0"], d["actions"],
        d["ref"], dt)
    dyn = FlightmareDynamics(modified_params=dict(MOD))
    for layout in ("aos", "soa"):
        s0, a, r = (d["state0"].to(dev), d["actions"].to(dev), d["ref"].to(dev))
        if layout == "soa":
            s0, a, r = soa_state(s0), soa_seq(a), soa_seq(r)
        res = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout=layout,
                                     want_states=True)
        rs, rga, rgs = res["states"], res["grad_actions"], res["grad_state0"]
        if layout == "soa":
            rs, rga, rgs = aos_seq(rs), aos_seq(rga), rgs.t()
        assert rel_err(N(rs), st.numpy()) < TOL
        assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
        assert rel_err(N(rga), ga.numpy()) < TOL
        assert rel_err(N(rgs), gs.numpy()) < TOL


def test_quad_rollout_full_size_vs_oracle(dev):
    """BASELINE config 2: B = 65 536, H = 10, dt = 0.1 - the whole batch is
    checked against the CPU autograd oracle (takes ~1 s on the host)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from oracle import torch_port as tp
    B, H, dt = 65536, 10, 0.1
    d = synthetic.quad_polynomial_batch(B, H, dt, seed=0)
    st, loss, ga, gs = tp.rollout_fwd_bwd(
        tp.QuadOracle(), tp.quad_mpc_loss, d["state0"], d["actions"], d["ref"],
        dt)
    dyn = FlightmareDynamics()
    s0 = soa_state(d["state0"].to(dev))
    a = soa_seq(d["actions"].to(dev))
    r = soa_seq(d["ref"].to(dev))
    res = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="soa",
                                 want_states=True)
    assert rel_err(N(aos_seq(res["states"])), st.numpy()) < TOL
    assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
    assert rel_err(N(aos_seq(res["grad_actions"])), ga.numpy()) < TOL
    assert rel_err(N(res["grad_state0"].t()), gs.numpy()) < TOL
    # per trajectory, not only relative to the tensor maximum: the float64
    # oracle arbitrates between the kernel and the float32 oracle
    d64 = {k: v.double() for k, v in d.items()}
    st64, _, ga64, gs64 = tp.rollout_fwd_bwd(
        tp.QuadOracle(dtype=torch.float64), tp.quad_mpc_loss, d64["state0"],
        d64["actions"], d64["ref"], dt)
    assert_no_worse_than_fp32(N(aos_seq(res["grad_actions"])), ga.numpy(),
                              ga64.numpy(), "soa dL/dactions")
    assert_no_worse_than_fp32(N(res["grad_state0"].t()), gs.numpy(), gs64.numpy(),
                              "soa dL/dstate0")
    assert_no_worse_than_fp32(N(aos_seq(res["states"])), st.numpy(), st64.numpy(),
                              "soa states")
    # size-independent properties: additivity of the loss over a batch split
    # and permutation equivariance of the gradients
    half = B // 2
    res_a = F.quad_rollout_fwd_bwd(
        s0[:, :half].contiguous(), a[:, :, :half].contiguous(),
        r[:, :, :half].contiguous(), dt, dyn.params, layout="soa",
        want_states=True)   # same kernel instantiation => bit-identical
    res_b = F.quad_rollout_fwd_bwd(
        s0[:, half:].contiguous(), a[:, :, half:].contiguous(),
        r[:, :, half:].contiguous(), dt, dyn.params, layout="soa",
        want_states=True)
    tot = res_a["loss"].item() + res_b["loss"].item()
    assert abs(tot - res["loss"].item()) / res["loss"].item() < 1e-5
    assert torch.equal(res_a["grad_actions"], res["grad_actions"][:, :, :half])
    assert torch.equal(res_b["grad_actions"], res["grad_actions"][:, :, half:])


def _packed_inputs(d, dev):
    from apg_trajectory_tracking_amd import synthetic as sy
    ref6 = torch.cat((d["ref"][:, :, :3], d["ref"][:, :, 6:9]), 2)
    return (sy.to_packed_state(d["state0"]).to(dev),
            sy.to_packed_seq(d["actions"]).to(dev),
            sy.to_packed_seq(ref6).to(dev))


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", MOD)])
def test_quad_rollout_golden_packed_layout(dev, tag, mp):
    """G2 through APG_LAYOUT_PACKED (rows [rows][B][C], 16-byte accesses per
    lane - the layout bench.py measures), H = 10 and the ragged H = 5 case."""
    from apg_trajectory_tracking_amd import functional as F, synthetic as sy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_rollout.npz")
    dyn = FlightmareDynamics(modified_params=dict(mp))
    cases = [("", tag + "_", float(g["dt"]))]
    if tag == "def":
        cases.append(("h5_", "h5_", float(g["h5_dt"])))
    for pre, out, dt in cases:
        d = {k: torch.from_numpy(np.ascontiguousarray(g[pre + k]))
             for k in ("state0", "actions", "ref")}
        s0, a, r = _packed_inputs(d, dev)
        res = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="packed",
                                     want_states=True)
        assert res["states"].shape == (a.shape[0], 3, a.shape[1], 4)
        assert rel_err(N(sy.from_packed_seq(res["states"])), g[out + "states"]) < 1e-5
        assert abs(res["loss"].item() - g[out + "loss"]) / g[out + "loss"] < 1e-5
        assert rel_err(N(sy.from_packed_seq(res["grad_actions"])),
                       g[out + "gactions"]) < TOL
        assert rel_err(N(sy.from_packed_state(res["grad_state0"])),
                       g[out + "gstate0"]) < TOL


def test_quad_rollout_packed_full_size_vs_oracle(dev):
    """BASELINE config 2 through the packed layout: B = 65 536, H = 10 against
    the CPU autograd oracle, a ragged batch (dead lanes store out of range),
    and bit-identity with the plane-layout kernel's arithmetic is NOT asked
    for - only the 1e-4 bar."""
    from apg_trajectory_tracking_amd import functional as F, synthetic as sy
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from oracle import torch_port as tp
    H, dt = 10, 0.1
    dyn = FlightmareDynamics()
    for B, seed in ((65536, 0), (65536 - 37, 3), (1, 4), (63, 5), (65, 6)):
        d = sy.quad_polynomial_batch(B, H, dt, seed=seed)
        st, loss, ga, gs = tp.rollout_fwd_bwd(
            tp.QuadOracle(), tp.quad_mpc_loss, d["state0"], d["actions"],
            d["ref"], dt)
        s0, a, r = _packed_inputs(d, dev)
        # canaries behind every output: a dead lane must not write
        out = {"grad_actions": torch.full((H + 1, B, 4), 7.0, device=dev)[:H],
               "grad_state0": torch.full((4, B, 4), 7.0, device=dev)[:3],
               "states": torch.full((H + 1, 3, B, 4), 7.0, device=dev)[:H]}
        res = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="packed",
                                     want_states=True, out=out)
        for k in out:       # the row after the last one is untouched
            assert torch.all(out[k]._base[-1] == 7.0), k
        assert rel_err(N(sy.from_packed_seq(res["states"])), st.numpy()) < TOL
        assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
        dev_ga = N(sy.from_packed_seq(res["grad_actions"]))
        assert rel_err(dev_ga, ga.numpy()) < TOL
        assert rel_err(N(sy.from_packed_state(res["grad_state0"])), gs.numpy()) < TOL
        if B < 1000:
            continue
        # per trajectory: the float64 oracle arbitrates (conftest)
        d64 = {k: v.double() for k, v in d.items()}
        _, _, ga64, gs64 = tp.rollout_fwd_bwd(
            tp.QuadOracle(dtype=torch.float64), tp.quad_mpc_loss, d64["state0"],
            d64["actions"], d64["ref"], dt)
        assert_no_worse_than_fp32(dev_ga, ga.numpy(), ga64.numpy(),
                                  f"packed dL/dactions B={B}")
        assert_no_worse_than_fp32(N(sy.from_packed_state(res["grad_state0"])),
                                  gs.numpy(), gs64.numpy(), f"packed dL/dstate0 B={B}")
        if B != 65536:
            continue
        # the EXACT launch bench.py times - quad_rollout_rows_kernel<10, false>,
        # no states, grad_state0 = NULL - against the oracles as well (VERDICT
        # r3 #4b), not only against the other instantiation
        lean = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="packed",
                                      want_grad_state0=False)
        assert lean["grad_state0"] is None and lean.get("states") is None
        lean_ga = N(sy.from_packed_seq(lean["grad_actions"]))
        assert rel_err(lean_ga, ga.numpy()) < TOL
        assert abs(lean["loss"].item() - loss.item()) / loss.item() < TOL
        assert_no_worse_than_fp32(lean_ga, ga.numpy(), ga64.numpy(),
                                  "bench launch <10,false> dL/dactions")
    # additivity over a batch split / per-trajectory independence (B = 65 536)
    B = 65536
    d = sy.quad_polynomial_batch(B, H, dt, seed=0)
    s0, a, r = _packed_inputs(d, dev)
    res = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="packed")
    half = B // 2
    parts = [F.quad_rollout_fwd_bwd(
        s0[:, sl].contiguous(), a[:, sl].contiguous(), r[:, sl].contiguous(),
        dt, dyn.params, layout="packed")
        for sl in (slice(0, half), slice(half, B))]
    tot = parts[0]["loss"].item() + parts[1]["loss"].item()
    assert abs(tot - res["loss"].item()) / res["loss"].item() < 1e-5
    assert torch.equal(parts[0]["grad_actions"], res["grad_actions"][:, :half])
    assert torch.equal(parts[1]["grad_actions"], res["grad_actions"][:, half:])


def test_quad_packed_layout_errors(dev):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    dyn = FlightmareDynamics()
    z = lambda *s: torch.zeros(*s, device=dev)
    with pytest.raises(ValueError):   # only the register-resident horizons
        F.quad_rollout_fwd_bwd(z(3, 8, 4), z(7, 8, 4), z(7, 8, 6), 0.1,
                               dyn.params, layout="packed")
    with pytest.raises(ValueError):   # packed reference rows are [pos, vel]
        F.quad_rollout_fwd_bwd(z(3, 8, 4), z(10, 8, 4), z(10, 8, 9), 0.1,
                               dyn.params, layout="packed")
    with pytest.raises(ValueError):   # state must be [3, B, 4]
        F.quad_rollout_fwd_bwd(z(8, 12), z(10, 8, 4), z(10, 8, 6), 0.1,
                               dyn.params, layout="packed")
    with pytest.raises(ValueError):   # no packed variant of the no-grad unroll
        F.quad_rollout_fwd(z(3, 8, 4), z(10, 8, 4), 0.1, dyn.params,
                           layout="packed")
    res = F.quad_rollout_fwd_bwd(z(3, 0, 4), z(10, 0, 4), z(10, 0, 6), 0.1,
                                 dyn.params, layout="packed")
    assert res["loss"].item() == 0.0


def test_quad_empty_and_errors(dev):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    dyn = FlightmareDynamics()
    z = lambda *s: torch.zeros(*s, device=dev)
    res = F.quad_rollout_fwd_bwd(z(0, 12), z(0, 10, 4), z(0, 10, 9), 0.1,
                                 dyn.params)
    assert res["loss"].item() == 0.0
    assert dyn(z(0, 12), z(0, 4), 0.1).shape == (0, 12)
    with pytest.raises(ValueError):    # H beyond APG_MAX_HORIZON
        F.quad_rollout_fwd_bwd(z(4, 12), z(4, 49, 4), z(4, 49, 9), 0.1,
                               dyn.params)
    with pytest.raises(ValueError):    # bad ref_cols
        F.quad_rollout_fwd_bwd(z(4, 12), z(4, 10, 4), z(4, 10, 7), 0.1,
                               dyn.params)
    with pytest.raises(RuntimeError):  # CPU tensors: no fallback
        dyn(torch.zeros(4, 12), torch.zeros(4, 4), 0.1)


def test_quad_features(dev):
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    g = load_golden("features.npz")
    s = D(g["state"], dev).requires_grad_(True)
    f = state_preprocessing(s)
    assert rel_err(N(f), g["feat"]) < 1e-6
    (gs,) = torch.autograd.grad(f, s, D(g["cot"], dev))
    assert rel_err(N(gs), g["gstate"]) < 1e-5


def test_quad_loss(dev):
    from apg_trajectory_tracking_amd.drone_loss import quad_mpc_loss
    g = load_golden("losses.npz")
    st = D(g["q_states"], dev).requires_grad_(True)
    act = D(g["q_actions"], dev).requires_grad_(True)
    loss = quad_mpc_loss(st, D(g["q_ref"], dev), act)
    gs, ga = torch.autograd.grad(loss, (st, act))
    assert abs(loss.item() - g["q_loss"]) / g["q_loss"] < 1e-6
    assert rel_err(N(gs), g["q_gstates"]) < 1e-6
    assert rel_err(N(ga), g["q_gactions"]) < 1e-6


def test_quad_unroll_through_step_api_matches_fused(dev):
    """The reference's own loop (scripts/train_drone.py:181-197) written with
    the drop-in objects gives the same loss / gradients as the fused kernel."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.drone_loss import quad_mpc_loss
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_rollout.npz")
    dyn = FlightmareDynamics()
    s0 = D(g["state0"], dev).requires_grad_(True)
    a = D(g["actions"], dev).requires_grad_(True)
    r = D(g["ref"], dev)
    B, H = a.shape[:2]
    inter = torch.zeros(B, H, 12, device=dev)
    cur = s0
    for k in range(H):
        cur = dyn(cur, a[:, k], float(g["dt"]))
        inter[:, k] = cur
    loss = quad_mpc_loss(inter, r, a)
    loss.backward()
    assert rel_err(N(inter), g["def_states"]) < 1e-5
    assert abs(loss.item() - g["def_loss"]) / g["def_loss"] < 1e-5
    assert rel_err(N(a.grad), g["def_gactions"]) < TOL
    assert rel_err(N(s0.grad), g["def_gstate0"]) < TOL
    s1 = D(g["state0"], dev).requires_grad_(True)
    a1 = D(g["actions"], dev).requires_grad_(True)
    loss2 = F.quad_rollout_loss(s1, a1, r, float(g["dt"]), dyn.params)
    (3.0 * loss2).backward()
    assert abs(loss2.item() - g["def_loss"]) / g["def_loss"] < 1e-5
    assert rel_err(N(a1.grad) / 3.0, g["def_gactions"]) < TOL
    assert rel_err(N(s1.grad) / 3.0, g["def_gstate0"]) < TOL


# ------------------------------------------------------------------ wing
def test_wing_known_answers(dev):
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    g = load_golden("wing.npz")
    dyn = FixedWingDynamics()
    nxt = dyn.simulate_fixed_wing(D(g["ka_state"], dev), D(g["ka_action"], dev),
                                  0.05)
    assert rel_err(N(nxt), g["ka_next"]) < 1e-5
    # tests/run_wing_sim.py: 1001-step open-loop trace via the rollout kernel
    state = torch.zeros(1, 12, device=dev)
    state[0, 3] = 11.5
    act = D(g["sim_action"], dev).reshape(1, 1, 4).repeat(1, 1000, 1)
    sts = dyn.rollout(state, act, 1 / 100)
    full = torch.cat((state[:, None], sts), 1)[0]
    got = N(full)[g["sim_rows"]]
    assert rel_err(got, g["sim_states"]) < 5e-4   # 1000 chained fp32 steps


@pytest.mark.parametrize("tag,mp", [("def", {}), ("mod", WMOD)])
def test_wing_step_and_vjp(dev, tag, mp):
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    g = load_golden("wing.npz")
    dyn = FixedWingDynamics(modified_params=dict(mp))
    s = D(g["step_state"], dev).requires_grad_(True)
    a = D(g["step_action"], dev).requires_grad_(True)
    nxt = dyn(s, a, 0.05)
    assert rel_err(N(nxt), g[f"step_{tag}_next"]) < 1e-5
    for i, c in enumerate(g["step_cot"]):
        gs, ga = torch.autograd.grad(nxt, (s, a), D(c, dev), retain_graph=True)
        assert rel_err(N(gs), g[f"step_{tag}_gstate"][i]) < TOL
        assert rel_err(N(ga), g[f"step_{tag}_gaction"][i]) < TOL


@pytest.mark.parametrize("layout", ["aos", "soa", "soa_two_per_lane"])
@pytest.mark.parametrize("H", [20, 10])
def test_wing_rollout_golden(dev, layout, H, wing_pk):
    """G5 rollouts (incl. samples beyond the alpha / beta clamp).
    soa_two_per_lane: the packed-fp32 kernel (two trajectories per lane) that
    large batches use, forced here on the 64-trajectory golden batch."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    g = load_golden("wing.npz")
    p = f"h{H}_"
    dyn = FixedWingDynamics()
    s0, a, r = D(g[p + "state0"], dev), D(g[p + "actions"], dev), D(g[p + "ref"], dev)
    if layout == "soa_two_per_lane":
        wing_pk(1)
        layout = "soa"
    else:
        wing_pk(0)
    if layout == "soa":
        s0, a, r = soa_state(s0), soa_seq(a), soa_seq(r)
    res = F.wing_rollout_fwd_bwd(s0, a, r, 0.05, dyn.params, layout=layout,
                                 want_states=True)
    st, ga, gs = res["states"], res["grad_actions"], res["grad_state0"]
    if layout == "soa":
        st, ga, gs = aos_seq(st), aos_seq(ga), gs.t()
    assert rel_err(N(st), g[p + "states"]) < 1e-5
    assert abs(res["loss"].item() - g[p + "loss"]) / g[p + "loss"] < 1e-5
    assert rel_err(N(ga), g[p + "gactions"]) < TOL
    assert rel_err(N(gs), g[p + "gstate0"]) < TOL


def test_wing_rollout_large_vs_oracle(dev):
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from oracle import torch_port as tp
    B, H, dt = 8192, 20, 0.05
    d = synthetic.wing_batch(B, H, dt, seed=5)
    st, loss, ga, gs = tp.rollout_fwd_bwd(
        tp.WingOracle(), tp.fixed_wing_mpc_loss, d["state0"], d["actions"],
        d["ref"], dt)
    dyn = FixedWingDynamics()
    res = F.wing_rollout_fwd_bwd(
        soa_state(d["state0"].to(dev)), soa_seq(d["actions"].to(dev)),
        soa_seq(d["ref"].to(dev)), dt, dyn.params, layout="soa",
        want_states=True)
    assert rel_err(N(aos_seq(res["states"])), st.numpy()) < TOL
    assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
    assert rel_err(N(aos_seq(res["grad_actions"])), ga.numpy()) < TOL
    assert rel_err(N(res["grad_state0"].t()), gs.numpy()) < TOL


# -------------------------------------------------------------- cartpole
def test_cartpole(dev):
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import (
        CartpoleDynamics)
    g = load_golden("cartpole.npz")
    dyn = CartpoleDynamics()
    nxt = dyn(D(g["ka_state"], dev), D(g["ka_action"], dev), 0.02)
    assert rel_err(N(nxt), g["ka_next"]) < 1e-6
    s = D(g["state0"], dev).requires_grad_(True)
    a1 = D(g["actions"][:, 0], dev).requires_grad_(True)
    nxt = dyn(s, a1, 0.02)
    gs, ga = torch.autograd.grad(nxt, (s, a1), D(g["step_cot"], dev))
    assert rel_err(N(nxt), g["step_next"]) < 1e-5
    assert rel_err(N(gs), g["step_gstate"]) < TOL
    assert rel_err(N(ga), g["step_gaction"]) < TOL
    for layout in ("aos", "soa"):
        s0, a = D(g["state0"], dev), D(g["actions"], dev)
        if layout == "soa":
            s0, a = soa_state(s0), soa_seq(a)
        res = F.cartpole_rollout_fwd_bwd(s0, a, float(g["dt"]), dyn.params,
                                         layout=layout, want_states=True)
        st, ga, gs = res["states"], res["grad_actions"], res["grad_state0"]
        if layout == "soa":
            st, ga, gs = aos_seq(st), aos_seq(ga), gs.t()
        assert rel_err(N(st), g["states"]) < 1e-5
        assert abs(res["loss"].item() - g["loss"]) / g["loss"] < 1e-5
        assert rel_err(N(ga), g["gactions"]) < TOL
        assert rel_err(N(gs), g["gstate0"]) < TOL


def test_cartpole_rollout_fwd_matches_stepwise(dev):
    """apg_cartpole_rollout_fwd (no-grad unroll) = the golden states (G6) =
    the per-step kernel in a loop, both layouts."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import (
        CartpoleDynamics)
    g = load_golden("cartpole.npz")
    dyn = CartpoleDynamics()
    s0, a = D(g["state0"], dev), D(g["actions"], dev)
    dt = float(g["dt"])
    st = dyn.rollout(s0, a, dt)
    assert rel_err(N(st), g["states"]) < 1e-5
    st_soa = F.cartpole_rollout_fwd(soa_state(s0), soa_seq(a), dt, dyn.params,
                                    layout="soa")
    assert torch.equal(aos_seq(st_soa), st)
    cur = s0
    for k in range(a.shape[1]):
        cur = dyn(cur, a[:, k], dt)
        assert torch.equal(cur, st[:, k])
    assert F.cartpole_rollout_fwd(s0[:0], a[:0], dt, dyn.params).shape == (0, 5, 4)


@pytest.mark.parametrize("layout", ["soa", "packed"])
def test_deferred_loss_chain(dev, layout):
    """ApgDeferredLoss: step i's loss reduced inside step i+1's launch equals
    the eager two-kernel loss (same partials, fixed summation orders)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    dyn = FlightmareDynamics()
    B, H, dt = 4096 + 17, 10, 0.1
    plans, eager = [], []
    for seed in range(3):
        d = synthetic.quad_polynomial_batch(B, H, dt, seed=seed)
        if layout == "packed":
            t = _packed_inputs(d, dev)
        else:
            t = (soa_state(d["state0"].to(dev)), soa_seq(d["actions"].to(dev)),
                 soa_seq(d["ref"].to(dev)))
        eager.append(F.quad_rollout_fwd_bwd(*t, dt, dyn.params, layout=layout))
        plans.append(F.RolloutPlan("quad", *t, dt, dyn.params, layout=layout,
                                   loss_mode="deferred"))
    for p in plans:
        p.out["loss"].fill_(-1.0)
    prev = None
    for p in plans:
        p.launch(after=prev)
        prev = p
    prev.flush()
    torch.cuda.synchronize()
    for p, e in zip(plans, eager):
        assert abs(p.out["loss"].item() - e["loss"].item()) <= 1e-6 * abs(e["loss"].item())
        assert torch.equal(p.out["grad_actions"], e["grad_actions"])
    with pytest.raises(ValueError):   # a plan cannot defer onto itself
        plans[0].launch(after=plans[0])


# ------------------------------------------------------- edge / size cases
def test_quad_large_angles_and_rates(dev):
    """Attitudes of tens of radians and fast body rates: the branch-free
    sincos (Cody-Waite reduction) must stay within tolerance far outside the
    training distribution."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from oracle import torch_port as tp
    B, H, dt = 512, 10, 0.1
    d = synthetic.quad_polynomial_batch(B, H, dt, seed=77)
    g = torch.Generator().manual_seed(78)
    s0 = d["state0"].clone()
    s0[:, 3:6] = 30.0 * torch.randn(B, 3, generator=g)
    s0[:, 9:12] = 8.0 * torch.randn(B, 3, generator=g)
    s0[:64, 3:6] *= 100.0                       # |att| up to ~1e4 rad
    st, loss, ga, gs = tp.rollout_fwd_bwd(
        tp.QuadOracle(), tp.quad_mpc_loss, s0, d["actions"], d["ref"], dt)
    dyn = FlightmareDynamics()
    for layout in ("soa", "aos"):
        a = (s0.to(dev), d["actions"].to(dev), d["ref"].to(dev))
        if layout == "soa":
            a = (soa_state(a[0]), soa_seq(a[1]), soa_seq(a[2]))
        res = F.quad_rollout_fwd_bwd(*a, dt, dyn.params, layout=layout,
                                     want_states=True)
        rs, rga, rgs = res["states"], res["grad_actions"], res["grad_state0"]
        if layout == "soa":
            rs, rga, rgs = aos_seq(rs), aos_seq(rga), rgs.t()
        assert torch.isfinite(rs).all()
        assert rel_err(N(rs), st.numpy()) < TOL
        assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
        assert rel_err(N(rga), ga.numpy()) < TOL
        assert rel_err(N(rgs), gs.numpy()) < TOL


def test_quad_rollout_max_batch_chunk_additivity(dev):
    """B = 524 288 (BASELINE config 3's global batch on ONE GPU): the result
    equals the concatenation of eight independent 65 536-trajectory launches
    bit for bit (trajectories are independent; what N-GPU sharding relies on),
    and the loss is the sum of the shard losses."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    dyn = FlightmareDynamics()
    B, H, dt, n = 524288, 10, 0.1, 8
    d = synthetic.quad_polynomial_batch(B, H, dt, seed=9)
    s0 = soa_state(d["state0"]).to(dev)
    a = soa_seq(d["actions"]).to(dev)
    r = soa_seq(d["ref"][:, :, [0, 1, 2, 6, 7, 8]]).to(dev)
    full = F.quad_rollout_fwd_bwd(s0, a, r, dt, dyn.params, layout="soa")
    assert torch.isfinite(full["grad_actions"]).all()
    total, c = 0.0, B // n
    for i in range(n):
        part = F.quad_rollout_fwd_bwd(
            s0[:, i * c:(i + 1) * c].contiguous(),
            a[:, :, i * c:(i + 1) * c].contiguous(),
            r[:, :, i * c:(i + 1) * c].contiguous(), dt, dyn.params,
            layout="soa")
        total += part["loss"].item()
        assert torch.equal(part["grad_actions"],
                           full["grad_actions"][:, :, i * c:(i + 1) * c])
        assert torch.equal(part["grad_state0"],
                           full["grad_state0"][:, i * c:(i + 1) * c])
    assert abs(total - full["loss"].item()) / full["loss"].item() < 1e-5


@pytest.mark.parametrize("B,H", [(100, 7), (1, 20), (333, 13), (258, 13)])
def test_wing_ragged_any_horizon_vs_oracle(dev, B, H, wing_pk):
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from oracle import torch_port as tp
    dt = 0.05
    d = synthetic.wing_batch(B, H, dt, seed=B + H)
    st, loss, ga, gs = tp.rollout_fwd_bwd(
        tp.WingOracle(WMOD), tp.fixed_wing_mpc_loss, d["state0"], d["actions"],
        d["ref"], dt)
    dyn = FixedWingDynamics(modified_params=dict(WMOD))
    # (even batches also through the two-trajectories-per-lane kernel, with the
    # modified coefficient table read from the kernel arguments)
    for layout in ("aos", "soa") + (("two_per_lane",) if B % 2 == 0 else ()):
        wing_pk(1 if layout == "two_per_lane" else 0)
        if layout == "two_per_lane":
            layout = "soa"
        a = (d["state0"].to(dev), d["actions"].to(dev), d["ref"].to(dev))
        if layout == "soa":
            a = (soa_state(a[0]), soa_seq(a[1]), soa_seq(a[2]))
        res = F.wing_rollout_fwd_bwd(*a, dt, dyn.params, layout=layout,
                                     want_states=True)
        rs, rga, rgs = res["states"], res["grad_actions"], res["grad_state0"]
        if layout == "soa":
            rs, rga, rgs = aos_seq(rs), aos_seq(rga), rgs.t()
        assert rel_err(N(rs), st.numpy()) < TOL
        assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
        assert rel_err(N(rga), ga.numpy()) < 2 * TOL
        assert rel_err(N(rgs), gs.numpy()) < 2 * TOL


def test_cartpole_ragged_longer_horizon_vs_oracle(dev):
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import (
        CartpoleDynamics)
    from oracle import torch_port as tp
    B, H, dt = 70, 10, 0.02
    d = synthetic.cartpole_batch(B, H, seed=3)
    dyn_o = tp.CartpoleOracle()
    s0 = d["state0"].clone().requires_grad_(True)
    a = d["actions"].clone().requires_grad_(True)
    inter = tp.unroll(dyn_o, s0, a, dt)
    loss = tp.cartpole_loss_mpc(inter, tp.cartpole_reference(s0, H), a)
    loss.backward()
    dyn = CartpoleDynamics()
    res = F.cartpole_rollout_fwd_bwd(d["state0"].to(dev), d["actions"].to(dev),
                                     dt, dyn.params, want_states=True)
    assert rel_err(N(res["states"]), inter.detach().numpy()) < TOL
    assert abs(res["loss"].item() - loss.item()) / loss.item() < TOL
    assert rel_err(N(res["grad_actions"]), a.grad.numpy()) < TOL
    assert rel_err(N(res["grad_state0"]), s0.grad.numpy()) < TOL


def test_step_kernels_soa_layout_through_c_abi(dev):
    """The single-step entry points in the device-native SoA layout (the
    drop-in classes use AoS): same numbers as the golden AoS results."""
    import ctypes
    from apg_trajectory_tracking_amd import _capi, functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    g = load_golden("quad_step.npz")
    dyn = FlightmareDynamics()
    lib = _capi.lib()
    s = D(g["state"], dev).t().contiguous()
    a = D(g["action"], dev).t().contiguous()
    out = torch.empty_like(s)
    st = torch.cuda.current_stream().cuda_stream
    _capi.check(lib.apg_quad_step_fwd(s.data_ptr(), a.data_ptr(), 0.1,
                                      ctypes.byref(dyn.params), s.shape[1],
                                      _capi.LAYOUT_SOA, out.data_ptr(), st),
                "apg_quad_step_fwd")
    assert rel_err(N(out.t()), g["def_dt010_next"]) < 1e-6
    cot = D(g["cot"][0], dev).t().contiguous()
    gs, ga = torch.empty_like(s), torch.empty_like(a)
    _capi.check(lib.apg_quad_step_bwd(s.data_ptr(), a.data_ptr(), 0.1,
                                      ctypes.byref(dyn.params), s.shape[1],
                                      _capi.LAYOUT_SOA, cot.data_ptr(),
                                      gs.data_ptr(), ga.data_ptr(), st),
                "apg_quad_step_bwd")
    assert rel_err(N(gs.t()), g["def_dt010_gstate"][0]) < 1e-5
    assert rel_err(N(ga.t()), g["def_dt010_gaction"][0]) < 1e-5
    f = torch.empty(15, s.shape[1], device=dev)
    _capi.check(lib.apg_quad_features_fwd(s.data_ptr(), s.shape[1],
                                          _capi.LAYOUT_SOA, f.data_ptr(), st),
                "apg_quad_features_fwd")
    assert torch.allclose(f.t(), F.quad_features(s.t().contiguous()), atol=0, rtol=0)


def test_planes_gemm_mfma_vs_torch(dev):
    """apg_planes_gemm (v_mfma_f32_32x32x2_f32 reduction GEMM) against
    torch.matmul in float64, incl. segment strides, the ones column, ragged
    N and an asymmetric operand (catches row/column swaps)."""
    from apg_trajectory_tracking_amd import functional as F
    g = torch.Generator().manual_seed(0)
    for (M, S, P, J, bstride, N) in ((32, 1, 199, 183, 0, 6400),
                                     (4, 1, 16, 8, 0, 777),
                                     (64, 1, 140, 112, 0, 3000),
                                     (64, 1, 20, 15, 0, 515),
                                     (20, 8, 90, 27, 9, 1300)):
        A = torch.randn(M * S, N, generator=g).to(dev)
        Bp = (torch.randn(P, N, generator=g) + torch.arange(P)[:, None] * 0.01).to(dev)
        if bstride:
            offs = [t * 9 + c for c in range(9) for t in range(3)]
        else:
            offs = list(range(P - J, P))
        C = F.planes_gemm(A, M, S, Bp, F.make_bdesc(dev, offs, bstride),
                          with_ones=True)
        A64, B64 = A.double().cpu(), Bp.double().cpu()
        ref = torch.zeros(M, J + 1, dtype=torch.float64)
        for m in range(M):
            for s in range(S):
                a = A64[m * S + s]
                rows = torch.stack([B64[o + s * bstride] for o in offs])
                ref[m, :J] += rows @ a
                ref[m, J] += a.sum()
        assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
    # ragged N next to a plane that holds non-finite values and is
    # NOT part of the product - the chunk's columns beyond N are that plane's
    # first ones; both operands' tails are zeroed (0 x Inf would be NaN).
    # Register-streaming kernel (S = 1) and LDS-tile kernel (segments).
    for (M, S, P, J, bstride, N) in ((16, 1, 12, 4, 0, 777), (20, 8, 91, 27, 9, 1300)):
        A = torch.randn(M * S, N, generator=g).to(dev)
        Bp = torch.randn(P, N, generator=g).to(dev)
        Bp[P - 1] = float("inf")
        offs = ([t * 9 + c for c in range(9) for t in range(3)] if bstride
                else list(range(P - 1 - J, P - 1)))
        C = F.planes_gemm(A, M, S, Bp, F.make_bdesc(dev, offs, bstride))
        assert torch.isfinite(C).all()
        A64, B64 = A.double().cpu(), Bp.double().cpu()
        ref = torch.zeros(M, J + 1, dtype=torch.float64)
        for m in range(M):
            for s_ in range(S):
                rows = torch.stack([B64[o + s_ * bstride] for o in offs])
                ref[m, :J] += rows @ A64[m * S + s_]
                ref[m, J] += A64[m * S + s_].sum()
        assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5
    # per-column two-level segment strides (segment = (pos, step)), strided output
    H, Bn = 5, 203
    A = torch.randn(20 * 8 * H, Bn, generator=g).to(dev)        # [ch][pos][k] planes
    inr = torch.randn((H + 9) * 9 + H * 12, Bn, generator=g).to(dev)
    P0 = (H + 9) * 9                                             # "position" planes
    offs = [t * 9 + c for c in range(9) for t in range(3)] + [P0, P0 + 1, P0 + 2]
    s1, s2 = [9] * 27 + [0] * 3, [9] * 27 + [12] * 3
    out = torch.zeros(20, 40, device=dev)
    C = F.planes_gemm(A, 20, 8 * H, inr, F.make_bdesc(dev, offs, s1, s2), sdiv=H,
                      out=out[:, 5:])
    assert C.data_ptr() == out[:, 5:].data_ptr() and float(out[:, :5].abs().sum()) == 0
    A64, B64 = A.double().cpu().view(20, 8, H, Bn), inr.double().cpu()
    ref = torch.zeros(20, 31, dtype=torch.float64)
    for pos in range(8):
        for k in range(H):
            rows = torch.stack([B64[o + pos * a + k * b] for o, a, b in zip(offs, s1, s2)])
            ref[:, :30] += A64[:, pos, k] @ rows.t()
            ref[:, 30] += A64[:, pos, k].sum(1)
    assert rel_err(C.cpu().numpy(), ref.numpy()) < 1e-5


def test_fused_policy_argument_errors(dev):
    """Argument validation of the in-kernel-policy entry points: wrong network
    shape, wrong horizon, short reference trajectory, oversized GEMM operands
    -> ValueError (APG_ERR_ARG), nothing launched."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    dyn = FlightmareDynamics()
    d = synthetic.quad_polynomial_batch(8, 10, 0.1, seed=1, ref_length=20)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    with pytest.raises(ValueError):     # linear reference branch: not fusable
        F.quad_mlp_rollout_loss(Net(15, 10, 9, 4, conv=0).to(dev), s0, in_ref, ref,
                                0.1, dyn.params)
    with pytest.raises(ValueError):     # in_ref shorter than 2H
        F.quad_mlp_rollout_loss(Net(15, 10, 9, 4, conv=1).to(dev), s0,
                                in_ref[:, :12], ref, 0.1, dyn.params)
    h0 = torch.zeros(8, 8, device=dev)
    with pytest.raises(ValueError):     # horizon-5 LSTM
        F.quad_lstm_rollout_loss(LSTM_NEW(15, 5, 9, 4, conv=1).to(dev), s0, in_ref,
                                 ref, 0.1, dyn.params, h0, h0)
    with pytest.raises(ValueError):     # reference trajectory not longer than H
        F.quad_mlp_closed_loop(Net(15, 10, 9, 4, conv=1).to(dev),
                               torch.zeros(4, 10, 9, device=dev), 0.1, dyn.params)
    A = torch.zeros(65, 64, device=dev)
    with pytest.raises(ValueError):     # M > 64
        F.planes_gemm(A, 65, 1, A, F.make_bdesc(dev, range(8)))
    with pytest.raises(ValueError):     # M > 32 with more than 128 columns
        F.planes_gemm(A[:64], 64, 1, torch.zeros(200, 64, device=dev),
                      F.make_bdesc(dev, range(150)))
    # B = 0 is a no-op with a zero loss
    e = torch.zeros(0, 12, device=dev)
    loss, st, ac = F.quad_mlp_rollout_loss(
        Net(15, 10, 9, 4, conv=1).to(dev), e, torch.zeros(0, 20, 9, device=dev),
        torch.zeros(0, 20, 9, device=dev), 0.1, dyn.params)
    assert float(loss.detach()) == 0.0 and st.shape == (10, 12, 0)


def test_planes_gemm_grouped_vs_torch(dev):
    """apg_planes_gemm_grouped: several products of different shape in one
    launch pair, strided outputs, separate bias vectors, ragged N."""
    from apg_trajectory_tracking_amd import functional as F
    g = torch.Generator().manual_seed(3)
    N = 5000 + 37
    Bp = torch.randn(150, N, generator=g).to(dev)
    A1 = torch.randn(64, N, generator=g).to(dev)
    A2 = torch.randn(4, N, generator=g).to(dev)
    A3 = torch.randn(20 * 8, N, generator=g).to(dev)
    o1 = torch.zeros(64, 120, device=dev)
    b1 = torch.zeros(64, device=dev)
    o2 = torch.zeros(4, 9, device=dev)
    o3, b3 = torch.zeros(20, 27, device=dev), torch.zeros(20, device=dev)
    offs3 = [t * 9 + c for c in range(9) for t in range(3)]
    F.planes_gemm_grouped([
        dict(A=A1, M=64, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(10, 122)),
             out=o1[:, 4:], bias_out=b1),
        dict(A=A2, M=4, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(140, 148)), out=o2),
        dict(A=A3, M=20, S=8, Bp=Bp, bdesc=F.make_bdesc(dev, offs3, 9), out=o3,
             bias_out=b3)])
    B64 = Bp.double().cpu()
    r1 = A1.double().cpu() @ B64[10:122].t()
    assert rel_err(o1[:, 4:116].cpu().numpy(), r1.numpy()) < 1e-5
    assert float(o1[:, :4].abs().sum()) == 0 and float(o1[:, 116:].abs().sum()) == 0
    assert rel_err(b1.cpu().numpy(), A1.double().cpu().sum(1).numpy()) < 1e-5
    r2 = torch.cat((A2.double().cpu() @ B64[140:148].t(),
                    A2.double().cpu().sum(1, keepdim=True)), 1)
    assert rel_err(o2.cpu().numpy(), r2.numpy()) < 1e-5
    A3d = A3.double().cpu().view(20, 8, N)
    r3 = torch.zeros(20, 27, dtype=torch.float64)
    for s_ in range(8):
        rows = torch.stack([B64[o + 9 * s_] for o in offs3])
        r3 += A3d[:, s_] @ rows.t()
    assert rel_err(o3.cpu().numpy(), r3.numpy()) < 1e-5
    assert rel_err(b3.cpu().numpy(), A3d.sum((1, 2)).numpy()) < 1e-5


def N_(t):
    return t.detach().cpu().numpy()


def test_planes_gemm_multi_vs_torch(dev):
    """apg_planes_gemm_multi: the products of a recurrent training step, one
    launch each with its own tile shape (3-deep DMA ring where the LDS holds
    it, 2-deep for the 32 x 192 shape) and ONE second-stage launch; strided
    outputs, separate bias vectors, ragged N, two-level segment strides."""
    from apg_trajectory_tracking_amd import functional as F
    g = torch.Generator().manual_seed(5)
    N = 9000 + 21
    Bp = torch.randn(240, N, generator=g).to(dev)
    A1 = torch.randn(64, N, generator=g).to(dev)
    A2 = torch.randn(4, N, generator=g).to(dev)
    A3 = torch.randn(20 * 8 * 3, N, generator=g).to(dev)
    A4 = torch.randn(32, N, generator=g).to(dev)
    o1, b1 = torch.zeros(64, 120, device=dev), torch.zeros(64, device=dev)
    o2 = torch.zeros(4, 65, device=dev)
    o3, b3 = torch.zeros(20, 30, device=dev), torch.zeros(20, device=dev)
    o4, b4 = torch.zeros(32, 183, device=dev), torch.zeros(32, device=dev)
    o5 = torch.zeros(64, 16, device=dev)
    offs3 = [t * 9 + c for c in range(9) for t in range(3)] + [200, 201, 202]
    s1, s2 = [9] * 27 + [0] * 3, [9] * 27 + [12] * 3
    F.planes_gemm_multi([
        dict(A=A1, M=64, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(10, 122)),
             out=o1[:, 4:], bias_out=b1),
        dict(A=A2, M=4, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(140, 204)), out=o2),
        dict(A=A3, M=20, S=8 * 3, Bp=Bp, bdesc=F.make_bdesc(dev, offs3, s1, s2),
             sdiv=3, out=o3, bias_out=b3),
        dict(A=A4, M=32, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(57, 240)),
             out=o4, bias_out=b4),
        dict(A=A1, M=64, S=1, Bp=Bp, bdesc=F.make_bdesc(dev, range(225, 240)),
             out=o5)])
    B64 = Bp.double().cpu()
    A1d = A1.double().cpu()
    assert rel_err(o1[:, 4:116].cpu().numpy(), (A1d @ B64[10:122].t()).numpy()) < 1e-5
    assert float(o1[:, :4].abs().sum()) == 0 and float(o1[:, 116:].abs().sum()) == 0
    assert rel_err(b1.cpu().numpy(), A1d.sum(1).numpy()) < 1e-5
    r2 = torch.cat((A2.double().cpu() @ B64[140:204].t(),
                    A2.double().cpu().sum(1, keepdim=True)), 1)
    assert rel_err(o2.cpu().numpy(), r2.numpy()) < 1e-5
    A3d = A3.double().cpu().view(20, 8, 3, N)
    r3 = torch.zeros(20, 30, dtype=torch.float64)
    for pos in range(8):
        for k in range(3):
            rows = torch.stack([B64[o + pos * a + k * b]
                                for o, a, b in zip(offs3, s1, s2)])
            r3 += A3d[:, pos, k] @ rows.t()
    assert rel_err(o3.cpu().numpy(), r3.numpy()) < 1e-5
    assert rel_err(b3.cpu().numpy(), A3d.sum((1, 2, 3)).numpy()) < 1e-5
    A4d = A4.double().cpu()
    assert rel_err(o4.cpu().numpy(), (A4d @ B64[57:240].t()).numpy()) < 1e-5
    assert rel_err(b4.cpu().numpy(), A4d.sum(1).numpy()) < 1e-5
    r5 = torch.cat((A1d @ B64[225:240].t(), A1d.sum(1, keepdim=True)), 1)
    assert rel_err(o5.cpu().numpy(), r5.numpy()) < 1e-5
    # tiny reduction lengths: fewer tiles than workgroups, one ragged tile
    for n in (1, 63, 64, 65, 200):
        C = F.planes_gemm(A1[:, :n].contiguous(), 64, 1, Bp[:, :n].contiguous(),
                          F.make_bdesc(dev, range(100, 164)))
        r = torch.cat((A1d[:, :n] @ B64[100:164, :n].t(),
                       A1d[:, :n].sum(1, keepdim=True)), 1)
        assert rel_err(C.cpu().numpy(), r.numpy()) < 1e-5, n


def test_planes_gemm_operands_beyond_2gib(dev):
    """Operand offsets are unsigned 32-bit and B is addressed from the first
    plane a product uses: a 3.2 GB B tensor (planes of 1 Mi floats) works for
    products over its last planes as well as its first (2.7 GB span)."""
    from apg_trajectory_tracking_amd import functional as F
    N = 1 << 20
    Bp = torch.empty(760, N, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for lo in range(0, 760, 40):
        Bp[lo:lo + 40].normal_(generator=g)
    A = torch.randn(32, N, device=dev, generator=g)
    for lo, hi in ((728, 760), (0, 32), (620, 652)):
        C = F.planes_gemm(A, 32, 1, Bp, F.make_bdesc(dev, range(lo, hi)),
                          with_ones=False)
        ref = A.double() @ Bp[lo:hi].double().t()
        assert rel_err(N_(C), N_(ref)) < 1e-5, (lo, hi)
    # columns 0 and 650: the span (651 planes x 4 MiB = 2.7 GB) is what counts
    C = F.planes_gemm(A, 32, 1, Bp, F.make_bdesc(dev, [0, 650]), with_ones=False)
    ref = A.double() @ Bp[[0, 650]].double().t()
    assert rel_err(N_(C), N_(ref)) < 1e-5
    big = torch.empty(1100, N, device=dev)      # 4.6 GB
    big[1060:1092].copy_(Bp[:32])
    C = F.planes_gemm(A, 32, 1, big, F.make_bdesc(dev, range(1060, 1092)),
                      with_ones=False)           # a narrow product anywhere in it
    assert rel_err(N_(C), N_(A.double() @ Bp[:32].double().t())) < 1e-5
    with pytest.raises(ValueError):              # span beyond 32 bits
        F.planes_gemm(A, 32, 1, big, F.make_bdesc(dev, [0, 1090]), with_ones=False)
    with pytest.raises(ValueError):              # descriptor beyond the tensor
        F.planes_gemm(A, 32, 1, Bp, F.make_bdesc(dev, [759, 760]), with_ones=False)


def test_to_soa_matches_permute(dev):
    """apg_to_soa: tiled transpose at the boundary, incl. ragged sizes and a
    leading slice of longer rows read in place."""
    from apg_trajectory_tracking_amd import functional as F
    g = torch.Generator().manual_seed(5)
    for shape in ((1, 12), (77, 15), (300, 20, 9), (4097, 10, 9)):
        t = torch.randn(*shape, generator=g).to(dev)
        want = t.permute(*range(1, t.dim()), 0).contiguous()
        assert torch.equal(F.to_soa(t), want)
    t = torch.randn(130, 20, 9, generator=g).to(dev)
    assert torch.equal(F.to_soa(t[:, :10]), t[:, :10].permute(1, 2, 0).contiguous())
    out = torch.zeros(4, 9, 130, device=dev)
    F.to_soa(t[:, 3:7], out=out)           # not a leading slice: copied first
    assert torch.equal(out, t[:, 3:7].permute(1, 2, 0).contiguous())
    # wide rows (several 128-column chunks), gathered rows, and several tensors
    # of one batch in ONE launch (apg_to_soa_multi)
    wide = torch.randn(200, 261, 9, generator=g).to(dev)
    assert torch.equal(F.to_soa(wide), wide.permute(1, 2, 0).contiguous())
    idx = torch.randperm(4097, generator=g)[:1000].to(dev)
    a = torch.randn(4097, 15, generator=g).to(dev)
    b = torch.randn(4097, 20, 9, generator=g).to(dev)
    c = torch.randn(4097, 12, generator=g).to(dev)
    pre = torch.zeros(10, 9, 1000, device=dev)
    oa, ob, oc = F.to_soa_multi([(a, None), (b[:, :10], pre), (c, None)], index=idx)
    assert ob.data_ptr() == pre.data_ptr()
    assert torch.equal(oa, a[idx].t().contiguous())
    assert torch.equal(ob, b[idx][:, :10].permute(1, 2, 0).contiguous())
    assert torch.equal(oc, c[idx].t().contiguous())
    many = [(torch.randn(70, 3 + i, generator=g).to(dev), None) for i in range(8)]
    for (t_, _), o in zip(many, F.to_soa_multi(many)):     # more than one launch
        assert torch.equal(o, t_.t().contiguous())
    with pytest.raises(ValueError):
        F.to_soa_multi([(a, None), (c[:100], None)])
