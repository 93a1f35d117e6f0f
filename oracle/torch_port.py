"""ORACLE (test infrastructure, never the product path).

PyTorch-eager CPU restatement of the reference's APG hot path, differentiated
by torch.autograd exactly as the reference does.  It plays two roles:
  * checker for the HIP kernels in tests/ and __graft_entry__.smoke();
  * `cpu_baseline` leg of bench.py ("the reference's CPU PyTorch autograd
    path" of the north star; kind = "port").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import
this module.  It is pinned against the golden vectors generated from the real
reference (tests/golden/make_golden.py -> tests/test_oracle_golden.py).

Each function cites the reference lines it restates (paths relative to
/root/reference).  The arithmetic order follows the reference where it
matters for fp32 rounding (e.g. 0.5*dt*dt*acc + 0.5*dt*vel, the add-then-
subtract of the gyroscopic cross term); tensor assembly is this file's own.
"""
import math
import numpy as np
import torch

# ---- parameter tables (neural_control/dynamics/config_quad.json:1-29,
#      config_fixed_wing.json:1-42, config_cartpole.json:1-11) -------------
QUAD_CFG = dict(
    mass=0.723, arm_length=0.31, frame_inertia=[4.5, 4.5, 7.0],
    gravity=[0.0, 0.0, -9.81], kinv_ang_vel_tau=[16.6, 16.6, 5.0],
    rotational_drag=[0.0, 0.0, 0.0], translational_drag=[0.0, 0.0, 0.0],
)
WING_CFG = dict(
    mass=1.01, I_xx=0.04766, I_yy=0.05005, I_zz=0.09558, I_xz=-0.00105,
    rho=1.225, S=0.276, c=0.185, b=1.54, g=9.81,
    CL0=0.39, CL_alpha=4.5321, CL_q=0.318, CL_del_e=0.527,
    CD0=0.0765, CD_alpha=0.3346, CD_q=0.354, CD_del_e=0.004,
    CY0=0.0, CY_beta=-0.033, CY_p=-0.1, CY_r=0.039, CY_del_a=0.0,
    CY_del_r=0.225,
    Cl0=0.0, Cl_beta=-0.081, Cl_p=-0.529, Cl_r=0.159, Cl_del_a=-0.453,
    Cl_del_r=0.005,
    Cm0=0.02, Cm_alpha=-1.4037, Cm_q=-0.1324, Cm_del_e=-0.4236,
    Cn0=0.0, Cn_beta=0.189, Cn_p=-0.083, Cn_r=-0.948, Cn_del_a=-0.041,
    Cn_del_r=-0.077, epsilon=0.16534698176788384,
)
CARTPOLE_CFG = dict(
    masscart=1.0, masspole=0.1, length=0.5, max_force_mag=30.0,
)
ALPHA_BOUND = float(10 / 180 * math.pi)  # fixed_wing_dynamics.py:10


def _rows(*rows):
    """rows: 3 tuples of 3 [B] tensors -> [B,3,3]."""
    return torch.stack([torch.stack(r, dim=1) for r in rows], dim=1)


# =========================== quadrotor ====================================
class QuadOracle:
    """FlightmareDynamics, quad_dynamics_flightmare.py:128-216 with
    quad_dynamics_base.py:11-127."""

    def __init__(self, modified_params=None, dtype=torch.float32):
        cfg = dict(QUAD_CFG)
        cfg.update(modified_params or {})
        self.cfg = cfg
        self.dtype = dtype
        self.mass = cfg["mass"]
        f64 = torch.float64
        inertia = (
            self.mass / 12.0 * cfg["arm_length"]**2 *
            torch.tensor(cfg["frame_inertia"], dtype=f64)
        )                                   # quad_dynamics_base.py:33-36
        self.J = torch.diag(inertia.to(torch.float32)).to(dtype)
        self.K = torch.diag(
            torch.tensor(cfg["kinv_ang_vel_tau"], dtype=f64).to(torch.float32)
        ).to(dtype)
        self.gravity = torch.tensor(cfg["gravity"], dtype=dtype)
        self.t_drag = torch.tensor(
            cfg["translational_drag"], dtype=torch.float32).to(dtype)
        self.r_drag = torch.tensor(
            cfg["rotational_drag"], dtype=torch.float32).to(dtype)

    @staticmethod
    def world_to_body(att):
        """quad_dynamics_base.py:59-94."""
        r, p, y = att[:, 0], att[:, 1], att[:, 2]
        cy, sy = torch.cos(y), torch.sin(y)
        cp, sp = torch.cos(p), torch.sin(p)
        cr, sr = torch.cos(r), torch.sin(r)
        return _rows(
            (cy * cp, sy * cp, -sp),
            (cy * sp * sr - cr * sy, cr * cy + sr * sy * sp, cp * sr),
            (cy * sp * cr + sr * sy, cr * sy * sp - cy * sr, cr * cp),
        )

    @staticmethod
    def euler_matrix(att):
        """quad_dynamics_base.py:96-118."""
        r, p = att[:, 0], att[:, 1]
        cp, sp = torch.cos(p), torch.sin(p)
        cr, sr = torch.cos(r), torch.sin(r)
        one, zero = torch.ones_like(sp), torch.zeros_like(sp)
        return _rows((one, zero, -sp), (zero, cr, cp * sr), (zero, -sr, cp * cr))

    def __call__(self, state, action, dt):
        state = state.to(self.dtype)
        action = action.to(self.dtype)
        pos, att = state[:, 0:3], state[:, 3:6]
        vel, omega = state[:, 6:9], state[:, 9:12]
        # :139-140 action scaling
        thrust = action[:, 0] * 15 - 7.5 + 9.81
        rates = action[:, 1:] - .5
        # :146-149 gyroscopic term
        j_omega = (self.J @ omega.unsqueeze(2))[:, :, 0]
        cross = torch.cross(omega, j_omega, dim=1)
        # run_flight_control :95-117
        force = self.mass * thrust
        k_err = self.K @ (rates - omega).unsqueeze(2)
        tau = (self.J @ k_err)[:, :, 0] + cross + self.r_drag
        # linear_dynamics :74-93 (force vector [0,0,F], body->world)
        zeros = torch.zeros_like(force)
        f_body = torch.stack((zeros, zeros, force), dim=1).unsqueeze(2)
        b2w = self.world_to_body(att).transpose(1, 2)
        acc = (1 / self.mass * (b2w @ f_body))[:, :, 0] + self.gravity \
            + self.t_drag
        # :172-175 (sic: 0.5*dt*velocity)
        new_pos = pos + 0.5 * dt * dt * acc + 0.5 * dt * vel
        new_vel = vel + dt * acc
        # :178-183
        j_inv = torch.inverse(self.J)
        ang_acc = (j_inv @ (tau - cross).unsqueeze(2))[:, :, 0]
        new_omega = omega + dt * ang_acc
        # :210 attitude from the OLD angular velocity
        e_rate = (self.euler_matrix(att) @ omega.unsqueeze(2))[:, :, 0]
        new_att = att + dt * e_rate
        return torch.cat((new_pos, new_att, new_vel, new_omega), dim=1)


class LearntQuadOracle:
    """LearntDynamics.forward, quad_dynamics_trained.py:10-69: the 4 x 4 action
    transform (:62-64), the Flightmare step with the kinv / inertia of
    CONSTRUCTION time (:48-50: torch.diag copies the step keeps using), the
    residual network 16 -> 64 -> 12 on [state, transformed action] (:52-58)
    added to the new state (:66-69).  `weights`: the module's state_dict
    (linear_at, linear_state_1/2.weight/.bias; the physical parameters in it do
    not enter the step)."""

    def __init__(self, weights, initial_params=None, dtype=torch.float32):
        self.base = QuadOracle(modified_params=initial_params, dtype=dtype)
        t = lambda k: torch.as_tensor(weights[k]).to(dtype)
        self.A = t("linear_at")
        self.w1, self.b1 = t("linear_state_1.weight"), t("linear_state_1.bias")
        self.w2, self.b2 = t("linear_state_2.weight"), t("linear_state_2.bias")
        self.dtype = dtype

    def __call__(self, state, action, dt):
        state, action = state.to(self.dtype), action.to(self.dtype)
        at = (self.A @ action.unsqueeze(2))[:, :, 0]
        new_state = self.base(state, at, dt)
        hidden = torch.relu(torch.cat((state, at), dim=1) @ self.w1.t() + self.b1)
        return new_state + hidden @ self.w2.t() + self.b2


def quad_mpc_loss(states, ref_states, action_seq):
    """neural_control/drone_loss.py:12-39."""
    prior = torch.tensor([.5, .5, .5], dtype=states.dtype)
    l_pos = torch.sum((states[:, :, :3] - ref_states[:, :, :3])**2)
    l_vel = torch.sum((states[:, :, 6:9] - ref_states[:, :, 6:9])**2)
    l_av = torch.sum(states[:, :, 9:12]**2)
    l_thrust = torch.sum((action_seq[:, :, 0] - .5)**2)
    l_rates = torch.sum((action_seq[:, :, 1:] - prior)**2)
    return 10 * l_pos + 1 * l_vel + 0.1 * l_av + 0.1 * l_rates + 5 * l_thrust


def quad_state_features(state):
    """state_preprocessing, neural_control/dataset.py:207-220 (+146-153)."""
    vel = state[:, 6:9]
    w2b = QuadOracle.world_to_body(state[:, 3:6])
    vel_body = (w2b @ vel.unsqueeze(2))[:, :, 0]
    rot = torch.reshape(w2b[:, :, :2], (-1, 6))
    return torch.cat((vel, rot, vel_body, state[:, 9:12]), dim=1)


# =========================== fixed wing ===================================
class WingOracle:
    """FixedWingDynamics, fixed_wing_dynamics.py:18-267."""

    def __init__(self, modified_params=None, dtype=torch.float32):
        cfg = dict(WING_CFG)
        cfg.update(modified_params or {})
        self.cfg = cfg
        self.dtype = dtype
        self.I = torch.tensor(
            [[cfg["I_xx"], 0, -cfg["I_xz"]], [0, cfg["I_yy"], 0],
             [-cfg["I_xz"], 0, cfg["I_zz"]]], dtype=torch.float32
        ).to(dtype)

    @staticmethod
    def r_body_wind(alpha, beta):
        """:48-63."""
        sa, sb = torch.sin(alpha), torch.sin(beta)
        ca, cb = torch.cos(alpha), torch.cos(beta)
        zero = torch.zeros_like(sa)
        return _rows((ca * cb, -ca * sb, -sa), (sb, cb, zero),
                     (sa * cb, -sa * sb, ca))

    @staticmethod
    def r_inertial_body(phi, theta, psi):
        """:65-93 (returns the transpose of the row-assembled matrix)."""
        sph, cph = torch.sin(phi), torch.cos(phi)
        sth, cth = torch.sin(theta), torch.cos(theta)
        sps, cps = torch.sin(psi), torch.cos(psi)
        m = _rows(
            (cth * cps, cth * sps, -sth),
            (-cph * sps + sph * sth * cps, cph * cps + sph * sth * sps,
             sph * cth),
            (sph * sps + cph * sth * cps, -sph * cps + cph * sth * sps,
             cph * cth),
        )
        return m.transpose(1, 2)

    def __call__(self, state, action, dt):
        c = self.cfg
        state = state.to(self.dtype)
        action = action.to(self.dtype)
        vel = state[:, 3:6]
        u, v, w = state[:, 3], state[:, 4], state[:, 5]
        phi, theta, psi = state[:, 6], state[:, 7], state[:, 8]
        omega = state[:, 9:12]
        p, q, r = state[:, 9], state[:, 10], state[:, 11]
        # normalize_action :41-46
        T = action[:, 0] * 7
        del_e = math.pi * (action[:, 1] * 40 - 20) / 180
        del_a = math.pi * (action[:, 2] * 5 - 2.5) / 180
        del_r = math.pi * (action[:, 3] * 40 - 20) / 180
        g_m = c["g"] * c["mass"]
        # :130-134
        V = torch.sqrt(u**2 + v**2 + w**2)
        alpha = torch.clamp(torch.arctan(w / u), -ALPHA_BOUND, ALPHA_BOUND)
        beta = torch.clamp(torch.arctan(v / V), -ALPHA_BOUND, ALPHA_BOUND)
        # :139-164 coefficients
        CL = c["CL0"] + c["CL_alpha"] * alpha + c["CL_q"] * c["c"] / (
            2 * V) * q + c["CL_del_e"] * del_e
        CD = c["CD0"] + c["CD_alpha"] * alpha + c["CD_q"] * c["c"] / (
            2 * V) * q + c["CD_del_e"] * del_e
        CY = c["CY0"] + c["CY_beta"] * beta + c["CY_p"] * c["b"] / (
            2 * V) * p + c["CY_r"] * c["b"] / (2 * V) * r + c[
                "CY_del_a"] * del_a + c["CY_del_r"] * del_r
        Cl = c["Cl0"] + c["Cl_beta"] * beta + c["Cl_p"] * c["b"] / (
            2 * V) * p + c["Cl_r"] * c["b"] / (2 * V) * r + c[
                "Cl_del_a"] * del_a + c["Cl_del_r"] * del_r
        Cm = c["Cm0"] + c["Cm_alpha"] * alpha + c["Cm_q"] * c["c"] / (
            2 * V) * q + c["Cm_del_e"] * del_e
        Cn = c["Cn0"] + c["Cn_beta"] * beta + c["Cn_p"] * c["b"] / (
            2 * V) * p + c["Cn_r"] * c["b"] / (2 * V) * r + c[
                "Cn_del_a"] * del_a + c["Cn_del_r"] * del_r
        # :167-175 forces / moments (all three moments scale with chord c)
        L = 1 / 2 * c["rho"] * V**2 * c["S"] * CL
        D = 1 / 2 * c["rho"] * V**2 * c["S"] * CD
        Y = 1 / 2 * c["rho"] * V**2 * c["S"] * CY
        l = 1 / 2 * c["rho"] * V**2 * c["S"] * c["c"] * Cl
        m = 1 / 2 * c["rho"] * V**2 * c["S"] * c["c"] * Cm
        n = 1 / 2 * c["rho"] * V**2 * c["S"] * c["c"] * Cn
        # :185-204 body forces
        zero = torch.zeros_like(theta)
        eps = zero + c["epsilon"]
        aero = torch.stack((-D, Y, -L), 1).unsqueeze(2)
        prop = torch.stack((T * torch.cos(eps), torch.zeros_like(T),
                            T * torch.sin(eps)), 1)
        b2i = self.r_inertial_body(phi, theta, zero).transpose(1, 2)
        g_m = g_m.item() if torch.is_tensor(g_m) else g_m      # a detached copy, :197
        grav = torch.tensor([[0.0], [0.0], [g_m]], dtype=self.dtype)
        f_xyz = self.r_body_wind(alpha, beta) @ aero + b2i @ grav \
            + prop.unsqueeze(2)
        # :213-221
        pos_dot = self.r_inertial_body(phi, theta, psi) @ vel.unsqueeze(2)
        uvw_dot = (1 / c["mass"]) * f_xyz[:, :, 0] - torch.cross(
            omega, vel, dim=1)
        # :225-245 euler rates
        one = torch.ones_like(phi)
        e_mat = _rows(
            (one, torch.sin(phi) * torch.tan(theta),
             torch.cos(phi) * torch.tan(theta)),
            (zero, torch.cos(phi), -torch.sin(phi)),
            (zero, torch.sin(phi) / torch.cos(theta),
             torch.cos(phi) / torch.cos(theta)),
        )
        om = omega.unsqueeze(2)
        eul_dot = e_mat @ om
        # :250-255
        rhs = torch.stack((l, m, n), 1) - torch.cross(
            omega, (self.I @ om)[:, :, 0], dim=1)
        omega_dot = torch.inverse(self.I) @ rhs.unsqueeze(2)
        state_dot = torch.cat(
            (pos_dot[:, :, 0], uvw_dot, eul_dot[:, :, 0], omega_dot[:, :, 0]),
            dim=1)
        return state + dt * state_dot


class LearntWingOracle:
    """LearntFixedWingDynamics, fixed_wing_dynamics.py:270-326: the step above
    with every config entry ([1] tensors, `cfg.<name>`) and the 3x3 inertia
    matrix `I` as leaves that require grad, plus the residual MLP
    (16 -> 64 relu -> 12) on [state, action] added to the simulated next
    state.  `weights`: name -> array under the reference's state_dict names.
    The weight g * mass enters WingOracle.__call__ through torch.tensor(...),
    i.e. detached, as in the reference (:197): `cfg.g` gets no gradient."""

    def __init__(self, weights, dtype=torch.float64):
        self.dtype = dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(True)
                  for k, v in weights.items()}
        self.phys = WingOracle(dtype=dtype)
        self.phys.I = self.p["I"]
        self.phys.cfg = {k[len("cfg."):]: v for k, v in self.p.items()
                         if k.startswith("cfg.")}

    def parameters(self):
        return self.p

    def __call__(self, state, action, dt):
        state, action = state.to(self.dtype), action.to(self.dtype)
        sa = torch.cat((state, action), dim=1)
        hidden = torch.relu(sa @ self.p["linear_state_1.weight"].t()
                            + self.p["linear_state_1.bias"])
        added = hidden @ self.p["linear_state_2.weight"].t() \
            + self.p["linear_state_2.bias"]
        return self.phys(state, action, dt) + added


def fixed_wing_mpc_loss(states, linear_reference, action):
    """neural_control/drone_loss.py:72-82."""
    prior = torch.tensor([.5, .5, .5], dtype=states.dtype)
    a_loss = torch.sum((action[:, :, 1:] - prior)**2)
    p_loss = torch.sum((states[:, :, :3] - linear_reference)**2)
    return 10 * p_loss + 0.1 * a_loss


def wing_linear_reference(state0, target, horizon, dt):
    """WingDataset._compute_target_pos / prepare_data,
    neural_control/dataset.py:309-347."""
    rel = target - state0[:, :3]
    nvec = (rel.t() / torch.sqrt(torch.sum(rel**2, dim=1))).t()
    out = torch.zeros(state0.shape[0], horizon, 3, dtype=state0.dtype)
    for i in range(horizon):
        out[:, i] = state0[:, :3] + nvec * (12 * dt) * (i + 1)
    return out


# =========================== cartpole =====================================
class CartpoleOracle:
    """CartpoleDynamics, cartpole_dynamics.py:23-119."""

    def __init__(self, modified_params=None, dtype=torch.float32):
        cfg = dict(CARTPOLE_CFG)
        cfg.update(modified_params or {})
        cfg["friction"] = .5
        cfg["total_mass"] = cfg["masspole"] + cfg["masscart"]
        cfg["polemass_length"] = cfg["masspole"] * cfg["length"]
        self.cfg = cfg
        self.dtype = dtype

    def __call__(self, state, action, dt):
        c = self.cfg
        g = 9.81
        state = state.to(self.dtype)
        force = action[..., 0].to(self.dtype) * c["max_force_mag"] * 0.5
        x_dot, theta, th_dot = state[..., 1], state[..., 2], state[..., 3]
        s, co = torch.sin(theta), torch.cos(theta)
        xacc = (
            -2 * c["polemass_length"] * (th_dot**2) * s
            + 3 * c["masspole"] * g * s * co + 4 * force
            - 4 * c["friction"] * x_dot
        ) / (4 * c["total_mass"] - 3 * c["masspole"] * co**2)
        thacc = (
            -3 * c["polemass_length"] * (th_dot**2) * s * co
            + 6 * c["total_mass"] * g * s
            + 6 * (force - c["friction"] * x_dot) * co
        ) / (4 * c["length"] * c["total_mass"]
             - 3 * c["polemass_length"] * co**2)
        sd, cd = torch.sin(th_dot * dt), torch.cos(th_dot * dt)
        new_sin = s * cd + co * sd
        new_cos = co * cd - s * sd
        return torch.stack(
            [state[..., 0] + x_dot * dt, x_dot + xacc * dt,
             torch.atan2(new_sin, new_cos), th_dot + thacc * dt], dim=-1)


def cartpole_reference(state0, horizon):
    """make_reference, scripts/train_cartpole.py:103-110."""
    ref = torch.zeros(state0.shape[0], horizon, state0.shape[1],
                      dtype=state0.dtype)
    for k in range(horizon - 1):
        ref[:, k] = state0 * (1 - 1 / (horizon - 1) * k)
    return ref


def cartpole_loss_mpc(states, ref_states, actions):
    """neural_control/drone_loss.py:136-145."""
    w = torch.tensor([0, 3, 10, 1], dtype=states.dtype)
    return torch.sum((states - ref_states)**2 * w) \
        + 0.01 * torch.sum(actions**2)


# =========================== unrolls ======================================
def unroll(dyn, state0, action_seq, dt):
    """The k-step loop of scripts/train_drone.py:181-190 (and the wing /
    cartpole twins): returns intermediate states [B,H,S]."""
    B, H = action_seq.shape[:2]
    inter = torch.zeros(B, H, state0.shape[1], dtype=state0.dtype)
    cur = state0
    for k in range(H):
        cur = dyn(cur, action_seq[:, k], dt)
        inter[:, k] = cur
    return inter


def rollout_fwd_bwd(dyn, loss_fn, state0, actions, ref, dt):
    """fwd+bwd through dynamics+loss with leaf state0/actions.
    Returns (states, loss, dL/dactions, dL/dstate0)."""
    s0 = state0.detach().clone().requires_grad_(True)
    a = actions.detach().clone().requires_grad_(True)
    inter = unroll(dyn, s0, a, dt)
    loss = loss_fn(inter, ref, a)
    loss.backward()
    return inter.detach(), loss.detach(), a.grad, s0.grad


def quad_recurrent_unroll(net, dyn, state0, in_ref, ref, horizon, dt,
                          legacy_inplace_ref=False):
    """scripts/train_drone.py:113-165 with the window copied before the
    relative-position subtraction (pinned semantics).
    legacy_inplace_ref: the loop AS SHIPPED (:138-142) - the window is a view and
    the subtraction writes through it, so a reference row is shifted by the
    current position of every step whose window holds it; forward only (call
    under torch.no_grad(): autograd refuses the in-place write).  The caller's
    in_ref is left alone (the reference destroys its batch)."""
    B = state0.shape[0]
    inter = torch.zeros(B, horizon, 12, dtype=state0.dtype)
    acts = torch.zeros(B, horizon, 4, dtype=state0.dtype)
    cur = state0
    if legacy_inplace_ref:
        in_ref = in_ref.clone()
    for k in range(horizon):
        rel = in_ref[:, k:k + horizon] if legacy_inplace_ref else in_ref[:, k:k + horizon].clone()
        rel[:, :, :3] = rel[:, :, :3] - cur[:, None, :3]
        a = torch.sigmoid(net(quad_state_features(cur), rel))
        acts[:, k] = a
        cur = dyn(cur, a, dt)
        inter[:, k] = cur
    loss = quad_mpc_loss(inter, ref[:, :horizon], acts)
    return inter, acts, loss


def quad_closed_loop(net, dyn, traj, dt, horizon, max_steps, thresh_div,
                     thresh_stable, test_time, hidden=None):
    """Batched restatement of `QuadEvaluator.follow_trajectory("rand")`
    (scripts/evaluate_drone.py:81-194) with
      Random.get_ref_traj / project_on_ref / get_current_full_state
        (neural_control/trajectory/random_traj.py:60-92; +3 on z :34),
      NetworkWrapper.predict_actions (controllers/network_wrapper.py:42-72),
      QuadDataset.prepare_data (neural_control/dataset.py:155-204),
      QuadRotorEnvBase.step / get_is_stable / zero_reset
        (neural_control/environments/drone_env.py:59-117,129-142).
    traj [B, L, 9] = (position, euler, velocity) rows; every trajectory runs
    its own loop (break / reset are per trajectory).  Returns dict with
    drone [B, T+1, 12], ref [B, T, 3], div [B, T], actions [B, T, 4] and
    steps [B] (iterations executed; entries beyond are left at zero)."""
    B, L, _ = traj.shape
    H = horizon
    ref = traj.clone()
    ref[:, :, 2] += 3.0
    T = min(max_steps, L + 1)
    state = torch.zeros(B, 12, dtype=traj.dtype)
    state[:, :3] = ref[:, 0, :3]
    cur = 0
    alive = torch.ones(B, dtype=torch.bool)
    out = dict(drone=torch.zeros(B, T + 1, 12), ref=torch.zeros(B, T, 3),
               div=torch.zeros(B, T), actions=torch.zeros(B, T, 4),
               steps=torch.zeros(B, dtype=torch.long))
    out["drone"][:, 0] = state
    last = ref[:, -1, :3]
    with torch.no_grad():
        for i in range(T):
            if cur >= L - H:
                left = ref[:, cur:]
                pad = torch.zeros(B, H - (L - cur), 9, dtype=traj.dtype)
                pad[:, :, :3] = last[:, None]
                window = torch.cat((left, pad), 1)
            else:
                window = ref[:, cur + 1:cur + H + 1]
                cur += 1
            rel = torch.cat((window[:, :, :3] - state[:, None, :3],
                             window[:, :, 6:9],
                             window[:, :, 6:9] - state[:, None, 6:9]), 2)
            raw = net(quad_state_features(state), rel)
            action = torch.sigmoid(raw)[:, :4].clamp(0.0, 1.0)
            new = dyn(state, action, dt)
            on_line = ref[:, cur, :3]
            div = torch.linalg.norm(on_line - new[:, :3], dim=1)
            stable = (new[:, 3:5].abs() < thresh_stable).all(1)
            rec = alive.clone()
            out["drone"][rec, i + 1] = new[rec]
            out["ref"][rec, i] = on_line[rec]
            out["div"][rec, i] = div[rec]
            out["actions"][rec, i] = action[rec]
            out["steps"][rec] = i + 1
            failed = (div > thresh_div) | ~stable
            if test_time:
                alive = alive & ~failed
            reset = torch.cat((ref[:, cur], torch.zeros(B, 3, dtype=traj.dtype)), 1)
            state = torch.where((failed & (not test_time))[:, None], reset, new)
            if i >= L:
                break
    return out


def wing_closed_loop(net, dyn, targets, dt, mean, std, data_dt, data_horizon,
                     max_steps, thresh_div, thresh_stable, test_time,
                     state0=None, des_speed=11.5):
    """Batched restatement of `FixedWingEvaluator.fly_to_point`
    (scripts/evaluate_fixed_wing.py:45-131) with
      WingDataset.prepare_data / _compute_target_pos
        (neural_control/dataset.py:309-350),
      FixedWingNetWrapper.predict_actions (controllers/network_wrapper.py:81-98),
      SimpleWingEnv.zero_reset / step (environments/wing_env.py:26-28,44-57),
      project_to_line (trajectory/q_funcs.py:6-18, evaluated in float64 as the
      reference's numpy arithmetic is).
    targets [B, n, 3]; every flight runs its own loop (target switch, break and
    reset are per flight).  As in the reference the policy keeps seeing the
    last SIMULATED state after a reset (:124 resets the environment only).
    Returns dict: traj [B, T, 16] (state after the step + action), div_linear
    [B, T], div_pass / div_fail [B, T] (what the step appended to div_target
    when a target was passed / on divergence, -1 otherwise), seen [B, T, 15]
    (state the policy saw + its target), steps [B]."""
    B, n_t, _ = targets.shape
    T = max_steps
    f64 = torch.float64
    mean = torch.as_tensor(mean, dtype=torch.float32)
    std = torch.as_tensor(std, dtype=torch.float32)
    if state0 is None:
        env = torch.zeros(B, 12)
        env[:, 3] = 11.5
    else:
        env = state0.clone().float()
    obs = env.clone()
    line = env[:, :3].to(f64)
    prev = env[:, :3].to(f64)
    ti = torch.zeros(B, dtype=torch.long)
    alive = torch.ones(B, dtype=torch.bool)
    out = dict(traj=torch.zeros(B, T, 16), div_linear=torch.zeros(B, T, dtype=f64),
               div_pass=-torch.ones(B, T, dtype=f64),
               div_fail=-torch.ones(B, T, dtype=f64), seen=torch.zeros(B, T, 15),
               steps=torch.zeros(B, dtype=torch.long))
    tg64 = targets.to(f64)

    def project(a, b, p):
        ab = b - a
        n2 = (ab * ab).sum(1, keepdim=True)
        d = (ab * (p - a)).sum(1, keepdim=True)
        return torch.where(n2 > 0, a + ab * d / n2.clamp(min=1e-300), a)

    rows = torch.arange(B)
    vec_len = 12 * data_dt
    with torch.no_grad():
        for k in range(T):
            tg = tg64[rows, ti]
            tg32 = tg.float()
            out["seen"][alive, k] = torch.cat((obs, tg32), 1)[alive]
            normed = ((obs - mean) / std)[:, 3:]
            rel = tg32 - obs[:, :3]
            nvec = (rel.t() / torch.sqrt(torch.sum(rel**2, dim=1))).t()
            last = obs[:, :3] + nvec * vec_len * data_horizon
            in_ref = last - obs[:, :3]
            action = torch.sigmoid(net(normed, in_ref))[:, :4]
            new = dyn(env, action, dt)
            obs = new
            stable = (new[:, 6:8].abs() < thresh_stable).all(1)
            pos = new[:, :3].to(f64)
            on_line = project(line, tg, pos)
            div = torch.linalg.norm(on_line - pos, dim=1)
            rec = alive.clone()
            out["traj"][rec, k] = torch.cat((new, action), 1)[rec]
            out["div_linear"][rec, k] = div[rec]
            out["steps"][rec] = k + 1
            passed = pos[:, 0] > tg[:, 0]
            d_pass = torch.linalg.norm(project(prev, pos, tg) - tg, dim=1)
            last_target = ti >= n_t - 1
            done = passed & last_target
            advance = passed & ~last_target
            failed = ~done & (~stable | (div > thresh_div))
            d_fail = torch.full((B,), float(thresh_div), dtype=f64)
            if test_time:
                d_fail = torch.linalg.norm(pos - tg, dim=1)
            out["div_pass"][rec & passed, k] = d_pass[rec & passed]
            out["div_fail"][rec & failed, k] = d_fail[rec & failed]
            # continue on the line towards the (old) target at des_speed
            v = tg - on_line
            reset = torch.zeros(B, 12, dtype=f64)
            reset[:, :3] = on_line
            reset[:, 3:6] = v / torch.linalg.norm(v, dim=1, keepdim=True) * des_speed
            do_reset = failed & (not test_time)
            env = torch.where(do_reset[:, None], reset.float(), new)
            line = torch.where(advance[:, None], pos, line)
            ti = ti + advance.long()
            prev = pos
            if test_time:
                done = done | failed
            alive = alive & ~done
            if not alive.any():
                break
    return out
