"""Staged (mode 0) against trajectory-major (mode 1) weight products of the
concurrent step: every parameter gradient of both against float64 autograd at
several batch shapes, and the step's kernels timed at B = 65 536.
    python tools/ab_weight_products.py [time]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import _capi, functional as F, synthetic
from apg_trajectory_tracking_amd.dataset import state_preprocessing
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net
dev = torch.device("cuda:0")
H, DT = 10, 0.1
lib = _capi.lib()
dyn = FlightmareDynamics()


def case(B, seed):
    d = synthetic.quad_polynomial_batch(B, H, DT, seed=seed, ref_length=H)
    s0, in_ref, ref = (d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    return state_preprocessing(s0), s0, in_ref, ref


def fp64(net, inputs):
    from oracle import torch_port as tp          # the checker (tools only)
    n64 = Net(15, H, 9, 4 * H, conv=1).double()
    n64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    normed, s0, in_ref, ref = (t.double().cpu() for t in inputs)
    acts = torch.sigmoid(n64(normed, in_ref)).view(-1, H, 4)
    q = tp.QuadOracle()
    s, states = s0, []
    for k in range(H):
        s = q(s, acts[:, k], DT)
        states.append(s)
    loss = tp.quad_mpc_loss(torch.stack(states, 1), ref, acts)
    loss.backward()
    return {k: p.grad for k, p in n64.named_parameters() if p.grad is not None}


torch.manual_seed(0)
net = Net(15, H, 9, 4 * H, conv=1).to(dev)
for B in (1, 31, 77, 256, 257, 300, 4113, 8195):
    inputs = case(B, B)
    want = fp64(net, inputs)
    row = {"B": B}
    for mode in (0, 1):
        assert lib.apg_quad_mlp_set_weight_products(mode) == 0
        loss, g, _ = F.quad_concurrent_policy_grads(net, *inputs, DT, dyn.params)
        err = {k: float((g[k].double().cpu() - want[k]).abs().max() / want[k].abs().max())
               for k in want}
        row[f"worst_mode{mode}"] = max(err.values())
        row[f"worst_param_mode{mode}"] = max(err, key=err.get)
    print(row)
if len(sys.argv) > 1:
    B = 65536
    inputs = case(B, 1)
    for mode in (0, 1, 0, 1):
        lib.apg_quad_mlp_set_weight_products(mode)
        plan = F.QuadConcurrentStepPlan(net, F.quad_concurrent_prepare(*inputs), DT, dyn.params)
        for _ in range(10):
            plan.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            plan.launch()
        torch.cuda.synchronize()
        print({"mode": mode, "ms_per_step_no_update": (time.perf_counter() - t0) / 200 * 1e3})
lib.apg_quad_mlp_set_weight_products(0)
