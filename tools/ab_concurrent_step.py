"""In-sweep weight gradients (round 4) against the plane + product path of the
concurrent training step: gradients of both against float64 autograd at small
batches (ragged ones included), kernel-level timing at B = 65 536.
    python tools/ab_concurrent_step.py"""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic  # noqa: E402
from apg_trajectory_tracking_amd.dataset import state_preprocessing  # noqa: E402
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (  # noqa: E402
    FlightmareDynamics)
from apg_trajectory_tracking_amd.models.hutter_model import Net  # noqa: E402
from oracle import torch_port as tp  # noqa: E402

dev = torch.device("cuda:0")
H, dt = 10, 0.1
dyn = FlightmareDynamics()
torch.manual_seed(8)
net = Net(15, H, 9, 4 * H, conv=1)
gnet = copy.deepcopy(net).to(dev)


def run(B, seed, in_sweep):
    d = synthetic.quad_polynomial_batch(B, H, dt, seed=seed)
    s0 = d["state0"].to(dev)
    with torch.no_grad():
        normed = state_preprocessing(s0)
    F.CONCURRENT_IN_SWEEP = in_sweep
    loss, grads, _ = F.quad_concurrent_policy_grads(
        gnet, normed, s0, d["in_ref"].to(dev), d["ref"].to(dev), dt, dyn.params)
    return d, loss.item(), {k: v.double().cpu().numpy() for k, v in grads.items()}


for B in (1, 31, 64, 257, 300, 1000, 4096 + 17):
    d, l1, g1 = run(B, B, True)
    _, l0, g0 = run(B, B, False)
    net64 = copy.deepcopy(net).double()
    s64 = d["state0"].double()
    acts = torch.sigmoid(net64(tp.quad_state_features(s64), d["in_ref"].double())).reshape(-1, H, 4)
    loss64 = tp.quad_mpc_loss(tp.unroll(tp.QuadOracle(dtype=torch.float64), s64, acts, dt),
                              d["ref"].double(), acts)
    loss64.backward()
    worst = {}
    for k, p in net64.named_parameters():
        if p.grad is None:
            continue
        w = p.grad.numpy()
        sc = max(np.abs(w).max(), 1e-30)
        worst[k] = (float("%.2g" % (np.abs(g1[k] - w).max() / sc)),
                    float("%.2g" % (np.abs(g0[k] - w).max() / sc)))
    print(json.dumps({"B": B, "loss_rel": [abs(l1 - loss64.item()) / loss64.item(),
                                           abs(l0 - loss64.item()) / loss64.item()],
                      "max_err_in_sweep": max(v[0] for v in worst.values()),
                      "max_err_planes": max(v[1] for v in worst.values()),
                      "per_param(in_sweep, planes)": worst}))

# timing at the bench size
B = 65536
d = synthetic.quad_polynomial_batch(B, H, dt, seed=0)
s0 = d["state0"].to(dev)
with torch.no_grad():
    normed = state_preprocessing(s0)
in_ref, ref = d["in_ref"].to(dev), d["ref"].to(dev)
for in_sweep in (True, False, True):
    F.CONCURRENT_IN_SWEEP = in_sweep
    F._STATIC_PLANES.entries.clear()
    step = lambda: F.quad_concurrent_policy_grads(gnet, normed, s0, in_ref, ref, dt,
                                                  dyn.params, static_inputs=True)
    for _ in range(5):
        step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"B": B, "in_sweep": in_sweep,
                      "us_per_step_without_sgd": e0.elapsed_time(e1) / 200 * 1e3}))
