"""The concurrent step with its batch named by rows (forward kernel reads the
data set through the index) against the gathered step (apg_to_soa_multi +
planes).  python tools/ab_rows.py [rows|gather|both] [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from apg_trajectory_tracking_amd import functional as F, synthetic          # noqa: E402
from apg_trajectory_tracking_amd.dataset import state_preprocessing          # noqa: E402
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (  # noqa: E402
    FlightmareDynamics)
from apg_trajectory_tracking_amd.models.hutter_model import Net              # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
B, N = 65536, 32 * 65536
d = synthetic.quad_polynomial_batch(N, 10, 0.1, seed=5, ref_length=10)
st, inr, rf = (d[k].to(dev).contiguous() for k in ("state0", "in_ref", "ref"))
normed = state_preprocessing(st).contiguous()
params = FlightmareDynamics().params
perm = torch.randperm(N, device=dev)
batches = [perm[i * B:(i + 1) * B] for i in range(N // B)]
for mode in (("rows", "gather") if which == "both" else (which,)):
    torch.manual_seed(1)
    net = Net(15, 10, 9, 40, conv=1).to(dev)
    bufs = {n: torch.zeros_like(p) for n, p in net.named_parameters() if n in F._MLP_PARAMS}
    update = (1e-9, 0.9, bufs)
    if mode == "rows":
        plan = F.QuadConcurrentStepPlan(net, None, 0.1, params, update=update,
                                        rows=(normed, st, inr, rf, B))
        step = lambda i: plan.launch(index=batches[i % len(batches)])
    else:
        prepared = F.quad_concurrent_prepare(normed, st, inr, rf, index=batches[0])
        plan = F.QuadConcurrentStepPlan(net, prepared, 0.1, params, update=update)

        def step(i):
            F.quad_concurrent_prepare(normed, st, inr, rf, index=batches[i % len(batches)],
                                      out=prepared)
            return plan.launch()
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    print(mode, "ms per step", (time.perf_counter() - t0) / steps * 1e3, flush=True)
