"""Kernel-level timing of the concurrent step's gradient computation (no SGD)
at B = 65 536: `python tools/time_conc_kernels.py [planes]` (default: in-sweep);
meant for `rocprofv3 --kernel-trace --stats`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic
from apg_trajectory_tracking_amd.dataset import state_preprocessing
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net
dev = torch.device("cuda:0")
B, H, dt = int(os.environ.get("B", 65536)), 10, 0.1
F.CONCURRENT_IN_SWEEP = not (len(sys.argv) > 1 and sys.argv[1] == "planes")
torch.manual_seed(8)
net = Net(15, H, 9, 40, conv=1).to(dev)
d = synthetic.quad_polynomial_batch(B, H, dt, seed=0)
s0 = d["state0"].to(dev)
with torch.no_grad():
    normed = state_preprocessing(s0)
in_ref, ref = d["in_ref"].to(dev), d["ref"].to(dev)
p = FlightmareDynamics().params
step = lambda: F.quad_concurrent_policy_grads(net, normed, s0, in_ref, ref, dt, p, static_inputs=True)
for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    step()
e1.record(); torch.cuda.synchronize()
print("in_sweep" if F.CONCURRENT_IN_SWEEP else "planes", "eager us/step", e0.elapsed_time(e1) / 50 * 1e3)
