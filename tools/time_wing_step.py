"""One fused fixed-wing training step (TrainFixedWing.train_concurrent_fused,
B = 131 072, H = 20 - bench.py's `secondary.wing_train_step`) for per-kernel
profiling:
    rocprofv3 --kernel-trace --stats -- python tools/time_wing_step.py
Prints the wall time per step over 30 steps (as bench.py times it)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import FixedWingDynamics
from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
dev = torch.device("cuda:0")
B, H, dt = 131072, 20, 0.05
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B,
           state_size=12, horizon=H, ref_dim=3, action_dim=4,
           learning_rate_controller=1e-9, system="fixed_wing", modified_params={})
w = FixedWingDynamics()
t = TrainFixedWing(w, w, cfg)
t.initialize_model(device=dev, seed=0)
d = t.state_data
def step():
    t.train_concurrent_fused(d.normed_states, d.states, d.in_ref_states, d.ref_states)
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
e0.record()
for _ in range(reps): step()
e1.record(); torch.cuda.synchronize()
print("wing step ms/step", e0.elapsed_time(e1) / reps)
