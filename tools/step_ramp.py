"""How long a training step takes to reach its steady rate behind an idle device:
    python tools/step_ramp.py concurrent|LSTM [idle_ms]
One event every 25 steps over 1500 back-to-back steps behind `idle_ms` of host
sleep (default 50); prints ms/step per segment."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.train_drone import TrainDrone
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
dev = torch.device("cuda:0")
B, H, dt = 65536, 10, 0.1
mode = sys.argv[1]
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B, state_size=12,
           horizon=H, train_mode=mode, ref_dim=9, action_dim=4, learning_rate_controller=1e-9,
           system="quad", modified_params={})
q = FlightmareDynamics()
t = TrainDrone(q, q, cfg)
t.initialize_model(device=dev, seed=0)
t.static_shard = t.graph_steps = t.borrow_loss = True
d = t.state_data
def step():
    if mode == "concurrent":
        t.train_concurrent_fused(d.normed_states, d.states, d.in_ref_states, d.ref_states)
    else:
        t.train_recurrent_model(d.normed_states, d.states, d.in_ref_states, d.ref_states)
for _ in range(40): step()
torch.cuda.synchronize()
for trial in range(2):
    time.sleep(idle * 1e-3)
    seg, n = 25, 60
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        for _ in range(seg): step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) / seg for i in range(n)]
    print(mode, "idle_ms", idle, "ms/step per 25 steps:", " ".join("%.4f" % m for m in ms))
