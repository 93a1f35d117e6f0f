"""One fused training step of TrainDrone at B = 65 536 on a resident shard,
eager launches (no graph), for per-kernel profiling:
    python tools/time_train_step.py concurrent|autoregressive|LSTM [graph]
    APG_STEP_LEGACY=1 ...   the LSTM step without round 6's resident tables / tail launch
    APG_STEP_REPS=n ...     timed steps (default 200 replays / 20 eager steps)
    rocprofv3 --kernel-trace --stats -- python tools/time_train_step.py <mode>
Prints the eager wall time per step (host gaps included); with `graph` the
step is replayed from the trainer's captured graph (what bench.py times)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.train_drone import TrainDrone
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
dev = torch.device("cuda:0")
B, H, dt = 65536, 10, 0.1
mode = sys.argv[1] if len(sys.argv) > 1 else "concurrent"
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B, state_size=12,
           horizon=H, train_mode=mode, ref_dim=9, action_dim=4, learning_rate_controller=1e-9,
           system="quad", modified_params={})
q = FlightmareDynamics()
t = TrainDrone(q, q, cfg)
torch.manual_seed(0)
t.initialize_model(device=dev, seed=0)
t.static_shard = True
if os.environ.get("APG_STEP_LEGACY"):     # A/B: the step as it was before round 6's
    t.resident_tables = False             # LSTM tail (tables packed per step,
    t.in_kernel_update = False            # optimizer.step() as a launch of its own)
t.graph_steps = len(sys.argv) > 2 and sys.argv[2] == "graph"
reps = int(os.environ.get("APG_STEP_REPS", 200 if t.graph_steps else 20))
d = t.state_data
if os.environ.get("APG_STEP_SYNTH"):      # bench.py's shard: synthetic.quad_polynomial_batch
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    sd = synthetic.quad_polynomial_batch(B, H, dt, seed=0, ref_length=t.ref_length)
    class d:
        states, in_ref_states, ref_states = (sd[k].to(dev) for k in ("state0", "in_ref", "ref"))
    d.normed_states = state_preprocessing(d.states)
    t.state_data = d
    print("shapes", d.states.shape, d.in_ref_states.shape, d.ref_states.shape)
else:
    print("shapes", d.states.shape, d.in_ref_states.shape, d.ref_states.shape)
t.borrow_loss = bool(os.environ.get("APG_STEP_BORROW"))
def step():
    if mode == "concurrent":
        t.train_concurrent_fused(d.normed_states, d.states, d.in_ref_states, d.ref_states)
    else:
        t.train_recurrent_model(d.normed_states, d.states, d.in_ref_states, d.ref_states)
for _ in range(5): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
h0 = time.perf_counter()
e0.record()
for _ in range(reps): step()
e1.record(); torch.cuda.synchronize()
host_ms = (time.perf_counter() - h0) * 1e3 / reps     # the same region on the host's clock
print(mode, "graph" if t.graph_steps else "eager", "ms/step", e0.elapsed_time(e1) / reps,
      dict(t.launch_form), "host clock", host_ms)
if os.environ.get("APG_STEP_CHUNKS"):     # bench.py's clock: four host-timed chunks
    import bench
    print("bench.timed_steps", bench.timed_steps(lambda: step() or 0, 400, None)[0])
