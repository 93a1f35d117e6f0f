"""cProfile of TrainDrone.run_epoch's host side (one epoch of 32 batches of 65 536):
    python tools/host_profile_epoch.py LSTM|autoregressive|concurrent [eager]
`eager`: graph_steps off.  Prints ms per batch and the 25 most expensive functions."""
import contextlib, cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.train_drone import TrainDrone
dev = torch.device("cuda:0")
B, H, dt, nb = 65536, 10, 0.1, 32
mode = sys.argv[1]
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=nb * B, self_play=0, batch_size=B,
           state_size=12, horizon=H, train_mode=mode, ref_dim=9, action_dim=4,
           learning_rate_controller=1e-9, system="quad", modified_params={},
           save_name="host_profile_epoch")
q = FlightmareDynamics()
t = TrainDrone(q, q, cfg)
with contextlib.redirect_stdout(sys.stderr):
    t.initialize_model(device=dev, seed=0)
    t.graph_steps = not (len(sys.argv) > 2 and sys.argv[2] == "eager")
    for e in range(4):
        t.run_epoch("controller", e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(6):
        t.run_epoch("controller", e)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (6 * nb) * 1e3
    pr = cProfile.Profile()
    pr.enable()
    for e in range(3):
        t.run_epoch("controller", e)
    pr.disable()
print(mode, "graph_steps", t.graph_steps, "loop", t.last_epoch_loop, "form", dict(t.launch_form),
      "ms/batch", round(ms, 4))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
