"""The real epoch loop (TrainDrone.run_epoch: shuffled index batches of
B = 65 536 out of a resident data set, fused step per batch) for per-kernel
profiling of what a BATCH costs beyond the step:
    python tools/time_run_epoch.py concurrent|autoregressive|LSTM [graph|eager] [batches] [prefetch|noprefetch] [noepoch|epoch] [after_reverse|after_forward]
    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/time_run_epoch.py ...
    python tools/trace_step.py out/*/*_kernel_trace.csv <first kernel of a batch>"""
import contextlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (  # noqa: E402
    FlightmareDynamics)
from apg_trajectory_tracking_amd.train_drone import TrainDrone  # noqa: E402

dev = torch.device("cuda:0")
B, H, dt = 65536, 10, 0.1
mode = sys.argv[1] if len(sys.argv) > 1 else "concurrent"
graph = not (len(sys.argv) > 2 and sys.argv[2] == "eager")
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
# (the trainer's own default unless asked for: `prefetch` forces the gather pipeline
# one batch ahead, which the LSTM / autoregressive rows paths replaced in round 6)
prefetch = {"prefetch": True, "noprefetch": False}.get(sys.argv[4] if len(sys.argv) > 4 else "")
epoch_graph = not (len(sys.argv) > 5 and sys.argv[5] == "noepoch")
fork = sys.argv[6] if len(sys.argv) > 6 else None
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=nb * B, self_play=0, batch_size=B,
           state_size=12, horizon=H, train_mode=mode, ref_dim=9, action_dim=4,
           learning_rate_controller=1e-9, system="quad", modified_params={},
           save_name="time_run_epoch")
q = FlightmareDynamics()
t = TrainDrone(q, q, cfg)
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    t.initialize_model(device=dev, seed=0)
    t.graph_steps = graph
    if prefetch is not None:
        t.prefetch_batches = prefetch
    t.graph_epochs = epoch_graph
    if fork:
        t.gather_fork = fork
    if os.environ.get("APG_EPOCH_NO_SHUFFLE"):   # experiment: batches of consecutive rows
        t.trainloader.shuffle = False
    t.run_epoch("controller", 0)     # (graph_epochs: eager epoch, then the capture)
    t.run_epoch("controller", 0)
    torch.cuda.synchronize()
    epochs = 4
    t0 = time.perf_counter()
    for e in range(epochs):
        t.run_epoch("controller", e + 1)
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / (epochs * nb) * 1e3
print(f'{{"mode": "{mode}", "graphed": {str(graph).lower()}, "prefetch": {str(bool(t.prefetch_batches)).lower()}, "epoch_graph": {str(epoch_graph and graph).lower()}, "batches_per_epoch": {nb}, '
      f'"gather_fork": "{t.gather_fork}", "ms_per_batch": {ms:.4f}, '
      f'"epoch_loop": "{t.last_epoch_loop}", "launch_form": "{t.launch_form.get(mode)}"}}')
