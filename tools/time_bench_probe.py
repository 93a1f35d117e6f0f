"""bench.py's trainer-step block for ONE mode in a process of its own:
    python tools/time_bench_probe.py LSTM [--train-steps 400]
The default bench run takes this block after the headline loops and the other
modes' blocks; this is the same block without that history (allocator state,
live graphs), to tell a process effect from a step effect."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
mode = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
args = bench.parse()
dev = torch.device("cuda:0")
out = bench.trainer_step_probe(args, dev, FlightmareDynamics(), None, mode)
print(json.dumps({k: v for k, v in out.items() if k.startswith("ms_") or k in ("launch", "launch_form")}))
