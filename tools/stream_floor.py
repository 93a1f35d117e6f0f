"""bench.py's `roofline.stream_floor` alone: the headline
launch's bytes in the fast copy shape, graph replays, 20 buffer sets.
    python tools/stream_floor.py > gpurun_out/stream_floor.json"""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = types.SimpleNamespace(horizon=10, batch=65536)
print(json.dumps(bench.stream_floor_probe(args, torch.device("cuda:0"), 20), indent=1))
