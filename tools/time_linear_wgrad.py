"""apg_linear_wgrad against torch's own weight gradient (rocBLAS) on the policy
layer shapes, B = 65 536 rows:
    python tools/time_linear_wgrad.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import nn as apg_nn  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B = 65536
    for M, N in ((64, 64), (64, 224), (40, 64), (64, 15), (64, 128), (128, 256)):
        x = torch.randn(B, N, device=dev)
        dy = torch.randn(B, M, device=dev)
        w = torch.randn(M, N, device=dev, requires_grad=True)
        b = torch.randn(M, device=dev, requires_grad=True)

        def ours():
            y = apg_nn.linear(x, w, b)
            torch.autograd.grad(y, (w, b), dy)

        def theirs():
            y = torch.nn.functional.linear(x, w, b)
            torch.autograd.grad(y, (w, b), dy)

        def fwd_only():
            torch.nn.functional.linear(x, w, b)
        f = timed(fwd_only)
        print(json.dumps({"B": B, "M": M, "N": N,
                          "us_fwd_plus_wgrad_apg": timed(ours),
                          "us_fwd_plus_wgrad_torch": timed(theirs), "us_fwd_only": f,
                          "stream_floor_us_at_5TBps": (M + N) * B * 4 / 5e12 * 1e6}))


if __name__ == "__main__":
    main()
