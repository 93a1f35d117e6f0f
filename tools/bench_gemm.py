"""Micro-benchmark of apg_planes_gemm on the shapes the fused policies use."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apg_trajectory_tracking_amd import functional as F

dev = torch.device("cuda:0")
B, H = 65536, 10
N = B * H + int(os.environ.get('NPAD', '0'))   # NPAD: plane pitch experiment
acts = torch.randn(431, N, device=dev)
d_pre = torch.randn(256, N, device=dev)
inr = torch.randn(2 * H * 9, B, device=dev)
Bc = B + int(os.environ.get('NPAD', '0')) // 8
d_conv = d_pre[:160]
R = lambda lo, hi: F.make_bdesc(dev, range(lo, hi))
refbuf = torch.randn(2 * H * 9 + (H + 1) * 12, Bc, device=dev)
conv_desc = F.make_bdesc(dev, [t * 9 + c for c in range(9) for t in range(3)] + [180, 181, 182],
                         [9] * 27 + [0] * 3, [9] * 27 + [12] * 3)
shapes = {
    "fc1a M64 J112": lambda: F.planes_gemm(d_pre[0:64], 64, 1, acts, R(15, 127), with_ones=False),
    "fc2  M64 J64+1": lambda: F.planes_gemm(d_pre[64:128], 64, 1, acts, R(239, 303)),
    "st   M64 J15+1": lambda: F.planes_gemm(d_pre[192:256], 64, 1, acts, R(0, 15)),
    "out  M4  J64+1": lambda: F.planes_gemm(d_pre[:4], 4, 1, acts, R(367, 431)),
    "lstm M32 J183+1": lambda: F.planes_gemm(d_pre[:32], 32, 1, acts, R(0, 183)),
    "conv M20 S80 J30+1": lambda: F.planes_gemm(d_conv, 20, 8 * H, refbuf, conv_desc,
                                                 sdiv=H, N=Bc),
}
planes = {"fc1a M64 J112": 176, "fc2  M64 J64+1": 128, "st   M64 J15+1": 79,
          "out  M4  J64+1": 68, "lstm M32 J183+1": 215, "conv M20 S80 J30+1": 160}
for wgs in [int(x) for x in os.environ.get("WGS", "0,256,512,1024").split(",")]:
    F._GEMM_WGS = wgs or None
    for name, fn in shapes.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = planes[name] * N * 4 / 1e9
        print(f"wgs {wgs:5d} {name:22s} {ms*1e3:8.1f} us  {gb/ms:7.2f} TB/s (HBM-unique)")
