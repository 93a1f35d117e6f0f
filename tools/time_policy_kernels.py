"""Kernel-only timing of the in-kernel-policy sweeps (forward / reverse) at
B = 65536 through the C ABI, e.g. for the APG_LIB experiment builds."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apg_trajectory_tracking_amd import _capi, functional as F, synthetic
from apg_trajectory_tracking_amd._capi import lib, check, ptr, stream_of
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net

dev = torch.device("cuda:0")
B, H = int(os.environ.get("B", 65536)), 10
N = B * H
torch.manual_seed(0)
net = Net(15, 10, 9, 4, conv=1).to(dev)
d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=3, ref_length=20)
dyn = FlightmareDynamics()
refbuf, inr, s0, states = F._ref_and_states(d["in_ref"].to(dev), d["state0"].to(dev), B, H)
rf = d["ref"].to(dev)[:, :H].permute(1, 2, 0).contiguous()
names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3", "w_out", "b_out")
ws_ = [net.states_in.weight, net.states_in.bias, net.conv_ref.weight, net.conv_ref.bias,
       net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, net.fc3.weight, net.fc3.bias,
       net.fc_out.weight, net.fc_out.bias]
pw = {k: v.detach().contiguous() for k, v in zip(names, ws_)}
pol = _capi.ApgMlpPolicy(**{k: ptr(v) for k, v in pw.items()})
new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
actions, acts = new(H, 4, B), new(431, N)
feat, x1, h = acts[:15], acts[15:239], acts[239:431]
mask = torch.empty(5, N, dtype=torch.int32, device=dev)
ws = new(lib().apg_quad_mlp_workspace_floats())
partials, loss = new(lib().apg_quad_mlp_loss_partials_count(B)), new(1)
d_pre, d_zout, d_conv = new(256, N), new(4, N), new(720, B)   # d_conv: diagonal sums
st = stream_of(s0)
wts = F.quad_loss_weights()


def fwd():
    check(lib().apg_quad_mlp_rollout_fwd(
        ptr(s0), ptr(inr), 0.1, ctypes.byref(dyn.params), ctypes.byref(pol), B, H,
        ptr(states), ptr(actions), ptr(feat), ptr(x1), ptr(h), mask.data_ptr(), ptr(ws), st), "fwd")


def bwd():
    check(lib().apg_quad_mlp_rollout_bwd(
        ptr(s0), ptr(states), ptr(actions), ptr(rf), 9, ptr(x1), ptr(h), mask.data_ptr(), 0.1,
        ctypes.byref(dyn.params), ctypes.byref(wts), ctypes.byref(pol), B, H, ptr(partials),
        ptr(loss), ptr(d_pre), ptr(d_zout), ptr(d_conv), None, ptr(ws), st), "bwd")


for name, fn in (("mlp fwd", fwd), ("mlp bwd", bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('APG_LIB', 'default')[-20:]:>20s} {name}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us (incl. pack kernel)")
