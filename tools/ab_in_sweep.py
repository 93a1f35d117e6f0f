"""In-sweep weight gradients (round 5) against the plane + product path of
rounds 1-4, autoregressive step at B = 65 536 (gradients only, eager launches,
resident inputs prepared once); the LSTM step (planes: its only form):
    python tools/ab_in_sweep.py [ar|lstm ...] [in|planes]
The plane sequence of the autoregressive step is tests/plane_path.py (round 6:
the package has one path per mode)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from apg_trajectory_tracking_amd import functional as F, synthetic
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net
from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
dev = torch.device("cuda:0")
B, H, DT = 65536, 10, 0.1
d = synthetic.quad_polynomial_batch(B, H, DT, seed=1, ref_length=2 * H)
inputs = tuple(d[k].to(dev) for k in ("state0", "in_ref", "ref"))
prepared = F.quad_recurrent_prepare(*inputs)
dyn = FlightmareDynamics()
torch.manual_seed(0)
h0, c0 = torch.randn(8, B, device=dev).t(), torch.randn(8, B, device=dev).t()
nets = {"ar": Net(15, H, 9, 4, conv=1).to(dev), "lstm": LSTM_NEW(15, H, 9, 4, conv=1).to(dev)}


def step(mode, in_sweep=True):
    if mode == "ar" and not in_sweep:
        import plane_path
        return plane_path.quad_mlp_rollout_grads_planes(nets[mode], None, None, None, DT,
                                                        dyn.params, prepared=prepared)
    if mode == "ar":
        return F.quad_mlp_rollout_grads(nets[mode], None, None, None, DT, dyn.params,
                                        prepared=prepared)
    return F.quad_lstm_rollout_grads(nets[mode], None, None, None, DT, dyn.params, h0, c0,
                                     prepared=prepared)


only = [a for a in sys.argv[1:] if a in ("in", "planes")]
for mode in ([a for a in sys.argv[1:] if a in ("ar", "lstm")] or ["ar", "lstm"]):
    for rep in range(2):
        for on in ((True, False) if not only else (only[0] == "in",)):
            if mode == "lstm" and on:
                continue            # (the LSTM step has no in-sweep form: tools/patches)
            for _ in range(5):
                step(mode, on)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                step(mode, on)
            e1.record()
            torch.cuda.synchronize()
            print(mode, "in_sweep" if on else "planes  ", "ms/step %.4f" % (e0.elapsed_time(e1) / 30))
