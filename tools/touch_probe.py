"""Probe (round 6): the concurrent rows step with the NEXT batch's rows touched while the
current step runs (side stream) / right before the step / not at all.
    python tools/touch_probe.py none|before|beside
32 shuffled batches of B = 65 536 out of 32 x B rows; prints ms per batch (events)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic
from apg_trajectory_tracking_amd.dataset import state_preprocessing
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net
how = sys.argv[1] if len(sys.argv) > 1 else "none"
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libtouch_probe.so"))
lib.touch_rows.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, H, dt, nb = 65536, 10, 0.1, 32
d = synthetic.quad_polynomial_batch(nb * B, H, dt, seed=5, ref_length=20)
st, inr, rf = (d[k].to(dev).contiguous() for k in ("state0", "in_ref", "ref"))
with torch.no_grad():
    normed = state_preprocessing(st).contiguous()
torch.manual_seed(1)
net = Net(15, H, 9, 4 * H, conv=1).to(dev)
bufs = {n: torch.zeros_like(p) for n, p in net.named_parameters() if n in F._MLP_PARAMS}
plan = F.QuadConcurrentStepPlan(net, None, dt, FlightmareDynamics().params,
                                update=(1e-9, 0.9, bufs), rows=(normed, st, inr, rf, B))
side = torch.cuda.Stream()
def touch(index, stream):
    for t, used in ((inr, 360), (rf, 360), (normed, 60), (st, 48)):
        lib.touch_rows(t.data_ptr(), t.stride(0) * 4, used, index.data_ptr(), B,
                       t.numel() * 4, ctypes.c_void_p(stream.cuda_stream))
def epoch():
    order = torch.randperm(nb * B, device=dev)
    batches = [order[i * B:(i + 1) * B] for i in range(nb)]
    main = torch.cuda.current_stream()
    for i, index in enumerate(batches):
        if how == "before":
            touch(index, main)
        if how == "beside" and i + 1 < nb:      # the next batch's rows, beside this step
            side.wait_stream(main)              # (behind the previous step)
            touch(batches[i + 1], side)
        plan.launch(index=index)
for _ in range(3): epoch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6): epoch()
e1.record(); torch.cuda.synchronize()
print(how, "ms/batch", e0.elapsed_time(e1) / (6 * nb))
