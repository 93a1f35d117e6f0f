"""Does the gap between two replays of the SAME captured step graph come from
the executable graph waiting for its own previous launch?  Captures the
concurrent training step twice (two executable graphs, private pools, same
parameters) and times  A A A A ...  against  A B A B ...  (and, as the floor,
the five kernels enqueued directly by the C entry point with every buffer
pre-allocated):
    python tools/ab_graph_alternation.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd.train_drone import TrainDrone
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
dev = torch.device("cuda:0")
B, H, dt = 65536, 10, 0.1
cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B, state_size=12,
           horizon=H, train_mode="concurrent", ref_dim=9, action_dim=4,
           learning_rate_controller=1e-9, system="quad", modified_params={})
q = FlightmareDynamics()
t = TrainDrone(q, q, cfg)
torch.manual_seed(0)
t.initialize_model(device=dev, seed=0)
t.static_shard, t.graph_steps, t.borrow_loss = True, True, True
t.plan_steps = False           # (this part is about captured graphs)
d = t.state_data
step = lambda: t.train_concurrent_fused(d.normed_states, d.states, d.in_ref_states, d.ref_states)
step(); step()
ga = t._graphs["concurrent"]
t._graphs.clear()
step(); step()
gb = t._graphs["concurrent"]
assert ga is not gb


def timed(fn, n=400):
    for _ in range(20):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    same = timed(lambda i: ga.graph.replay())
    alt = timed(lambda i: (ga if i & 1 else gb).graph.replay())
    t.plan_steps = True        # ... and the trainer's default: the step plan
    via = timed(lambda i: step())
    t.plan_steps = False
    print(f'{{"same_graph_ms": {same:.4f}, "alternating_ms": {alt:.4f}, '
          f'"through_trainer_ms": {via:.4f}}}')

# the floor: the entry point called directly (five launches from one C call,
# nothing else between steps: every buffer and struct made once)
import ctypes
from apg_trajectory_tracking_amd import _capi, functional as F
lib = _capi.lib()
net = t.net
acts, s0, rf = F.quad_concurrent_prepare(d.normed_states, d.states, d.in_ref_states,
                                         d.ref_states)
names = ("w_s", "b_s", "conv_w", "conv_b", "w_1", "b_1", "w_2", "b_2", "w_3", "b_3",
         "w_out", "b_out")
params = [p.detach() for p in F._net_params(net, F._MLP_PARAMS)]
new = lambda *s: torch.zeros(s, device=dev)
struct = lambda ts: _capi.ApgMlpPolicyGrads(**{k: v.data_ptr() for k, v in zip(names, ts)})
pol = _capi.ApgMlpPolicy(**{k: v.data_ptr() for k, v in zip(names, params)})
grads, bufs = [new(*p.shape) for p in params], [new(*p.shape) for p in params]
gs = struct(grads)
upd = _capi.ApgMlpSgdUpdate(lr=1e-9, momentum=0.9, param=struct(params),
                            momentum_buf=struct(bufs))
mask = torch.empty(5, B, dtype=torch.int32, device=dev)
dz, lp = new(40, B), new(lib.apg_quad_mlp_loss_partials_count(B))
loss = new(1)
ws = new(lib.apg_quad_mlp_step_workspace_floats())
part = new(lib.apg_quad_mlp_step_partials_floats(B))
w = F.quad_loss_weights()
st = torch.cuda.current_stream().cuda_stream
args = (s0.data_ptr(), rf.data_ptr(), rf.shape[1], dt, ctypes.byref(q.params),
        ctypes.byref(w), ctypes.byref(pol), B, H, acts.data_ptr(), mask.data_ptr(),
        dz.data_ptr(), lp.data_ptr(), loss.data_ptr(), ctypes.byref(gs), None,
        ws.data_ptr(), part.data_ptr(), ctypes.byref(upd), None, st)
fn = lib.apg_quad_mlp_concurrent_train_step
for rep in range(3):
    direct = timed(lambda i: fn(*args))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        fn(*args)
    host = (time.perf_counter() - t0) / 400 * 1e3      # enqueue cost alone
    torch.cuda.synchronize()
    print(f'{{"direct_c_call_ms": {direct:.4f}, "host_enqueue_ms": {host:.4f}}}')
