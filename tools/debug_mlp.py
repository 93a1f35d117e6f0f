"""Stage-by-stage check of the fused autoregressive kernels (mlp.hip) against
a plain torch evaluation of the same network: prints the relative error of
every saved plane, so a layout bug is localised in one GPU run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apg_trajectory_tracking_amd import functional as F, synthetic
from apg_trajectory_tracking_amd.dataset import state_preprocessing
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net

dev = torch.device("cuda:0")
B, H = int(os.environ.get("B", 96)), 10
torch.manual_seed(1)
net = Net(15, 10, 9, 4, conv=1).to(dev)
d = synthetic.quad_polynomial_batch(B, 10, 0.1, seed=3, ref_length=20)
state0, in_ref, ref = d["state0"].to(dev), d["in_ref"].to(dev), d["ref"].to(dev)
dyn = FlightmareDynamics()


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# torch reference with all intermediates
cur = state0.clone()
R = dict(feat=[], s1=[], cv=[], h1=[], h2=[], h3=[], act=[], st=[])
with torch.no_grad():
    for k in range(H):
        rel_w = in_ref[:, k:k + H].clone()
        rel_w[:, :, :3] -= cur[:, None, :3]
        f = state_preprocessing(cur)
        s1 = torch.tanh(net.states_in(f))
        cv = torch.relu(net.conv_ref(rel_w.transpose(1, 2))).reshape(-1, 160)
        h1 = torch.tanh(net.fc1(torch.cat((s1, cv), 1)))
        h2 = torch.tanh(net.fc2(h1))
        h3 = torch.tanh(net.fc3(h2))
        a = torch.sigmoid(net.fc_out(h3))
        cur = dyn(cur, a, dt=0.1)
        for key, v in zip(R, (f, s1, cv, h1, h2, h3, a, cur)):
            R[key].append(v)
R = {k: torch.stack(v, 0) for k, v in R.items()}   # [H,B,*]

loss, states, actions = F.quad_mlp_rollout_loss(net, state0, in_ref, ref, 0.1, dyn.params)
fn = loss.grad_fn
refbuf, acts, d_pre, d_zout, d_conv = fn.saved_tensors
N = H * B
pl = lambda t, lo, hi: t[lo:hi].reshape(hi - lo, H, B).permute(1, 2, 0)
print("feat", rel(pl(acts, 0, 15), R["feat"]))
print("s1  ", rel(pl(acts, 15, 79), R["s1"]))
print("cv  ", rel(pl(acts, 79, 239), R["cv"]))
print("h1  ", rel(pl(acts, 239, 303), R["h1"]))
print("h2  ", rel(pl(acts, 303, 367), R["h2"]))
print("h3  ", rel(pl(acts, 367, 431), R["h3"]))
print("act ", rel(actions.permute(0, 2, 1), R["act"]))
print("st  ", rel(states.permute(0, 2, 1), R["st"]))

# gradients vs torch autograd of the unfused path
net2 = Net(15, 10, 9, 4, conv=1).to(dev)
net2.load_state_dict(net.state_dict())
cur = state0.clone()
sts, acs = [], []
for k in range(H):
    rel_w = in_ref[:, k:k + H].clone()
    rel_w[:, :, :3] = rel_w[:, :, :3] - cur[:, None, :3]
    a = torch.sigmoid(net2(state_preprocessing(cur), rel_w))
    cur = dyn(cur, a, dt=0.1)
    sts.append(cur), acs.append(a)
from apg_trajectory_tracking_amd.drone_loss import quad_mpc_loss
l2 = quad_mpc_loss(torch.stack(sts, 1), ref[:, :H], torch.stack(acs, 1))
l2.backward()
loss.backward()
print("loss", loss.item(), l2.item())
for (k, p), (_, q) in zip(net.named_parameters(), net2.named_parameters()):
    if q.grad is not None:
        print("grad", k, rel(p.grad, q.grad))
