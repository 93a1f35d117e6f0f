"""Debug aid for the in-sweep weight gradients: which trajectories' terms
delta_n x_n^T does the kernel's dW_out contain?"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic, _capi
from apg_trajectory_tracking_amd.dataset import state_preprocessing
from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
from apg_trajectory_tracking_amd.models.hutter_model import Net
dev = torch.device("cuda:0")
H, dt = 10, 0.1
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(8)
net = Net(15, H, 9, 40, conv=1).to(dev)
d = synthetic.quad_polynomial_batch(B, H, dt, seed=B)
s0 = d["state0"].to(dev)
with torch.no_grad():
    normed = state_preprocessing(s0)
prep = F.quad_concurrent_prepare(normed, s0, d["in_ref"].to(dev), d["ref"].to(dev))
acts = prep[0]
F.CONCURRENT_IN_SWEEP = True
loss, gr, flat = F.quad_concurrent_policy_grads(net, None, None, None, None, dt,
                                                FlightmareDynamics().params, prepared=prep)
torch.cuda.synchronize()
h3 = acts[367:431].double().cpu().numpy()      # [64][B]
h2 = acts[303:367].double().cpu().numpy()
# d_zout is scratch inside forward: recompute the plane path's to get it
F.CONCURRENT_IN_SWEEP = False
ctx = F._DirectCtx(); ctx.prepared = prep
with torch.no_grad():
    F._QuadConcurrentPolicyLoss.forward(ctx, None, None, None, None,
        *F._net_params(net, F._MLP_PARAMS), dt, FlightmareDynamics().params,
        F.quad_loss_weights(), None)
_, cot = ctx.saved_tensors
dz = cot[:40].double().cpu().numpy()            # [40][B]
dp3 = cot[40 + 128:40 + 192].double().cpu().numpy()
got = gr["fc_out.weight"].double().cpu().numpy()
got3 = gr["fc3.weight"].double().cpu().numpy()
print("B", B)
full = dz @ h3.T
print("fc_out: err vs full sum", np.abs(got - full).max() / np.abs(full).max())
for n in range(min(B, 8)):
    one = np.outer(dz[:, n], h3[:, n])
    # least-squares coefficient of this trajectory's term in `got`
    others = full - one
    print(" traj", n, "coef", float((got * one).sum() / (one * one).sum()))
# permutation hypothesis: got = sum_n dz[:, n] h3[:, perm(n)]^T ?
if B <= 64:
    C = np.zeros((B, B))
    for a in range(B):
        for b in range(B):
            C[a, b] = (got * np.outer(dz[:, a], h3[:, b])).sum()
    # solve got ~ sum_ab M[a,b] dz_a h3_b^T  (Gram of outer products)
    G = (dz.T @ dz)[:, None, :, None] * (h3.T @ h3)[None, :, None, :]
    M = np.linalg.lstsq(G.reshape(B * B, B * B), C.reshape(-1), rcond=None)[0].reshape(B, B)
    np.set_printoptions(precision=2, suppress=True, linewidth=200)
    print("pairing matrix M[a, b] (delta of trajectory a with x of trajectory b):")
    print(M[:min(B, 16), :min(B, 16)])
