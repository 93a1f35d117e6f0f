"""Drop-in for the three MPC-aligned losses of neural_control/drone_loss.py
(quad_mpc_loss :12-39, fixed_wing_mpc_loss :72-82, cartpole_loss_mpc
:136-145).  Unlike the reference, importing this module does NOT switch on
torch.autograd anomaly detection (drone_loss.py:6)."""
from . import functional as F


def quad_mpc_loss(states, ref_states, action_seq, printout=0):
    """states [B,H,12], ref_states [B,H,9], action_seq [B,H,4] -> scalar
    (sum over batch and horizon, weights pos 10 / vel 1 / av 0.1 /
    rates 0.1 / thrust 5)."""
    return F.quad_loss(states, ref_states, action_seq)


def fixed_wing_mpc_loss(drone_states, linear_reference, action, printout=0):
    """drone_states [B,H,12], linear_reference [B,H,3], action [B,H,4] ->
    10 * sum (pos - ref)^2 + 0.1 * sum (action[:, :, 1:] - 0.5)^2.
    Cold path (the trainers use the fused rollout): plain device torch ops."""
    a_loss = ((action[:, :, 1:] - 0.5)**2).sum()
    p_loss = ((drone_states[:, :, :3] - linear_reference)**2).sum()
    return 10 * p_loss + 0.1 * a_loss


def cartpole_loss_mpc(states, ref_states, actions):
    """sum((states - ref)^2 * [0, 3, 10, 1]) + 0.01 * sum(actions^2)."""
    w = states.new_tensor([0.0, 3.0, 10.0, 1.0])
    return ((states - ref_states)**2 * w).sum() + 0.01 * (actions**2).sum()
