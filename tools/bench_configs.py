#!/usr/bin/env python
"""Secondary measurements for the BASELINE.json configs that are not the
bench.py line (they are parity-test cases; this script records what they cost
on one MI355X for DESIGN.md).  Prints one JSON object per line.

  wing      config 4: fixed-wing concurrent, H = 20, B = 131 072 (fused kernel)
  cartpole  config 1: B = 64, H = 5 (fused kernel; launch-latency bound)
  quad_aos  config 2 through the reference's row-major tensors
  quad_ar_* config 3 shape per GPU: autoregressive unroll, policy in the loop
            (unfused: torch policy + step kernels; fused: mlp.hip)
  quad_lstm config 5: LSTM unroll
  quad_train config 2 as a FULL training step (policy fwd/bwd + rollout + SGD)
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps  # ms


def emit(name, B, H, ms, extra=None):
    out = {"config": name, "batch": B, "horizon": H, "ms_per_step": ms,
           "env_steps_per_s": B * H / ms * 1e3}
    out.update(extra or {})
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    want = lambda n: not args.only or n in args.only.split(",")
    nsets = 8

    if want("wing"):
        from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import FixedWingDynamics
        dyn = FixedWingDynamics()
        B, H, dt = 131072, 20, 0.05
        plans = []
        for i in range(4):
            d = synthetic.wing_batch(B, H, dt, seed=i)
            plans.append(F.RolloutPlan(
                "wing", synthetic.to_soa_state(d["state0"]).to(dev),
                synthetic.to_soa_seq(d["actions"]).to(dev),
                synthetic.to_soa_seq(d["ref"]).to(dev), dt, dyn.params,
                layout="soa", loss_mode="none"))
        it = [0]

        def step():
            plans[it[0] % 4].launch()
            it[0] += 1
        ms = timed(step, args.steps, 10)
        emit("wing", B, H, ms, {"algorithmic_GBps": B * (48 + 16 * H + 12 * H + 16 * H) / ms / 1e6})

    if want("wing_train"):
        from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import FixedWingDynamics
        from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
        B, H, dt = 131072, 20, 0.05
        cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0,
                   batch_size=B, state_size=12, horizon=H, ref_dim=3, action_dim=4,
                   learning_rate_controller=1e-9, system="fixed_wing",
                   modified_params={})
        wdyn = FixedWingDynamics()
        tw = TrainFixedWing(wdyn, wdyn, cfg)
        tw.initialize_model(device=dev, seed=0)
        dw = tw.state_data

        def step():
            acts = torch.sigmoid(tw.net(dw.normed_states, dw.in_ref_states))
            tw.train_controller_model(dw.states, acts.reshape(-1, H, 4),
                                      dw.in_ref_states, dw.ref_states)
        emit("wing_train_step", B, H, timed(step, 20, 5))

        def step_fused():
            tw.train_concurrent_fused(dw.normed_states, dw.states, dw.in_ref_states,
                                      dw.ref_states)
        emit("wing_train_step_fused_policy", B, H, timed(step_fused, 20, 5))

    if want("cartpole"):
        from apg_trajectory_tracking_amd.dynamics.cartpole_dynamics import CartpoleDynamics
        dyn = CartpoleDynamics()
        B, H = 64, 5
        d = synthetic.cartpole_batch(B, H, seed=0)
        s0, a = d["state0"].to(dev), d["actions"].to(dev)
        out = F.cartpole_rollout_fwd_bwd(s0, a, 0.05, dyn.params)
        ms = timed(lambda: F.cartpole_rollout_fwd_bwd(s0, a, 0.05, dyn.params, out=out), args.steps * 5, 20)
        emit("cartpole", B, H, ms)

    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import FlightmareDynamics
    qdyn = FlightmareDynamics()
    B, H, dt = 65536, 10, 0.1

    if want("quad_aos"):
        plans = []
        for i in range(nsets):
            d = synthetic.quad_polynomial_batch(B, H, dt, seed=i)
            plans.append(F.RolloutPlan("quad", d["state0"].to(dev), d["actions"].to(dev),
                                       d["ref"].to(dev), dt, qdyn.params, layout="aos",
                                       loss_mode="none"))
        it = [0]

        def step():
            plans[it[0] % nsets].launch()
            it[0] += 1
        emit("quad_aos", B, H, timed(step, args.steps * 2, 20))

    def make_trainer(mode):
        from apg_trajectory_tracking_amd.train_drone import TrainDrone
        cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0,
                   batch_size=B, state_size=12, horizon=H, train_mode=mode,
                   ref_dim=9, action_dim=4, learning_rate_controller=1e-9,
                   system="quad", modified_params={})
        t = TrainDrone(qdyn, qdyn, cfg)
        torch.manual_seed(0)
        t.initialize_model(device=dev, seed=0)
        t.hidden_generator = torch.Generator(device=dev).manual_seed(1)  # draw on the GPU
        return t

    if want("quad_train"):
        t = make_trainer("concurrent")
        d = t.state_data

        def step():
            acts = torch.sigmoid(t.net(d.normed_states, d.in_ref_states))
            t.train_controller_model(d.states, acts.reshape(-1, H, 4),
                                     d.in_ref_states, d.ref_states)
        emit("quad_train_step_aos", B, H, timed(step, 30, 5))

        ref_soa = synthetic.to_soa_seq(d.ref_states)
        s0_soa = synthetic.to_soa_state(d.states)

        def step_soa():
            t.optimizer_controller.zero_grad()
            acts = torch.sigmoid(t.net.forward_soa(d.normed_states, d.in_ref_states))
            loss = F.quad_rollout_loss(s0_soa, acts.reshape(H, 4, -1), ref_soa, dt,
                                       qdyn.params, layout="soa")
            t._step(loss)
        emit("quad_train_step_soa_head", B, H, timed(step_soa, 30, 5))

        # row-layout path (run_epoch's branch for ANY PyTorch policy): the data
        # set's cached packed tensors, Net.forward_packed -> [H, B, 4] rows ->
        # quad_rollout_rows_kernel
        rows = d.packed()

        def step_packed():
            t.train_controller_packed(d.normed_states, d.in_ref_states, *rows)
        emit("quad_train_step_packed_rows", B, H, timed(step_packed, 30, 5))

        def step_fused():      # policy inside the kernels
            t.train_concurrent_fused(d.normed_states, d.states, d.in_ref_states,
                                     d.ref_states)
        emit("quad_train_step_fused_policy", B, H, timed(step_fused, 30, 5))

    if want("quad_run_epoch"):
        # the real epoch loop (whole-tensor shuffled batches + fused step)
        from apg_trajectory_tracking_amd.train_drone import TrainDrone
        nb = 8
        cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=nb * B, self_play=0,
                   batch_size=B, state_size=12, horizon=H, train_mode="concurrent",
                   ref_dim=9, action_dim=4, learning_rate_controller=1e-9,
                   system="quad", modified_params={})
        te = TrainDrone(qdyn, qdyn, cfg)
        te.initialize_model(device=dev, seed=0)
        te.run_epoch("controller", 0)
        torch.cuda.synchronize()
        import time as _t
        t0 = _t.perf_counter()
        for e in range(3):
            te.run_epoch("controller", e + 1)
        torch.cuda.synchronize()
        emit("quad_run_epoch_per_batch", B, H, (_t.perf_counter() - t0) / (3 * nb) * 1e3)
        # the same loop with the minibatch steps replayed from ONE captured
        # graph (TrainBase.graph_steps: persistent index buffer)
        te.graph_steps = True
        te.run_epoch("controller", 4)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for e in range(3):
            te.run_epoch("controller", e + 5)
        torch.cuda.synchronize()
        emit("quad_run_epoch_per_batch_graphed", B, H,
             (_t.perf_counter() - t0) / (3 * nb) * 1e3)
        for mode in ("autoregressive", "LSTM"):
            cfg_r = dict(cfg, train_mode=mode, epoch_size=4 * B)
            tr = TrainDrone(qdyn, qdyn, cfg_r)
            tr.initialize_model(device=dev, seed=0)
            for graphed in (False, True):
                tr.graph_steps = graphed
                tr.run_epoch("controller", 0)
                torch.cuda.synchronize()
                t0 = _t.perf_counter()
                for e in range(2):
                    tr.run_epoch("controller", e + 1)
                torch.cuda.synchronize()
                emit(f"quad_run_epoch_per_batch_{mode}" + ("_graphed" if graphed else ""),
                     B, H, (_t.perf_counter() - t0) / (2 * 4) * 1e3)
            del tr
            torch.cuda.empty_cache()

    for mode, name, fused in (("autoregressive", "quad_ar_unfused", False),
                              ("autoregressive", "quad_ar_fused", True),
                              ("LSTM", "quad_lstm_unfused", False),
                              ("LSTM", "quad_lstm_fused", True)):
        if want(name):
            t = make_trainer(mode)
            t.fused_policy = fused
            d = t.state_data
            emit(name, B, H, timed(lambda: t.train_recurrent_model(
                None, d.states, d.in_ref_states, d.ref_states), 10, 2))


    if want("quad_closed_loop"):
        # N2: batched closed-loop evaluation (251 steps each) with an untrained
        # AR policy; resets keep every run alive for all steps
        from apg_trajectory_tracking_amd.models.hutter_model import Net
        torch.manual_seed(0)
        net = Net(15, 10, 9, 4, conv=1).to(dev)
        nt, L, steps = 16384, 501, 251
        traj = synthetic.quad_eval_trajectories(nt, L, dt, seed=2).to(dev)
        traj[:, :, 2] += 3

        def step():
            F.quad_mlp_closed_loop(net, traj, dt, qdyn.params, max_steps=steps,
                                   thresh_div=1.0, thresh_stable=1.0, test_time=0)
        emit("quad_closed_loop", nt, steps, timed(step, 5, 2))


if __name__ == "__main__":
    main()
