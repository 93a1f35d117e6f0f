#!/usr/bin/env python
"""bench.py - env-steps/s of the fused APG rollout (forward + backward through
the quadrotor dynamics + quad_mpc_loss) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
on rank 0.  N > 1 runs one process per GPU under torch.distributed.run: either
the caller launches it that way (the driver's form) or - when RANK is not in
the environment - this file launches itself (`self_launch`), so the plain
command works for every N the node has GPUs for.

A "step" is one pass of the hot path over one batch: the fused kernel
apg_quad_rollout_fwd_bwd (H x dynamics, loss, analytic adjoint down to
dL/daction_seq) followed by the fixed-order loss reduction; inputs are
resident in HBM.  Workload = BASELINE.json configs[1]: quadrotor, concurrent
mode, horizon 10, batch 65 536 synthetic polynomial trajectories per GPU
(weak scaling: every rank owns its own 65 536-trajectory shard; the
dynamics-only metric has no cross-rank exchange).
The timing loop rotates over --sets independent buffer sets.  Round 3: the
default is 20 sets, so that the READ-ONLY inputs alone (29.4 MB per set) are
588 MB > 2 x the 256 MiB Infinity Cache and every launch reads them from HBM
(with SURVEY.md §8d's 8 sets the 235 MB of inputs stay cache resident: that
protocol, and the one-set cache-resident launch, are reported next to the
headline, labelled, never as `value`).
The K steps are captured into one HIP graph; the graph is replayed R times so
that the timed region is >= --min-ms (1 s: long enough for the driver's
GPU-activity sampling to see it); `steps` stays K, `config.replays` = R.
`--dry-run-cpu` executes the rank logic of this file (process group, replay
agreement, barriers, max over ranks, rank-0 print) under gloo on the CPU with
a stub in place of the kernels - tests/test_distributed_cpu.py runs it with
two ranks so that the first multi-GPU run cannot die on plumbing.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
QUAD_BYTES_PER_TRAJ = {  # state0 + actions + ref(pos,vel) + dL/dactions
    "base": lambda H: 48 + 16 * H + 24 * H + 16 * H,
    "grad_state0": 48,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--layout", choices=["packed", "soa", "aos"],
                    default="packed",
                    help="packed: rows [rows][B][C] (16-byte accesses per lane, "
                         "the fast path); soa: planes [C][B]; aos: the "
                         "reference's row-major tensors")
    ap.add_argument("--min-ms", type=float, default=1000.0,
                    help="replay the K-step graph until the timed region is "
                         "at least this long")
    ap.add_argument("--sets", type=int, default=20,
                    help="rotating buffer sets of the headline (20: the read-only "
                         "inputs alone are 588 MB > 2 x the Infinity Cache)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--grad-state0", action="store_true")
    ap.add_argument("--loss-mode", choices=["deferred", "eager", "none"],
                    default="deferred",
                    help="how each step's scalar loss is materialised "
                         "(DESIGN.md: deferred = folded into the next launch)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every step from Python instead of replaying "
                         "one captured HIP graph of the K steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline launches (for the rocprofv3 pass whose "
                         "per-kernel average must not mix protocols)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="rank logic only, gloo on the CPU, stub kernels")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the fixed-wing dual-roofline block")
    ap.add_argument("--train-steps", type=int, default=400,
                    help="steps of the secondary full-training-step "
                         "measurement (0 disables it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_sets(args, rank, dev, nsets=None, first=0):
    """--sets independent (state0, actions, ref) triples in the chosen layout.
    ref is packed [pos, vel] (6 columns) for SoA - the 608 B/trajectory
    algorithmic layout - and the reference's 9-column rows for AoS."""
    from apg_trajectory_tracking_amd import synthetic
    sets = []
    for i in range(first, first + (nsets or args.sets)):
        d = synthetic.quad_polynomial_batch(
            args.batch, args.horizon, args.dt,
            seed=args.seed + 1000 * i + rank)
        ref6 = torch.cat((d["ref"][:, :, :3], d["ref"][:, :, 6:9]), 2)
        if args.layout == "packed":
            t = (synthetic.to_packed_state(d["state0"]),
                 synthetic.to_packed_seq(d["actions"]),
                 synthetic.to_packed_seq(ref6))
        elif args.layout == "soa":
            t = (synthetic.to_soa_state(d["state0"]),
                 synthetic.to_soa_seq(d["actions"]), synthetic.to_soa_seq(ref6))
        else:
            t = (d["state0"], d["actions"], d["ref"])
        sets.append(tuple(x.to(dev) for x in t))
    return sets


def cpu_baseline(args, gpu_set0=None):
    """The reference's CPU PyTorch autograd path (restated in
    oracle/torch_port.py, pinned to the reference by tests/golden) timed on
    this box's host cores on the same workload shape.  The intra-op thread
    count is chosen by a short sweep (oversubscribing a 128-core host with
    ~600 tiny eager ops per iteration is several times slower than 16-32
    threads); the best setting is then timed on a bounded sample."""
    from apg_trajectory_tracking_amd import synthetic
    from oracle import torch_port as tp
    d = synthetic.quad_polynomial_batch(args.batch, args.horizon, args.dt,
                                        seed=args.seed)
    dyn = tp.QuadOracle()
    run = lambda: tp.rollout_fwd_bwd(dyn, tp.quad_mpc_loss, d["state0"],
                                     d["actions"], d["ref"], args.dt)
    # parity of the very launches that were timed: buffer set 0 of rank 0 is
    # this seed's batch - the GPU's loss and dL/dactions of it against the CPU
    # port's, on the same tensors (north_star: 1e-4 relative)
    parity = None
    if gpu_set0 is not None:
        _, c_loss, c_ga, _ = run()
        g_ga = gpu_set0["grad_actions"].double()
        c_ga = c_ga.double()
        per = ((g_ga - c_ga).abs().flatten(1).amax(1)
               / c_ga.abs().flatten(1).amax(1).clamp_min(1e-30))
        parity = {
            "what": "GPU (the timed launch of buffer set 0: quad_rollout_rows_kernel"
                    "<10,false> when packed) vs the CPU port on the same tensors",
            "loss_gpu": gpu_set0["loss"], "loss_cpu": float(c_loss),
            "loss_rel_err": abs(gpu_set0["loss"] - float(c_loss)) / abs(float(c_loss)),
            "grad_actions_rel_err": float((g_ga - c_ga).abs().max() / c_ga.abs().max()),
            "grad_actions_worst_trajectory_rel_err": float(per.max()),
            "tolerance": 1e-4,
        }
        parity["ok"] = bool(parity["loss_rel_err"] < 1e-4
                            and parity["grad_actions_rel_err"] < 1e-4)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    sweep = {}
    for t in sorted({min(t, ncpu) for t in (8, 16, 32, 64, default_threads)}):
        torch.set_num_threads(t)
        run()
        t0 = time.perf_counter()
        run()
        sweep[t] = time.perf_counter() - t0
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    one = sweep[threads]
    iters = max(3, min(50, int(args.cpu_seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    el = time.perf_counter() - t0
    # the reference turns autograd anomaly detection on at import
    # (neural_control/drone_loss.py:6): its as-shipped path pays for it
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.autograd.detect_anomaly(check_nan=True):
            run()
            t0 = time.perf_counter()
            for _ in range(2):
                run()
            anomaly_ms = (time.perf_counter() - t0) / 2 * 1e3
    torch.set_num_threads(default_threads)
    # second CPU data point: the C oracle (matrix-form restatement with a
    # hand-written reverse sweep, OpenMP over the batch, all host cores)
    c_port = None
    try:
        from oracle import c_oracle as co
        s0, a, r = (d["state0"].numpy(), d["actions"].numpy(), d["ref"].numpy())
        co.quad_rollout_fwd_bwd(s0, a, r, args.dt, want_states=False)
        t0 = time.perf_counter()
        n = 0
        while n < 5 or time.perf_counter() - t0 < 2.0:
            co.quad_rollout_fwd_bwd(s0, a, r, args.dt, want_states=False)
            n += 1
        c_el = time.perf_counter() - t0
        c_port = {"value": args.batch * args.horizon * n / c_el,
                  "unit": "env-steps/s", "cores": ncpu,
                  "what": "oracle/apg_oracle.c fp32, OpenMP, hand adjoint",
                  "ms_per_iter": c_el / n * 1e3}
    except Exception as e:  # the C oracle is optional equipment
        c_port = {"error": repr(e)}
    # config 4 (fixed wing, B = 131 072, H = 20) on the host cores: the C
    # restatement with its hand reverse sweep (oracle/apg_oracle_wing.c)
    c_wing = None
    try:
        from oracle import c_oracle as co
        wb, wh, wdt = 131072, 20, 0.05
        w = synthetic.wing_batch(wb, wh, wdt, seed=args.seed)
        ws, wa, wr = (w["state0"].numpy(), w["actions"].numpy(), w["ref"].numpy())
        co.wing_rollout_fwd_bwd(ws, wa, wr, wdt, want_states=False)
        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 2.0:
            co.wing_rollout_fwd_bwd(ws, wa, wr, wdt, want_states=False)
            n += 1
        w_el = time.perf_counter() - t0
        c_wing = {"value": wb * wh * n / w_el, "unit": "env-steps/s",
                  "cores": ncpu, "batch": wb, "horizon": wh,
                  "what": "oracle/apg_oracle_wing.c fp32, OpenMP, hand adjoint",
                  "ms_per_iter": w_el / n * 1e3}
    except Exception as e:
        c_wing = {"error": repr(e)}
    return {
        "parity_check": parity,
        "c_oracle": c_port,
        "c_oracle_wing": c_wing,
        # the same two numbers as flat, named fields (all host CPUs, OpenMP)
        "c_oracle_quad_env_steps_per_s": (c_port or {}).get("value"),
        "c_oracle_wing_env_steps_per_s": (c_wing or {}).get("value"),
        "c_oracle_cores": ncpu,
        "host_logical_cpus": ncpu,
        "value": args.batch * args.horizon * iters / el,
        "unit": "env-steps/s",
        # threads of THIS (PyTorch-eager) number: the best of the sweep below,
        # not the size of the host - see host_logical_cpus
        "cores": threads,
        "kind": "port",
        "sample": (f"{iters} iterations of the full B={args.batch} H={args.horizon} "
                   f"rollout fwd+bwd, PyTorch-eager CPU autograd (the reference's "
                   f"op sequence), anomaly mode off, {threads} intra-op threads "
                   f"(best of sweep {sorted(sweep)}) on {ncpu} logical CPUs"),
        "ms_per_iter": el / iters * 1e3,
        "ms_per_iter_anomaly_mode_on": anomaly_ms,
        "thread_sweep_ms": {str(k): v * 1e3 for k, v in sorted(sweep.items())},
    }


# ----------------------------------------------------------------------------
# Training steps (secondary blocks of the line).  Each is the REAL trainer
# method on this rank's shard and carries its own roofline: the step's MFMA
# work (policy sweeps as fp16-split products on v_mfma_f32_32x32x16_f16 and
# the weight-gradient products as six bf16 products per multiply-add on
# v_mfma_f32_16x16x32_bf16, both against the 2.5 PFLOP/s dense 16-bit peak;
# the small conv product still issues fp32 matrix instructions and is counted
# with the others), its algorithmic
# plane traffic against 8 TB/s, and the achieved fraction of the LARGER of the
# two floors.  Per-wave MFMA counts are the kernels' static instruction counts
# (hipcc -S of csrc/mlp_rollout.hip / mlp_concurrent.hip, lstm.hip; one wave = 32 trajectories, one fp16
# MFMA = 32 x 32 x 16 x 2 flop); plane bytes are the planes
# the sweeps write / read once plus one more read by the products
# (measured FETCH / WRITE equal these counts).
FP32_MFMA_PEAK_TFLOPS = 157.3
# the sweeps' layers run as v_mfma_f32_32x32x16_f16 on operands split into two
# fp16 terms (three instructions per k-block of 16, 32 768 flop each, every one
# of them counted): dense fp16 peak of MI355X_MICROARCH.md
FP16_MFMA_PEAK_TFLOPS = 2500.0
STEP_MODELS = {
    # mode: fp16 MFMAs per wave (forward + reverse; per step for the unrolled
    # modes), plane bytes per env-step and per trajectory
    # round 4: weight gradients inside the reverse kernel.  Forward 222; the
    # trajectory-major reverse kernel 474 per wave: head 24 + 18 + 18 (weight
    # blocks, transposed product, chain), fc3 / fc2 24 + 24 + 24 each, fc1 84 +
    # 24 + 12 (its 14 blocks, states_in's cotangent and blocks) + conv 5 x (12 +
    # 18).  (The staged kernel: 150 + 248.)
    # Round 6: the weight blocks' A operands by identity transposition instead of
    # a second, swapped chain: 442 (PMC, SQ_INSTS_MFMA per wave:
    # profiles/r06_pmc_concurrent_step.txt).
    "concurrent": dict(mfma_once=222 + 442, mfma_per_step=0, products_in_sweep=True,
                       # forward writes 431 planes; reverse reads 256 (tanh') +
                       # 521 (x of the products) + 80 (d_zout twice) planes and
                       # writes 30 x 4 KB of partials per 256 trajectories, read
                       # once more by the second stage; inputs 828
                       bytes_per_traj=(431 + 256 + 521 + 80) * 4 + 2 * 30 * 4096 // 256 + 828,
                       bytes_per_step=0,
                       # features 60 + state0 48 + in_ref H*36 + ref H*36
                       algo_bytes_per_traj=60 + 48 + 360 + 360),
    # round 5: weight gradients inside the reverse sweep
    # (mlp_rollout_bwd_tm_kernel).  Forward 198; reverse 571 per wave and step:
    # h3 transposes 8, head 2 + 6 + 6, fc3 / fc2 24 + 24 + 24 + 8 + 2 each (weight
    # blocks, transposed product, chain, tanh' transposes, exponents), fc1 on
    # s1 24 + 24 + 24 + 8 + 2 + states_in 12 + 12, the feature-major conv
    # cotangent 60, five conv blocks of 12 + 12 + 1 + 18.  (Rounds 3-4: 198 +
    # 144 + the nine plane products.)
    # Round 6: ONE chain orientation (identity transposition for head, fc3, fc2,
    # fc1 and states_in; the conv blocks keep the swapped product): 525 per wave
    # and step (PMC: 5 250 per wave over the ten steps, profiles/r06_pmc_ar_step.txt).
    "autoregressive": dict(mfma_once=0, mfma_per_step=198 + 525, products_in_sweep=True,
                           # fwd writes 436 planes + states / actions (1 808 B per
                           # env-step); the reverse sweep reads the 431
                           # activation planes trajectory-major + masks, states,
                           # actions, reference (1 844); the per-workgroup
                           # accumulators (120 KB x 10 read-modify-writes) stay
                           # in the caches; once per trajectory: state0, in_ref
                           # by both sweeps, windows for the conv block
                           bytes_per_step=1808 + 1844,
                           bytes_per_traj=48 + 2 * 20 * 36 + 10 * 36 * 2,
                           # state0 48 + in_ref 2H*36 + ref H*36
                           algo_bytes_per_traj=48 + 720 + 360),
    # Round 6: the gate / head weight gradients from lstm_gate_wgrad_kernel (conv
    # recomputed trajectory-major: 108 fp16 matrix instructions per 32
    # trajectories and step next to the sweeps' 90 + 42); the forward sweep no
    # longer writes the 160 relu(conv) planes.  Per env-step: forward writes
    # features 60 + h / c 64 + gates 128 + h_new 32 + relu mask 20 + states /
    # actions 64; the reverse sweep reads 316 of them and writes the cotangents
    # 144; the weight-gradient kernel reads cotangents 144 + features, h_prev,
    # h_new 124 + position 12.  Per trajectory: inputs (state0, in_ref, ref, h0 /
    # c0) by the three kernels 1 972, the conv cotangents' 720 diagonal planes
    # written and read once 5 760, the window + position planes of the two conv
    # products 840.  (Rounds 2-5: 2 916 per env-step, x alone 700 twice.)
    # (+ lstm_conv_wgrad_kernel: 36 segments x 6 per 32 trajectories, 22 per step)
    "LSTM": dict(mfma_once=0, mfma_per_step=90 + 42 + 108 + 22, products_in_sweep=True,
                 bytes_per_step=368 + 316 + 144 + 280,
                 bytes_per_traj=1972 + 5760 + 840,
                 # state0 48 + in_ref 720 + ref 360 + h0 / c0 64
                 algo_bytes_per_traj=48 + 720 + 360 + 64),
}


def step_roofline(mode, B, H, n_params, ms):
    m = STEP_MODELS[mode]
    waves = (B + 31) // 32
    sweep_flops = (m["mfma_once"] + m["mfma_per_step"] * H) * 32768.0 * waves
    cols = B * (H if m["mfma_per_step"] else 1)      # columns of the products
    # three-term bf16 operands, six products per multiply-add (planes_gemm.hip);
    # none when the products are part of the sweep's own instruction count
    product_flops = 0.0 if m.get("products_in_sweep") else 6 * 2.0 * n_params * cols
    nbytes = float(B) * (m["bytes_per_traj"] + m["bytes_per_step"] * H)
    mfma_ms = (sweep_flops + product_flops) / (FP16_MFMA_PEAK_TFLOPS * 1e12) * 1e3
    hbm_ms = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
    floor = max(mfma_ms, hbm_ms)
    # the step's ALGORITHMIC inputs: what a step must read whatever its design
    # (per trajectory: features 60 + state0 48 + the reference rows the policy
    # and the loss read) + the parameters in and their gradients out
    algo = float(B) * m["algo_bytes_per_traj"] + 2.0 * 4 * n_params
    return {
        # the plane bytes are this design's own traffic - the
        # fractions that do not grade the step against itself
        "frac_vs_mfma_floor": mfma_ms / ms,
        "plane_bytes_over_algorithmic": nbytes / algo,
        "algorithmic_bytes_per_step": algo,
        "frac_vs_algorithmic_bytes": algo / (HBM_PEAK_GBS * 1e9) * 1e3 / ms,
        "bound": "mfma" if mfma_ms >= hbm_ms else "hbm",
        "mfma": {"sweep_fp16_flops_per_step": sweep_flops,
                 "product_bf16x3_flops_per_step": product_flops,
                 "peak_TFLOPs": {"fp16": FP16_MFMA_PEAK_TFLOPS,
                                 "fp32": FP32_MFMA_PEAK_TFLOPS},
                 "floor_ms": mfma_ms},
        "hbm": {"plane_bytes_per_step": nbytes, "peak_GBps": HBM_PEAK_GBS,
                "floor_ms": hbm_ms, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9},
        "frac": floor / ms,
        "what": "larger of ((sweep fp16-MFMA flops + product bf16-MFMA flops) / 2.5 PF, "
                "plane bytes / 8 TB/s) over the measured step; the sweeps' layers "
                "are fp16-split products (csrc/policy_mfma16.h), the weight-gradient "
                "products three-term bf16 splits, six products per multiply-add "
                "(csrc/planes_gemm.hip) - both at fp32 accuracy",
    }


def timed_steps(step, n, dist, ramp_ms=40.0):
    """(ms per step, last output, per-chunk ms) of `n` calls, each chunk
    bracketed by barrier + synchronize (so every rank sees the slowest rank).
    The calls are timed in four chunks and the MEDIAN chunk is reported: one
    host or driver stall inside a region (seen once: 60 ms in an LSTM block,
    3.5 ms "per step") would otherwise be the number; every chunk's mean is in
    the line next to it.  Chunks of 100 steps by default (--train-steps 400): a
    chunk pays one pipeline fill and one synchronize, ~50 us - at 10 steps per
    chunk that was 4 % of the concurrent step.  `ramp_ms` of untimed steps run
    right before every chunk's barrier + synchronize (round 6): a chunk starts
    on a device at its working clocks, as a step inside an epoch does."""
    for _ in range(3):
        out = step()
    # how many steps `ramp_ms` are: from ten timed ones, and the SAME count on
    # every rank (the largest) - a step of a multi-rank run holds an all-reduce,
    # ranks that ran different numbers of them would wait for each other forever
    torch.cuda.synchronize()
    r0 = time.perf_counter()
    for _ in range(10):
        out = step()
    torch.cuda.synchronize()
    ramp = int(ramp_ms * 1e-3 / max((time.perf_counter() - r0) / 10, 1e-6)) + 1
    if dist is not None:
        agreed = torch.tensor([ramp], dtype=torch.int64, device="cuda")
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
        ramp = int(agreed.item())
    ramp = min(ramp, 4 * max(n, 100))
    chunks = []
    sizes = [n // 4 + (1 if i < n % 4 else 0) for i in range(4)]
    import gc
    for m in [c for c in sizes if c > 0]:
        # (a full collection of the interpreter's garbage - tens of ms with torch
        # loaded - is taken HERE, not wherever its allocation count happens to
        # trip inside a timed chunk: round 4's 0.75 ms outlier chunk)
        gc.collect()
        # (the collection leaves the device idle for tens of ms and it clocks
        # down: the first steps behind it ran slow enough to cost a 100-step
        # chunk 2.3 ms - 6 % of the LSTM step; the ramp is 12-20 ms long,
        # profiles/r06_step_ramp.txt.  Untimed steps take it; the synchronize
        # below is microseconds)
        for _ in range(ramp):
            out = step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(m):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        chunks.append((time.perf_counter() - t0) / m * 1e3)
    return sorted(chunks)[(len(chunks) - 1) // 2], out, chunks


def chunk_stats(chunks):
    """mean and max next to the median chunk (an outlier
    chunk - a host or driver stall - is visible, not dropped)."""
    return {"ms_per_step_mean": sum(chunks) / len(chunks), "ms_per_step_max": max(chunks)}


def trainer_step_probe(args, dev, dyn, dist, mode):
    """One training step per rank through the REAL trainer method
    (TrainDrone): policy inside the fused kernels (matrix cores), weight-
    gradient products, ONE in-place all-reduce(sum) of the flat gradient buffer
    + loss slot when world > 1 (RCCL over xGMI), momentum SGD.
      concurrent      BASELINE configs[1] as a full step (train_concurrent_fused)
      autoregressive  configs[2] shape: 65 536 per GPU (train_recurrent_model)
      LSTM            configs[4] (train_recurrent_model, carried state in-kernel)
      packed          configs[1] with a policy that is NOT the reference
                      architecture, through run_epoch's row-layout path
                      (train_controller_packed -> quad_rollout_rows_kernel)"""
    from apg_trajectory_tracking_amd import synthetic
    from apg_trajectory_tracking_amd.dataset import state_preprocessing
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    from apg_trajectory_tracking_amd.models.rnn import LSTM_NEW
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    H, B = args.horizon, args.batch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    tmode = "concurrent" if mode == "packed" else mode
    cfg = dict(delta_t=args.dt, horizon=H, batch_size=B * world, ref_dim=9,
               action_dim=4, train_mode=tmode,
               learning_rate_controller=1e-9, system="quad")
    t = TrainDrone(dyn, dyn, cfg)
    torch.manual_seed(4321 + rank)       # init_optimizer broadcasts rank 0's
    if mode == "LSTM":
        t.net = LSTM_NEW(15, H, 9, 4, conv=1).to(dev)
    elif mode == "packed":
        t.net = PlainPolicy(H).to(dev)
    else:
        t.net = Net(15, H, 9, 4 * H if mode == "concurrent" else 4, conv=1).to(dev)
    d = synthetic.quad_polynomial_batch(B, H, args.dt, seed=args.seed + rank,
                                        ref_length=t.ref_length)

    class Shard:     # this rank's shard, resident on the device
        states, in_ref_states, ref_states = (
            d[k].to(dev) for k in ("state0", "in_ref", "ref"))
    with torch.no_grad():
        Shard.normed_states = state_preprocessing(Shard.states)
    t.state_data = Shard
    t.static_shard = True        # every step is on this resident shard:
    # plane copies kept, the step replayed from captured graphs - with more
    # than one rank as two graphs around the eager all-reduce (_GraphedStep),
    # i.e. the N > 1 step replays the same kernels as the N = 1 step
    t.graph_steps = True
    t.init_optimizer()           # (swaps torch.nn.Linear leaves: swap_linear)
    if mode == "packed":
        ref6 = torch.cat((d["ref"][:, :, :3], d["ref"][:, :, 6:9]), 2)
        rows = (synthetic.to_packed_state(d["state0"]).to(dev),
                synthetic.to_packed_seq(ref6).to(dev))
        step = lambda: t.train_controller_packed(
            Shard.normed_states, Shard.in_ref_states, *rows)
    elif mode == "concurrent":
        step = lambda: t.train_concurrent_fused(
            Shard.normed_states, Shard.states, Shard.in_ref_states, Shard.ref_states)
    else:
        step = lambda: t.train_recurrent_model(
            None, Shard.states, Shard.in_ref_states, Shard.ref_states)
    # the loss of a graphed step is taken as run_epoch's loops take it: the
    # captured output buffer, valid until the next step (`borrow_loss`); a
    # private copy per step is one more launch behind every replay
    # (`ms_per_step_private_loss`)
    t.borrow_loss = True
    ms, total, chunks = timed_steps(step, args.train_steps, dist)
    total = total.clone()
    n_params = sum(p.numel() for p in t.net.parameters() if p.requires_grad)
    out = {
        "ms_per_step": ms,
        **chunk_stats(chunks),
        "ms_per_step_chunks": chunks,   # four timed chunks; ms_per_step = their median
        "env_steps_per_s": world * B * H / (ms * 1e-3),
        "batch_per_gpu": B, "global_batch": world * B,
        "allreduce_floats": n_params + 1,
        "global_loss": float(total.item()),
        "launch": ("one captured HIP graph per step" if world == 1 else
                   "two captured HIP graphs per step around the eager all-reduce"),
    }
    default_graph = bool(t._graphs) and not any(
        getattr(g, "planned", False) for g in t._graphs.values())
    if any(getattr(g, "planned", False) for g in t._graphs.values()):
        # (single-process concurrent step: buffers and argument structs made
        # once, one library call = five launches per step, no capture)
        out["launch"] = "step plan: one library call per step, no graph"
    elif not t._graphs:
        # (TrainBase.launch_form: measured at the first capture of the mode)
        out["launch"] = "kernels launched in stream order (measured faster than graph replays)"
    # graph replay or stream order: measured by the trainer at its first capture
    # of this mode (TrainBase._measure_launch_form), not a constant
    out["launch_form"] = [dict(r) for r in t.results_dict.get("launch_form", [])]
    if mode == "concurrent":
        # one process: the optimizer's update is applied by the second stage of
        # the step itself (apg_quad_mlp_concurrent_train_step); the split form
        # below - what N > 1 runs - keeps optimizer.step() behind the all-reduce
        out["optimizer_update"] = ("inside the step's second-stage kernel"
                                   if t._in_kernel_update() is not None
                                   else "optimizer.step() (fused torch SGD)")
    # like-for-like companions: the same step launched eagerly
    # (no graphs: what a multi-rank step was before round 4) and, on one rank,
    # in the split form the N > 1 step uses (graph A, empty all-reduce slot,
    # graph B) - a scaling curve compares `ms_per_step` at N with
    # `ms_per_step_split_graph` at 1
    if world > 1:
        # the same N-rank step with its collective switched off (split graphs,
        # message and optimizer slot unchanged; all GPUs busy at once): what the
        # all-reduce costs THIS step on THIS node
        from apg_trajectory_tracking_amd import parallel
        try:
            with parallel.collectives_suspended():
                out["ms_per_step_no_collective"], _, _ = timed_steps(
                    step, max(8, args.train_steps // 2), dist)
            parallel.broadcast_module(t.net)  # (the replicas drifted: lr 1e-9, but still)
            out["parallel_efficiency"] = out["ms_per_step_no_collective"] / ms
            out["parallel_efficiency_what"] = (
                "N-rank step without its all-reduce / N-rank step, same ranks, same "
                "process; <= 1.  The driver computes the curve's own efficiency from "
                "`value` at N = 1, 2, 4, 8")
        except Exception as e:        # (this leg must not cost the block its numbers)
            out["parallel_efficiency"] = {"error": repr(e)}
    t.borrow_loss = False
    out["ms_per_step_private_loss"], _, _ = timed_steps(
        step, max(8, args.train_steps // 2), dist)
    t.borrow_loss = True
    chosen = dict(t.launch_form)
    t.measure_launch_form = False
    if world == 1 and not default_graph:
        # the same step replayed from ONE captured graph (rounds 3-4's form)
        t.launch_form, t.plan_steps = {tmode: "graph"}, False
        t._graphs.clear()
        out["ms_per_step_single_graph"], _, _ = timed_steps(
            step, max(8, args.train_steps // 2), dist)
        t.launch_form, t.plan_steps = chosen, True
        t._graphs.clear()
    t.graph_steps = False
    out["ms_per_step_eager"], _, _ = timed_steps(step, max(8, args.train_steps // 2), dist)
    if world == 1:
        t.launch_form = {tmode: "graph"}
        t.graph_steps, t.split_graph = True, True
        t._graphs.clear()
        out["ms_per_step_split_graph"], _, _ = timed_steps(
            step, max(8, args.train_steps // 2), dist)
        t.split_graph, t.launch_form = None, chosen
        t._graphs.clear()
    t.graph_steps = True
    # how close the default is to the best launch form of this box
    forms = [out[k] for k in ("ms_per_step", "ms_per_step_single_graph", "ms_per_step_eager")
             if k in out]
    out["default_over_best_form"] = out["ms_per_step"] / min(forms)
    if mode == "packed":
        out["what"] = ("TrainDrone.train_controller_packed: an arbitrary PyTorch policy "
                       "(stock torch.nn.Linear layers, switched in place to the library's "
                       "weight gradient by init_optimizer) -> [H, B, 4] action rows -> "
                       "quad_rollout_rows_kernel -> autograd backward + SGD")
        out["linear_layers"] = sorted({type(m).__module__ + "." + type(m).__name__
                                       for m in t.net.modules()
                                       if isinstance(m, torch.nn.Linear)})
        # opt-out comparison: the same policy with rocBLAS weight gradients
        t2 = TrainDrone(dyn, dyn, cfg)
        t2.swap_linear = False
        t2.net = PlainPolicy(H).to(dev)
        t2.state_data, t2.static_shard, t2.graph_steps = Shard, True, True
        t2.init_optimizer()
        step2 = lambda: t2.train_controller_packed(
            Shard.normed_states, Shard.in_ref_states, *rows)
        ms2, _, chunks2 = timed_steps(step2, args.train_steps, dist)
        out["with_stock_torch_linear"] = {
            "ms_per_step": ms2, **chunk_stats(chunks2), "ms_per_step_chunks": chunks2,
            "what": "swap_linear = False: dW = dY^T X through rocBLAS"}
    else:
        fused = {"concurrent": t.train_concurrent_fused(None, None, None, None, probe=True),
                 "autoregressive": t.fused_policy and t._fusable_mlp(),
                 "LSTM": t.fused_policy and t._fusable()}[mode]
        out["fused"] = bool(fused)
        out["what"] = {
            "concurrent": "TrainDrone.train_concurrent_fused",
            "autoregressive": "TrainDrone.train_recurrent_model (autoregressive)",
            "LSTM": "TrainDrone.train_recurrent_model (LSTM)"}[mode] + (
            ": policy inside the kernels + weight-gradient products + flat-buffer "
            "all-reduce(sum) + SGD")
        if fused and H == 10:
            out["roofline"] = step_roofline(mode, B, H, n_params, ms)
    if world > 1 and mode == "autoregressive":
        buf = torch.zeros(n_params + 1, device=dev)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        out["allreduce_us_alone"] = (time.perf_counter() - t0) / 50 * 1e6
    return out


def run_epoch_probe(args, dev, dyn):
    """what TrainBase.run_epoch runs (scripts/train_base.py:
    188-218), timed as it is: a resident data set of 32 x B trajectories,
    shuffled index batches of B = 65 536 (device-side permutation, the gather
    folded into the fused step's layout change), every step through the real
    trainer method, the loss accumulated on the device and read back once per
    epoch (the reference's `loss.item()` per batch became one `.item()` per
    epoch; it is still a synchronisation per epoch, amortised over the epoch's
    32 batches here - the reference's own configuration has 62 per epoch).
    Eager launches, and the default since round 4: graphs - the first epoch
    per-step graphs, from the second on ONE graph per epoch (`graph_epochs`:
    a fresh permutation is copied into the buffer the captured gathers read;
    in the concurrent mode the next batch's gather runs behind the reverse
    kernel of the current one).  ms per BATCH, host clock around whole
    epochs."""
    import contextlib
    from apg_trajectory_tracking_amd.train_drone import TrainDrone
    H, B, nb = args.horizon, args.batch, 32
    out = {"batches_per_epoch": nb, "batch": B,
           "what": "TrainDrone.run_epoch('controller'): shuffled index batches, "
                   "fused step per batch, one loss read-back per epoch"}
    for key, mode in (("concurrent", "concurrent"), ("ar", "autoregressive"),
                      ("lstm", "LSTM")):
        cfg = dict(delta_t=args.dt, delta_t_train=args.dt, epoch_size=nb * B,
                   self_play=0, batch_size=B, state_size=12, horizon=H,
                   train_mode=mode, ref_dim=9, action_dim=4,
                   learning_rate_controller=1e-9, system="quad", modified_params={},
                   save_name="bench_run_epoch")
        res = {}
        try:
            with contextlib.redirect_stdout(sys.stderr):
                t = TrainDrone(dyn, dyn, cfg)
                t.initialize_model(device=dev, seed=args.seed)
                for graphed in (False, True):
                    t.graph_steps = graphed
                    t._graphs.clear()
                    t.run_epoch("controller", 0)          # warm-up (eager epoch)
                    t.run_epoch("controller", 0)          # capture of the epoch graph
                    # (the capture left the device idle; three epochs of the
                    # concurrent mode are 12 ms, the length of the clock ramp
                    # behind an idle device - profiles/r06_step_ramp.txt: 40 ms
                    # of untimed epochs first, then >= 100 ms of timed ones)
                    torch.cuda.synchronize()
                    r0, warm = time.perf_counter(), 0
                    while time.perf_counter() - r0 < 0.04:
                        t.run_epoch("controller", 0)
                        warm += 1
                    torch.cuda.synchronize()
                    per_epoch = (time.perf_counter() - r0) / warm
                    epochs = max(3, int(0.1 / per_epoch + 0.999))
                    res["epochs_timed" if graphed else "epochs_timed_eager"] = epochs
                    t0 = time.perf_counter()
                    for e in range(epochs):
                        t.run_epoch("controller", e + 1)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / (epochs * nb) * 1e3
                    res["ms_per_batch" if graphed else "ms_per_batch_eager"] = ms
                res["env_steps_per_s"] = B * H / (res["ms_per_batch"] * 1e-3)
                res["graphs"] = sorted(str(k) for k in t._graphs)
                res["epoch_graphs"] = sorted(
                    str(k) for k, v in t._epoch_graphs.items() if v.get("graph") is not None)
            del t
        except Exception as e:      # secondary block
            res = {"error": repr(e)}
        torch.cuda.empty_cache()
        out[key] = res
    return out


class PlainPolicy(torch.nn.Module):
    """A policy that is not the reference architecture (for `train_step_packed`)."""

    def __init__(self, horizon):
        super().__init__()
        self.a = torch.nn.Linear(15 + horizon * 9, 64)
        self.b = torch.nn.Linear(64, 64)
        self.c = torch.nn.Linear(64, 4 * horizon)

    def forward(self, state, ref):
        x = torch.cat((state, ref.flatten(1)), 1)
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def measured_copy_bandwidth(dev):
    """the copy bandwidth this box delivers, next to the 8 TB/s
    datasheet peak.  apg_stream_copy (16-byte accesses, non-temporal stores -
    the rollouts' access pattern without arithmetic) over 1 GiB -> 1 GiB (far
    beyond the 256 MiB Infinity Cache), HIP events on the launch stream;
    GB/s counts bytes read + bytes written."""
    from apg_trajectory_tracking_amd import _capi
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    src.fill_(1)
    st = _capi.stream_of(src)
    run = lambda: _capi.check(_capi.lib().apg_stream_copy(
        src.data_ptr(), dst.data_ptr(), n, st), "apg_stream_copy")
    for _ in range(3):
        run()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ok = bool((dst[:: 1 << 20] == 1).all())
    # other launch shapes of the same copy (the guide quotes
    # 6.29 TB/s for "a float4 copy" without its shape - is the probe slow or
    # the box?): csrc/common.hip, apg_stream_copy_shape
    shapes = {}
    names = {1: "one float4 per thread, whole-array grid",
             2: "capped grid (8 blocks per CU), four float4 in flight per thread",
             3: "as 2, non-temporal stores", 4: "as 1, 1024-thread blocks"}
    for shape, what in names.items():
        try:
            rs = lambda: _capi.check(_capi.lib().apg_stream_copy_shape(
                src.data_ptr(), dst.data_ptr(), n, shape, st), "apg_stream_copy_shape")
            for _ in range(2):
                rs()
            e0.record()
            for _ in range(reps):
                rs()
            e1.record()
            torch.cuda.synchronize()
            shapes[what] = 2 * n / (e0.elapsed_time(e1) / reps * 1e-3) / 1e9
        except Exception as e:
            shapes[what] = repr(e)
    del src, dst
    torch.cuda.empty_cache()
    best = max([2 * n / (ms * 1e-3) / 1e9] + [v for v in shapes.values()
                                              if isinstance(v, float)])
    return {"GBps": 2 * n / (ms * 1e-3) / 1e9, "bytes_each_way": n, "ms": ms,
            "verified": ok, "other_shapes_GBps": shapes, "best_GBps": best,
            "guide_GBps": 6290.0,
            "what": "apg_stream_copy 1 GiB -> 1 GiB, read + write bytes per second; "
                    "guide_GBps: MI355X_MICROARCH.md's 'float4 copy' figure (shape not "
                    "stated there)"}


def stream_floor_probe(args, dev, nsets):
    """the headline launch's bytes - 28 input rows + 10 output
    rows of B x 16 B - moved without arithmetic, in the fastest copy shape of this
    GPU (one float4 per thread, whole-array grid; csrc/common.hip
    `stream_rows_probe_kernel`, six store placements) AND in the rollout's own
    access pattern (`stream_rows_lane_kernel`: a trajectory per lane), under
    the headline's own protocol: `nsets` rotating buffer sets, 2 000 launches
    replayed from a captured graph, one HIP-event pair on the launch stream.  No arithmetic and no trajectory
    structure: a floor for the rollout kernel's launch, not a model of it."""
    from apg_trajectory_tracking_amd import _capi
    H, B = args.horizon, args.batch
    in_bytes = B * (48 + 16 * H + 24 * H)          # state0 + actions + ref(pos, vel)
    out_bytes = B * 16 * H                         # dL/dactions
    ins = [torch.randn(in_bytes // 4, device=dev) for _ in range(nsets)]
    outs = [torch.empty(out_bytes // 4, device=dev) for _ in range(nsets)]
    lib = _capi.lib()
    names = {1: "first 10/28 of the threads store, plain", 2: "first 10/28 store, nt",
             3: "stores spread between the loads, plain", 4: "stores spread, nt",
             5: "as 1, one wave per workgroup", 6: "as 2, one wave per workgroup",
             7: "the rollout's own pattern: a trajectory per lane, 28 row loads + "
                "10 nt row stores per lane, one wave per workgroup"}
    res = {}
    per_graph, replays = 400, 5
    n = per_graph * replays
    side = torch.cuda.Stream(device=dev)
    for shape, what in names.items():
        def run(i):
            _capi.check(lib.apg_stream_rows_probe(
                ins[i % nsets].data_ptr(), in_bytes, outs[i % nsets].data_ptr(), out_bytes,
                shape, torch.cuda.current_stream(dev).cuda_stream), "apg_stream_rows_probe")
        try:
            with torch.cuda.stream(side):
                for i in range(40):
                    run(i)
        except ValueError:       # (shape 7 is the H = 10 row counts only)
            continue
        torch.cuda.synchronize()
        # as the headline: the launches replayed from ONE captured graph (a Python /
        # ctypes launch costs about what this kernel takes - the host must not pace it)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(per_graph):
                run(i)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            g.replay()
            e0.record()
            for _ in range(replays):
                g.replay()
            e1.record()
        torch.cuda.synchronize()
        res[what] = e0.elapsed_time(e1) / n * 1e3
        del g
    st = torch.cuda.current_stream(dev).cuda_stream
    # the stores landed where they should (shape 1: out == the head of in)
    _capi.check(lib.apg_stream_rows_probe(ins[0].data_ptr(), in_bytes, outs[0].data_ptr(),
                                          out_bytes, 1, st), "apg_stream_rows_probe")
    ok = bool(torch.equal(outs[0], ins[0][:out_bytes // 4]))
    best = min(res.values())
    return {"us_by_shape": res, "best_us": best, "verified": ok,
            "in_bytes": in_bytes, "out_bytes": out_bytes, "buffer_sets": nsets,
            "launches": n, "GBps_best": (in_bytes + out_bytes) / best / 1e3,
            "what": "apg_stream_rows_probe: the headline launch's algorithmic bytes, one "
                    "float4 per thread over a whole-array grid, 400 launches per captured graph x 5 "
                    "replays, HIP events on the launch stream"}


KERNEL_SOURCES = ("quad.hip", "quad_math.h", "apg_device.h")
WING_SOURCES = ("wing.hip", "wing_math.h", "apg_device.h")
FP32_VALU_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2   # 157.3 TFLOP/s spec = one
# wave64 fma per SIMD every 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)


def kernel_build_id(sources=KERNEL_SOURCES):
    """sha256 of the sources the dominant kernel is compiled from: PMC numbers
    measured on another build of the kernel must not be reported for this one."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "apg_trajectory_tracking_amd", "csrc")
    for name in sources:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_pmc_traffic(args):
    """(L2 <-> fabric bytes per launch, note): the committed rocprofv3 PMC pass
    of this very command (profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE,
    calibrated) - only if it was taken on THIS build of the kernel and on this
    buffer-set protocol.  These counters sit on the L2's memory side, so
    Infinity-Cache hits are counted too (MI355X_MICROARCH.md §HBM): the number
    shows over-fetch (ratio to the algorithmic bytes), not where the bytes
    came from - the buffer-set protocol takes care of that."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    key = f"quad_B{args.batch}_H{args.horizon}_{args.layout}"
    try:
        with open(path) as f:
            entry = json.load(f).get(key)
    except (OSError, ValueError):
        entry = None
    if not entry:
        return None, f"no PMC entry {key} in profiles/pmc_traffic.json"
    build = kernel_build_id()
    if entry.get("kernel_build") != build:
        return None, (f"PMC entry is for kernel build {entry.get('kernel_build')}, "
                      f"this is {build}: stale, not reported")
    return (entry.get("l2_fabric_bytes_per_launch"),
            f"PMC pass on kernel build {build}, {entry.get('buffer_sets')} buffer sets")


def wing_secondary(args, dev):
    """BASELINE configs[3] (fixed wing, H = 20, B = 131 072) is
    VALU-bound, so BOTH fractions are reported: algorithmic HBM bytes (928 B /
    trajectory) per second against 8 TB/s, and issued fp32 VALU
    wave-instructions per second against the vector peak (the instruction
    count per launch is a PMC figure, SQ_INSTS_VALU, committed in
    profiles/pmc_traffic.json and only used for the build it was taken on)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    B, H, dt = 131072, 20, 0.05
    dyn = FixedWingDynamics()
    plans = []
    for i in range(4):
        d = synthetic.wing_batch(B, H, dt, seed=args.seed + i)
        plans.append(F.RolloutPlan(
            "wing", synthetic.to_soa_state(d["state0"]).to(dev),
            synthetic.to_soa_seq(d["actions"]).to(dev),
            synthetic.to_soa_seq(d["ref"]).to(dev), dt, dyn.params,
            layout="soa", loss_mode="none"))
    for i in range(8):
        plans[i % 4].launch()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    n = 60
    e0.record()
    for i in range(n):
        plans[i % 4].launch()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / n * 1e-3
    algo = B * (48 + 16 * H + 12 * H + 16 * H)
    out = {"workload": "fixed-wing concurrent rollout fwd+bwd, BASELINE configs[3]",
           "batch": B, "horizon": H, "us_per_launch": sec * 1e6,
           "env_steps_per_s": B * H / sec,
           "hbm": {"algorithmic_bytes_per_launch": algo,
                   "achieved_GBps": algo / sec / 1e9,
                   "frac": algo / sec / 1e9 / HBM_PEAK_GBS},
           "fp32_valu": None}
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            entry = json.load(f).get(f"wing_B{B}_H{H}_soa")
    except (OSError, ValueError):
        entry = None
    build = kernel_build_id(WING_SOURCES)
    # flop-based roofline next to the instruction-count one:
    # USEFUL fp32 flops = what one evaluation of the step, its adjoint (which
    # contains a second evaluation) and the loss need, the checkpoint scheme's
    # re-integration excluded.  Counted on the scalar arithmetic of
    # csrc/wing_math.h (one evaluation of state_dot ~190 VALU
    # ops, the adjoint ~300, 815 per env-step with 0.75 re-integrations): 680
    # ops per env-step at 1.4 flop per op (40 % of them fused multiply-adds)
    useful = 680 * 1.4 * B * H
    out["fp32_flops"] = {
        "useful_flops_per_launch": useful,
        "achieved_TFLOPs": useful / sec / 1e12,
        "peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS,
        "frac": useful / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS,
        "what": "680 useful scalar ops per env-step (evaluation + adjoint incl. its own "
                "evaluation, no re-integration) x 1.4 flop per op over the 157.3 TFLOP/s "
                "fp32 vector peak; the shader clock under this kernel is 2.25 GHz "
                "(profiles/r04_wing_clock.jsonl)"}
    if entry and entry.get("kernel_build") == build:
        n_valu = entry["valu_wave_instructions_per_launch"]
        out["fp32_valu"] = {
            "wave_instructions_per_launch": n_valu,
            "achieved_wave_instr_per_s": n_valu / sec,
            "peak_wave_instr_per_s": FP32_VALU_WAVE_INSTR_PER_S,
            "frac": n_valu / sec / FP32_VALU_WAVE_INSTR_PER_S,
            "frac_of_single_issue_rate": n_valu / sec / (FP32_VALU_WAVE_INSTR_PER_S / 2),
            "kernel_build": build}
    else:
        out["fp32_valu_note"] = f"no PMC entry for wing kernel build {build}"
    return out


def wing_eval_secondary(args, dev):
    """Beyond SURVEY §8: the batched fixed-wing closed-loop evaluation
    (FixedWingEvaluator.run_eval's flights in ONE launch of
    apg_wing_mlp_closed_loop) with a random-init Net(9, 1, 3, 40) - an
    untrained controller diverges and is put back on its line again and again
    (the training-time branch), every flight runs until it passes x = 50 m or
    uses up max_steps.  Informational: policy-in-the-loop steps per second."""
    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dataset import SyntheticWingDataset
    from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (
        FixedWingDynamics)
    from apg_trajectory_tracking_amd.models.hutter_model import Net
    B, T = 16384, 200
    torch.manual_seed(args.seed)
    net = Net(9, 1, 3, 40, conv=False).to(dev)
    g = torch.Generator().manual_seed(args.seed)
    targets = torch.zeros(B, 1, 3)
    targets[:, 0, 0] = 50.0
    targets[:, 0, 1:] = torch.rand(B, 2, generator=g) * 10 - 5
    targets = targets.to(dev)
    kw = dict(data_dt=0.05, data_horizon=10, max_steps=T, thresh_div=4.0,
              thresh_stable=0.4, test_time=0)
    mean, std = SyntheticWingDataset.MEAN, SyntheticWingDataset.STD
    p = FixedWingDynamics().params
    out = F.wing_mlp_closed_loop(net, targets, 0.05, p, mean, std, **kw)
    torch.cuda.synchronize()
    steps = int(out["steps"].sum())
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        F.wing_mlp_closed_loop(net, targets, 0.05, p, mean, std, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return {"workload": "fixed-wing closed-loop evaluation, random-init controller",
            "flights": B, "max_steps": T, "closed_loop_steps": steps,
            "ms_per_launch": ms, "closed_loop_steps_per_s": steps / ms * 1e3}


# ----------------------------------------------------------------------------
# Rank plumbing shared by the real run and --dry-run-cpu
def _event_ms(fn, n, warm=3, reps=3):
    """ms per call of `fn`: the median of `reps` event-timed runs of `n` calls.
    The interpreter's garbage is collected BEFORE the runs and not during them: a
    full collection (~70 ms with torch loaded) that trips inside a 30-call run of a
    host-launched step is 2 ms "per step" (round 6's last full run: the fixed-wing
    step 2.58 ms instead of 0.37; `timed_steps` has collected up front since
    round 5)."""
    import gc
    for _ in range(warm):
        fn()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        runs = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            runs.append(e0.elapsed_time(e1) / n)
    finally:
        if was_enabled:
            gc.enable()
    return sorted(runs)[len(runs) // 2]


def more_secondaries(args, dev):
    """The README's other headline rows in the driver's record (VERDICT r4 weak
    #8): the fixed-wing TRAINING step (configs[3] with the policy on the matrix
    cores), the controller phase through LearntDynamics (N3) and the batched
    quadrotor closed-loop evaluation (N2)."""
    from apg_trajectory_tracking_amd import functional as F, synthetic
    out = {}
    try:
        from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import FixedWingDynamics
        from apg_trajectory_tracking_amd.train_fixed_wing import TrainFixedWing
        B, H, dt = 131072, 20, 0.05
        cfg = dict(delta_t=dt, delta_t_train=dt, epoch_size=B, self_play=0, batch_size=B,
                   state_size=12, horizon=H, ref_dim=3, action_dim=4,
                   learning_rate_controller=1e-9, system="fixed_wing", modified_params={})
        wdyn = FixedWingDynamics()
        tw = TrainFixedWing(wdyn, wdyn, cfg)
        tw.initialize_model(device=dev, seed=0)
        dw = tw.state_data
        ms = _event_ms(lambda: tw.train_concurrent_fused(
            dw.normed_states, dw.states, dw.in_ref_states, dw.ref_states), 30)
        out["wing_train_step"] = {
            "ms_per_step": ms, "batch": B, "horizon": H,
            "env_steps_per_s": B * H / (ms * 1e-3),
            "what": "TrainFixedWing.train_concurrent_fused: policy forward / reverse on "
                    "the matrix cores around the fused rollout, weight products, SGD"}
        del tw, dw
    except Exception as e:
        out["wing_train_step"] = {"error": repr(e)}
    try:
        from apg_trajectory_tracking_amd.dynamics.quad_dynamics_trained import LearntDynamics
        B, H, dt = 65536, 10, 0.1
        dyn = LearntDynamics().to(dev)
        g = torch.Generator().manual_seed(0)
        with torch.no_grad():
            dyn.linear_at.add_(0.05 * torch.randn(4, 4, generator=g).to(dev))
            for lin, sc in ((dyn.linear_state_1, 0.2), (dyn.linear_state_2, 0.02)):
                lin.weight.add_(sc * torch.randn(lin.weight.shape, generator=g).to(dev))
                lin.bias.add_(sc * torch.randn(lin.bias.shape, generator=g).to(dev))
        d = synthetic.quad_polynomial_batch(B, H, dt, seed=3)
        act = torch.rand(B, H, 4, generator=g)
        soa = (synthetic.to_soa_state(d["state0"]).to(dev), synthetic.to_soa_seq(act).to(dev),
               synthetic.to_soa_seq(d["ref"]).to(dev))
        res = F.quad_learnt_rollout_fwd_bwd(dyn, *soa, dt, layout="soa")
        ms = _event_ms(lambda: F.quad_learnt_rollout_fwd_bwd(dyn, *soa, dt, layout="soa",
                                                             out=res), 30)
        out["learnt_controller_phase"] = {
            "ms_per_launch": ms, "batch": B, "horizon": H,
            "env_steps_per_s": B * H / (ms * 1e-3),
            "what": "quad_learnt_rollout_kernel: H x LearntDynamics.forward (4 x 4 action "
                    "transform, analytic step, 16 -> 64 -> 12 residual network) + "
                    "quad_mpc_loss + backward to the actions, one launch"}
    except Exception as e:
        out["learnt_controller_phase"] = {"error": repr(e)}
    try:
        from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
            FlightmareDynamics)
        from apg_trajectory_tracking_amd.models.hutter_model import Net
        B, L, T = 16384, 300, 251
        torch.manual_seed(2)
        net = Net(15, 10, 9, 40, conv=1).to(dev)
        traj = synthetic.quad_eval_trajectories(B, L, 0.1, seed=5).to(dev)
        qd = FlightmareDynamics()
        ms = _event_ms(lambda: F.quad_mlp_closed_loop(net, traj, 0.1, qd.params, max_steps=T,
                                                      test_time=0), 5, warm=2)
        out["quad_closed_loop"] = {
            "ms_per_launch": ms, "trajectories": B, "steps": T,
            "closed_loop_steps_per_s": B * T / (ms * 1e-3),
            "what": "mlp_closed_loop_kernel: QuadEvaluator.follow_trajectory for 16 384 "
                    "reference trajectories x 251 steps in one launch (self-play mode: "
                    "reset on divergence, every trajectory runs all steps)"}
    except Exception as e:
        out["quad_closed_loop"] = {"error": repr(e)}
    return out


RCCL_WORLD1_PROBE = r"""
import json, os, sys, tempfile, torch, torch.distributed as dist
dev = torch.device("cuda:0")
store = dist.FileStore(os.path.join(tempfile.mkdtemp(), "s"), 1)
dist.init_process_group("nccl", store=store, rank=0, world_size=1)
out = {}
for n in (32729, 30389, 12341):
    buf = torch.zeros(n, device=dev)
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    out[str(n)] = e0.elapsed_time(e1) / 100 * 1e3
dist.destroy_process_group()
print("RCCL_WORLD1 " + json.dumps(out))
"""


def allreduce_world1_probe():
    """the latency of the step's ONE collective through a
    live RCCL communicator of world size 1 (everything but the wire:
    communicator, RCCL's stream and events) for the three message sizes
    (concurrent / autoregressive / LSTM gradient + loss slot, floats), in a
    child process with a time limit - a hang there must not cost the line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", RCCL_WORLD1_PROBE], capture_output=True,
                           text=True, timeout=90,
                           env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        for line in r.stdout.splitlines():
            if line.startswith("RCCL_WORLD1 "):
                return {"us_per_allreduce_by_floats": json.loads(line[len("RCCL_WORLD1 "):]),
                        "what": "torch.distributed all_reduce(sum) on a nccl (= RCCL) group "
                                "of world size 1, 100 back-to-back calls, HIP events; "
                                "DESIGN.md 6 predicts 15-30 us per call at N = 8"}
        return {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:       # informational only
        return {"error": repr(e)}


def agree_replays(replays, dist, dev):
    """Every rank must run the same number of graph replays."""
    if dist is None:
        return replays
    t = torch.tensor([replays], device=dev, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def max_over_ranks(seconds, dist, dev):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def headline_fields(args, world, nsteps, elapsed, replays, nset):
    """The contract's top-level fields (BASELINE.json metric / config)."""
    H, B = args.horizon, args.batch
    return {
        "metric": "env-steps/sec (fwd+bwd through dynamics), quad horizon=10 batch=65536",
        "value": world * B * H * nsteps / elapsed,
        "unit": "env-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / nsteps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("quadrotor concurrent rollout fwd+bwd (dynamics + "
                         "quad_mpc_loss + adjoint), BASELINE configs[1]"),
            "batch_per_gpu": B, "global_batch": world * B, "horizon": H,
            "dt": args.dt, "layout": args.layout, "buffer_sets": nset,
            "input_bytes_all_sets": nset * B * (48 + 40 * H),
            "grad_state0": bool(args.grad_state0),
            "loss_mode": args.loss_mode,
            "launch": "python" if args.no_graph else "hip-graph replay of the K steps",
            "replays": replays, "timed_steps": nsteps,
            "parallelism": f"batch-sharded x{world}, no data-path collective",
        },
    }


def dry_run_cpu(args, dist, rank, world):
    """The rank logic of main() without a GPU: same process-group calls, same
    replay agreement, barriers and max over ranks; the 'step' is a stub that
    sleeps.  The line is marked `dry_run` - its numbers mean nothing."""
    dev = torch.device("cpu")
    stub = lambda n: time.sleep(2e-5 * n * (1 + rank))   # ranks differ on purpose

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(n, replays):
        barrier()
        t0 = time.perf_counter()
        for _ in range(replays):
            stub(n)
        barrier()
        return time.perf_counter() - t0
    stub(args.warmup)
    probe = timed(args.steps, 1)
    replays = max(1, min(50, int(-(-min(args.min_ms, 20.0) // max(probe * 1e3, 1e-3)))))
    replays = agree_replays(replays + rank, dist, dev)   # unequal proposals
    elapsed = max_over_ranks(timed(args.steps, replays), dist, dev)
    nsteps = args.steps * replays
    out = headline_fields(args, world, nsteps, elapsed, replays, args.sets)
    out["dry_run"] = True
    out["roofline"] = None
    out["cpu_baseline"] = None
    if dist is not None:
        # the collective of the training steps: flat gradient buffer + loss slot
        flat = torch.full((30389 + 1,), float(rank + 1))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        out["allreduce_check"] = float(flat[-1]) == world * (world + 1) / 2
        out["rccl"] = collective_probe(dist, dev, world)   # (gloo here)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["steps_summary"] = steps_summary(out)
        print(json.dumps(out))


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) as a plain command: this process
    becomes the launcher - the same `torch.distributed.run` line the driver
    uses, one rank per GPU, rendezvous on 127.0.0.1 and a free port - and
    returns the launcher's exit code; rank 0 of the children prints the line.
    N is checked against the node first: an explicit error, not a hang in the
    rendezvous or in RCCL's init."""
    import socket
    import subprocess
    if not args.dry_run_cpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
        have = torch.cuda.device_count()
        if args.gpus > have:
            raise SystemExit(f"--gpus {args.gpus}: this node shows {have} GPU(s) "
                             "(torch.cuda.device_count()); one rank per GPU is the "
                             "only form this bench runs")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL needs it here)
    env.setdefault("OMP_NUM_THREADS", "8")              # (the launcher would say 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def collective_probe(dist, dev, world):
    """What the process group saw: world, backend, library version and the
    latency of the training steps' ONE collective - all_reduce(sum) of the flat
    gradient + loss-slot message - for the three message sizes (concurrent /
    autoregressive / LSTM, floats), 100 back-to-back calls on the live group.
    Every rank calls this; the times are rank 0's (HIP events on the GPU, the
    host clock under gloo)."""
    backend = dist.get_backend()
    out = {"world": world, "backend": backend, "allreduce_us": {}}
    if backend == "nccl":
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    for n in (32729, 30389, 12341):
        buf = torch.ones(n + 1, device=dev)
        for _ in range(5):
            dist.all_reduce(buf)
        buf.fill_(1.0)
        sync()
        dist.barrier()
        reps = 100
        if dev.type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                dist.all_reduce(buf)
            e1.record()
            sync()
            us = e0.elapsed_time(e1) / reps * 1e3
        else:
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(buf)
            us = (time.perf_counter() - t0) / reps * 1e6
        out["allreduce_us"][str(n + 1)] = us
        # the sum really crossed the ranks: rank + 1 in every slot -> N (N + 1) / 2
        buf.fill_(float(dist.get_rank() + 1))
        dist.all_reduce(buf)
        out["sums_ok"] = bool(out.get("sums_ok", True)
                              and float(buf[0]) == float(buf[-1]) == world * (world + 1) / 2)
    out["what"] = ("torch.distributed all_reduce(sum) of the step's flat gradient + loss "
                   "message (floats) on the live group, 100 back-to-back calls; nccl = "
                   "RCCL over xGMI")
    return out


def steps_summary(out):
    """The numbers README's table quotes, compact (<= 1 200 characters) and LAST
    in the line: the driver's record keeps the final ~2 000 characters of
    stdout whatever else it drops."""
    r4 = lambda v: None if v is None else float(f"{v:.4g}")
    s = {}
    for key, name in (("train_step", "concurrent"), ("train_step_packed", "packed"),
                      ("train_step_ar", "ar"), ("train_step_lstm", "lstm")):
        blk = out.get(key)
        if isinstance(blk, dict) and "ms_per_step" in blk:
            rf = blk.get("roofline") or {}
            s[name] = {"ms": r4(blk["ms_per_step"]),
                       "frac_vs_mfma_floor": r4(rf.get("frac_vs_mfma_floor")),
                       "bytes_ratio": r4(rf.get("plane_bytes_over_algorithmic"))}
            if isinstance(blk.get("parallel_efficiency"), float):
                s[name]["parallel_efficiency"] = r4(blk["parallel_efficiency"])
                s[name]["global_batch"] = blk.get("global_batch")
    sec = out.get("secondary") if isinstance(out.get("secondary"), dict) else {}
    ws = sec.get("wing_train_step") or {}
    if "ms_per_step" in ws:
        s["wing_step"] = {"ms": r4(ws["ms_per_step"]),
                          "frac_vs_mfma_floor": r4(ws.get("frac_vs_mfma_floor")),
                          "bytes_ratio": r4(ws.get("plane_bytes_over_algorithmic"))}
    re_ = out.get("run_epoch") if isinstance(out.get("run_epoch"), dict) else {}
    ep = {k: {"ms_per_batch": r4(v.get("ms_per_batch")),
              "over_train_step": r4(v.get("over_train_step"))}
          for k, v in re_.items() if isinstance(v, dict) and "ms_per_batch" in v}
    if ep:
        s["run_epoch"] = ep
    wr = sec.get("wing_rollout") or {}
    if "us_per_launch" in wr:
        s["wing_rollout_us"] = r4(wr["us_per_launch"])
        s["wing_rollout_frac_hbm"] = r4((wr.get("hbm") or {}).get("frac"))
    for key, name in (("quad_closed_loop", "quad_closed_loop_ms"),
                      ("wing_closed_loop_eval", "wing_closed_loop_ms"),
                      ("learnt_controller_phase", "learnt_phase_ms")):
        if "ms_per_launch" in (sec.get(key) or {}):
            s[name] = r4(sec[key]["ms_per_launch"])
    rf = out.get("roofline") or {}
    s["headline"] = {"kernel_us": r4(rf.get("kernel_us_avg")), "frac": r4(rf.get("frac")),
                     "stream_floor_us": r4(rf.get("stream_floor_us")),
                     "n_gpus": out.get("n_gpus")}
    rc = out.get("rccl") if isinstance(out.get("rccl"), dict) else None
    if rc and "allreduce_us" in rc:
        s["rccl"] = {"world": rc["world"], "backend": rc["backend"],
                     "allreduce_us": {k: r4(v) for k, v in rc["allreduce_us"].items()}}
    cb = out.get("cpu_baseline") if isinstance(out.get("cpu_baseline"), dict) else {}
    if "value" in cb:
        s["cpu_port_env_steps_per_s"] = r4(cb["value"])
    return s


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} "
                         "ranks (one process per GPU: the two must agree)")
    launched = "RANK" in os.environ   # by torch.distributed.run
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    if args.dry_run_cpu:
        dist = None
        if launched:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        return dry_run_cpu(args, dist, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but this node shows "
                         f"{torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if launched:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from apg_trajectory_tracking_amd import functional as F
    from apg_trajectory_tracking_amd.dynamics.quad_dynamics_flightmare import (
        FlightmareDynamics)
    import gc
    dyn = FlightmareDynamics()
    sets = make_sets(args, rank, dev)
    # all launches of the timed loops go to ONE non-default stream, so that the
    # K steps can be captured into a HIP graph: the GPU then runs them back to
    # back and a host hiccup (a descheduled Python process costs tens of ms)
    # cannot leak into the measurement
    side = torch.cuda.Stream(device=dev)
    nset = len(sets)
    deferred = args.loss_mode == "deferred"
    with torch.cuda.stream(side):
        plans = [F.RolloutPlan("quad", *s, args.dt, dyn.params,
                               layout=args.layout,
                               want_grad_state0=args.grad_state0,
                               loss_mode=args.loss_mode) for s in sets]
        kplans = [F.RolloutPlan("quad", *s, args.dt, dyn.params,
                                layout=args.layout,
                                want_grad_state0=args.grad_state0,
                                loss_mode="none") for s in sets]

    def run_steps(n):
        """n steps; with deferred losses step i's launch also reduces step
        i-1's partials, and the chain is flushed inside the timed region."""
        prev = None
        for i in range(n):
            p = plans[i % nset]
            p.launch(after=prev if deferred else None)
            prev = p
        if deferred and prev is not None:
            prev.flush()

    def run_kernel_only(n):
        for i in range(n):
            kplans[i % nset].launch()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def graph_of(fn, n):
        if args.no_graph:
            return None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn(n)
        return g

    gc.collect()
    gc.disable()
    with torch.cuda.stream(side):
        run_steps(args.warmup)
    torch.cuda.synchronize()
    g_steps = graph_of(run_steps, args.steps)
    g_kernel = graph_of(run_kernel_only, args.steps)

    def timed(fn, n, graph, replays):
        """(host seconds, HIP-event ms) for `replays` x n steps, bracketed by
        barrier + synchronize on both sides."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record()
            for _ in range(replays):
                if graph is not None:
                    graph.replay()
                else:
                    fn(n)
            e1.record()
        barrier()
        return time.perf_counter() - t0, e0.elapsed_time(e1)

    # one untimed pass (graph upload, clocks) that also sizes the region
    _, probe_ms = timed(run_steps, args.steps, g_steps, 1)
    replays = max(1, int(-(-args.min_ms // max(probe_ms, 1e-3))))
    replays = agree_replays(replays, dist, dev)
    elapsed, ev_ms = timed(run_steps, args.steps, g_steps, replays)
    elapsed = max_over_ranks(elapsed, dist, dev)
    loss_check = float(plans[0].out["loss"].item())
    nsteps = args.steps * replays

    # roofline pass: the same rollout launches alone (no loss reduction),
    # bracketed by one HIP-event pair on the launch stream -> average launch
    # duration of the dominant kernel including the kernel-to-kernel boundary
    # (a per-launch event pair would add ~2.5 us of its own to a ~8 us kernel)
    with torch.cuda.stream(side):
        run_kernel_only(min(args.warmup, 10))
    _, k_ms = timed(run_kernel_only, args.steps, g_kernel, replays)
    kernel_ms = k_ms / nsteps

    def events_over(launches):
        """average launch duration over a SUBSET of the buffer sets"""
        torch.cuda.synchronize()
        k0 = torch.cuda.Event(enable_timing=True)
        k1 = torch.cuda.Event(enable_timing=True)
        n = 2000
        with torch.cuda.stream(side):
            for i in range(40):
                launches[i % len(launches)].launch()
            k0.record()
            for i in range(n):
                launches[i % len(launches)].launch()
            k1.record()
        torch.cuda.synchronize()
        return k0.elapsed_time(k1) / n
    other_protocols = {}
    overlap_ms = None
    if not args.headline_only:
        # labelled secondaries: SURVEY §8d's 8 sets (read-only inputs 235 MB:
        # Infinity-Cache resident) and ONE set (everything cache resident)
        for k in (8, 1):
            if k < nset:
                ms = events_over(kplans[:k])
                other_protocols[f"{k}_sets"] = {
                    "kernel_us_avg": ms * 1e3, "launch": "python, HIP events",
                    "input_bytes_all_sets": k * args.batch * (48 + 40 * args.horizon),
                    "note": "inputs (partly) Infinity-Cache resident: NOT an HBM number"}
        # informational: the same launches as TWO independent chains of one
        # graph (even / odd buffer sets on two streams): the tail of one launch
        # and the kernel boundary overlap the head of the next.  Never used
        # for `value` or the roofline.
        if not args.no_graph and nset >= 2:
            try:
                side2 = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(side2):
                    oplans = [F.RolloutPlan("quad", *s, args.dt, dyn.params,
                                            layout=args.layout,
                                            want_grad_state0=args.grad_state0,
                                            loss_mode="none") for s in sets[1::2]]
                eplans = kplans[0::2]

                def run_two_chains(n):
                    fork = torch.cuda.Event()
                    fork.record(side)
                    side2.wait_event(fork)
                    for i in range(n):
                        (eplans if i % 2 == 0 else oplans)[(i // 2) % len(oplans)].launch()
                    join = torch.cuda.Event()
                    join.record(side2)
                    side.wait_event(join)

                with torch.cuda.stream(side):
                    run_two_chains(4)
                torch.cuda.synchronize()
                g_two = graph_of(run_two_chains, args.steps)
                o_replays = max(1, replays // 10)
                _, o_ms = timed(run_two_chains, args.steps, g_two, o_replays)
                overlap_ms = o_ms / (args.steps * o_replays)
                del g_two
            except Exception:        # informational only
                overlap_ms = None
    gc.enable()

    H, B = args.horizon, args.batch
    bytes_per_traj = QUAD_BYTES_PER_TRAJ["base"](H) + (
        QUAD_BYTES_PER_TRAJ["grad_state0"] if args.grad_state0 else 0)
    algo_bytes = B * bytes_per_traj
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_note = load_pmc_traffic(args)
    out = headline_fields(args, world, nsteps, elapsed, replays, nset)
    out["ms_per_step_hip_events"] = ev_ms / nsteps
    out["roofline"] = {
        "bound": "hbm",
        "kernel": {"packed": "quad_rollout_rows_kernel",
                   "soa": "quad_rollout_reg_kernel",
                   "aos": "quad_rollout_aos_kernel"}[args.layout],
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_what": ("L2 <-> fabric bytes per launch (rocprofv3 FETCH_SIZE + "
                         "WRITE_SIZE, calibrated; Infinity-Cache hits are counted "
                         "too): over-fetch check, not a residency proof"),
        "traffic_note": traffic_note,
        "kernel_build": kernel_build_id(),
        "algorithmic_bytes_per_launch": algo_bytes,
        "kernel_us_avg": kernel_ms * 1e3,
        "buffer_sets": nset,
        "other_protocols": other_protocols,
    }
    out["loss_check"] = loss_check
    if overlap_ms is not None:
        out["overlapped_launches"] = {
            "streams": 2, "ms_per_step": overlap_ms,
            "env_steps_per_s": world * B * H / (overlap_ms * 1e-3),
            "algorithmic_GBps": algo_bytes / (overlap_ms * 1e-3) / 1e9,
            "what": "informational: the K kernel-only launches as two independent "
                    "chains of one graph (even / odd buffer sets); `value` and "
                    "`roofline` are the serial launches"}
    # results of buffer set 0 (the cpu_baseline leg compares them with the CPU
    # port on the same tensors: `cpu_baseline.parity_check`)
    gpu_set0 = None
    if rank == 0:
        from apg_trajectory_tracking_amd import synthetic as _sy
        ga0 = plans[0].out["grad_actions"]
        ga0 = {"packed": _sy.from_packed_seq, "soa": _sy.from_soa_seq,
               "aos": lambda t: t}[args.layout](ga0)
        gpu_set0 = {"loss": loss_check, "grad_actions": ga0.detach().cpu()}
    copy_gbps = None
    if not args.headline_only:
        try:
            copy_gbps = measured_copy_bandwidth(dev)
        except Exception as e:          # informational only
            copy_gbps = {"error": repr(e)}
    out["roofline"]["copy_GBps_measured"] = copy_gbps
    if not args.headline_only:
        try:
            fl = stream_floor_probe(args, dev, nset)
            out["roofline"]["stream_floor_us"] = fl["best_us"]
            out["roofline"]["stream_floor"] = fl
            out["roofline"]["kernel_over_stream_floor"] = kernel_ms * 1e3 / fl["best_us"]
        except Exception as e:          # informational only
            out["roofline"]["stream_floor"] = {"error": repr(e)}
    del plans, kplans, sets, g_steps, g_kernel
    gc.collect()
    torch.cuda.empty_cache()
    if args.train_steps > 0 and not args.headline_only:
        for key, mode in (("train_step", "concurrent"),
                          ("train_step_packed", "packed"),
                          ("train_step_ar", "autoregressive"),
                          ("train_step_lstm", "LSTM")):
            try:
                out[key] = trainer_step_probe(args, dev, dyn, dist, mode)
            except Exception as e:      # secondary blocks must not kill the line
                out[key] = {"error": repr(e)}
    if (rank == 0 and world == 1 and args.train_steps > 0
            and not (args.no_secondary or args.headline_only)):
        try:
            out["run_epoch"] = run_epoch_probe(args, dev, dyn)
            for mode_key, step_key in (("concurrent", "train_step"), ("ar", "train_step_ar"),
                                       ("lstm", "train_step_lstm")):
                ts, re_ = out.get(step_key, {}), out["run_epoch"].get(mode_key, {})
                if "ms_per_step" in ts and "ms_per_batch" in re_:
                    out["run_epoch"][mode_key]["over_train_step"] = (
                        re_["ms_per_batch"] / ts["ms_per_step"])
        except Exception as e:
            out["run_epoch"] = {"error": repr(e)}
    # the step blocks again, compact and inside `roofline` (the driver's record
    # keeps that key whole)
    steps = {}
    for key, name in (("train_step", "concurrent"), ("train_step_packed", "packed"),
                      ("train_step_ar", "ar"), ("train_step_lstm", "lstm")):
        blk = out.get(key)
        if isinstance(blk, dict) and "ms_per_step" in blk:
            rf = blk.get("roofline", {})
            steps[name] = {
                "ms": blk["ms_per_step"], "ms_mean": blk.get("ms_per_step_mean"),
                "ms_max": blk.get("ms_per_step_max"),
                "frac_vs_mfma_floor": rf.get("frac_vs_mfma_floor"),
                "bytes_ratio": rf.get("plane_bytes_over_algorithmic"),
                "launch": (blk.get("launch_form") or [{}])[-1].get("chosen", "plan / graph"),
                "default_over_best_form": blk.get("default_over_best_form")}
    if steps:
        out["roofline"]["steps"] = steps
    if rank == 0 and world == 1 and not (args.no_secondary or args.headline_only):
        try:
            out["secondary"] = {"wing_rollout": wing_secondary(args, dev)}
        except Exception as e:      # informational only
            out["secondary"] = {"error": repr(e)}
        try:
            out["secondary"]["wing_closed_loop_eval"] = wing_eval_secondary(args, dev)
        except Exception as e:
            out["secondary"]["wing_closed_loop_eval"] = {"error": repr(e)}
        out["secondary"].update(more_secondaries(args, dev))
        out["roofline"]["allreduce_us_world1"] = allreduce_world1_probe()
    if rank == 0 and world == 1 and not (args.no_cpu_baseline or args.headline_only):
        out["cpu_baseline"] = cpu_baseline(args, gpu_set0)
        out["parity_check"] = out["cpu_baseline"].pop("parity_check")
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None and world > 1:
        try:
            out["rccl"] = collective_probe(dist, dev, world)
        except Exception as e:          # informational only
            out["rccl"] = {"error": repr(e)}
    if rank == 0:
        out["steps_summary"] = steps_summary(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
