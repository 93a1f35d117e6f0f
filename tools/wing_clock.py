"""Shader clock under the fixed-wing rollout: a variant build
of wing.hip (-DAPG_WING_CLOCK, tools/build_wing_variant.sh clock) stamps
s_memtime / s_memrealtime at the first and last instruction of every wave;
this script launches configs[3] (B = 131 072, H = 20: 1 024 waves of two
trajectories per lane = one wave per SIMD) and a half-chip batch, reads the
stamps back and reports the effective shader frequency per wave.
    APG_LIB=tools/exp/libapg_wing_clock.so python tools/wing_clock.py"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import _capi, functional as F, synthetic  # noqa: E402
from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (  # noqa: E402
    FixedWingDynamics)

REF_HZ = 100e6      # s_memrealtime: the 100 MHz reference clock


def main():
    dev = torch.device("cuda:0")
    lib = _capi.lib()
    lib.apg_wing_clock_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dyn = FixedWingDynamics()
    H, dt = 20, 0.05
    # (sets: 1 = one buffer set, inputs cache resident; 4 = bench.py's
    # secondary.wing_rollout protocol, four rotating sets)
    for B, force, sets in ((131072, None, 1), (131072, None, 4), (65536, 1, 1), (32768, 1, 1)):
        if force is not None:
            _capi.check(lib.apg_wing_set_two_per_lane(force), "set")
        plans = []
        for i in range(sets):
            d = synthetic.wing_batch(B, H, dt, seed=i)
            plans.append(F.RolloutPlan(
                "wing", synthetic.to_soa_state(d["state0"]).to(dev),
                synthetic.to_soa_seq(d["actions"]).to(dev),
                synthetic.to_soa_seq(d["ref"]).to(dev), dt, dyn.params,
                layout="soa", loss_mode="none"))
        for i in range(200):          # the chip in its sustained state
            plans[i % sets].launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            plans[i % sets].launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        waves = B // 128
        buf = np.zeros(4 * waves, dtype=np.uint64)
        _capi.check(lib.apg_wing_clock_read(buf.ctypes.data, buf.size), "clock_read")
        c = buf.reshape(waves, 4).astype(np.float64)
        cycles, ref = c[:, 1] - c[:, 0], c[:, 3] - c[:, 2]
        ghz = cycles / (ref / REF_HZ) / 1e9
        span_us = (c[:, 3].max() - c[:, 2].min()) / REF_HZ * 1e6
        # where the microseconds between a wave's own
        # duration and the launch are - the ramp (first wave's start to the
        # last wave's start), the waves themselves, the tail (median end to
        # last end), and what the HIP events add around the span
        t0 = c[:, 2].min()
        us_of = lambda v: float((v - t0) / REF_HZ * 1e6)
        attribution = {
            "last_wave_start_us": us_of(c[:, 2].max()),
            "median_wave_end_us": us_of(np.median(c[:, 3])),
            "last_wave_end_us": us_of(c[:, 3].max()),
            "events_minus_span_us": float(us - span_us),
            "wave_us_at_2p25GHz": float(np.median(cycles) / 2.25e3),
            "clock_loss_us": float(np.median(ref) / REF_HZ * 1e6 - np.median(cycles) / 2.25e3)}
        print(json.dumps({
            "batch": B, "waves": waves, "buffer_sets": sets, "us_per_launch_events": us,
            "wave_cycles_median": float(np.median(cycles)),
            "wave_us_median": float(np.median(ref) / REF_HZ * 1e6),
            "launch_span_us_first_start_to_last_end": float(span_us),
            "shader_GHz_median": float(np.median(ghz)),
            "shader_GHz_p05_p95": [float(np.percentile(ghz, 5)), float(np.percentile(ghz, 95))],
            "attribution": attribution, "ref_clock_Hz": REF_HZ}))
    _capi.check(lib.apg_wing_set_two_per_lane(2), "set")


if __name__ == "__main__":
    main()
