"""Kernel-only timing of the fused fixed-wing rollout at several batch sizes
(1 / 2 / 4 waves per SIMD at B = 65 536 / 131 072 / 262 144):
    python tools/time_wing.py [B ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from apg_trajectory_tracking_amd import functional as F, synthetic  # noqa: E402
from apg_trajectory_tracking_amd.dynamics.fixed_wing_dynamics import (  # noqa: E402
    FixedWingDynamics)


def main():
    # WING_PK=0|1|2: which rollout kernel (apg_wing_set_two_per_lane; the
    # library itself reads nothing from the environment)
    if "WING_PK" in os.environ:
        from apg_trajectory_tracking_amd import _capi
        _capi.check(_capi.lib().apg_wing_set_two_per_lane(int(os.environ["WING_PK"])),
                    "apg_wing_set_two_per_lane")
    dev = torch.device("cuda:0")
    dyn = FixedWingDynamics()
    H, dt = 20, 0.05
    for B in [int(a) for a in sys.argv[1:]] or [65536, 131072, 262144]:
        plans = []
        for i in range(4):
            d = synthetic.wing_batch(B, H, dt, seed=i)
            plans.append(F.RolloutPlan(
                "wing", synthetic.to_soa_state(d["state0"]).to(dev),
                synthetic.to_soa_seq(d["actions"]).to(dev),
                synthetic.to_soa_seq(d["ref"]).to(dev), dt, dyn.params,
                layout="soa", loss_mode="none"))
        for i in range(10):
            plans[i % 4].launch()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        n = 100
        e0.record()
        for i in range(n):
            plans[i % 4].launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(json.dumps({"config": "wing_rollout", "batch": B, "horizon": H,
                          "us_per_launch": us,
                          "env_steps_per_s": B * H / us * 1e6,
                          "algorithmic_GBps": B * 928 / us / 1e3}))


if __name__ == "__main__":
    main()
