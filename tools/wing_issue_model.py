"""Per-wave issue-time model of the fixed-wing rollout kernel.

A wave that has its SIMD to itself (the kernel's regime: B = 131 072 at two
trajectories per lane = ONE wave per SIMD) issues in order, one instruction at
a time: its run time is the SUM of its instructions' issue costs plus the
stalls of its waits - there is no instruction-level parallelism to find, the
"critical path" IS the instruction stream.  This tool prices the stream:

  * the kernel is compiled to ISA (hipcc -S), split into basic blocks, every
    instruction is put into a cost class;
  * the per-class costs are the ones MEASURED on this part for a lone wave
    (tools/issue_probe*.hip -> profiles/r02_issue_probe.jsonl,
    profiles/r03_issue_probe2.jsonl; DESIGN.md 3.1's table);
  * a block's multiplicity is the trip count of the loop the compiler's own
    annotations put it in (forward sweep: H; reverse sweep: H / 4 checkpoint
    groups; both from the kernel source), given on the command line.

Output: cycles per class and per loop, the total, and the total at 2.25 GHz next
to the measured wave time (profiles/r04_wing_clock.jsonl: 123 616 cycles).

  python tools/wing_issue_model.py [--asm FILE] [--kernel SUBSTR]
        [--trips BB4_73=20 BB4_93=5 BB4_4=0] [--json OUT]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "apg_trajectory_tracking_amd", "csrc")

# cycles of issue for a lone wave on its SIMD (sources in the module docstring)
COST = collections.OrderedDict([
    ("valu_pk", (5.14, "v_pk_*_f32: r03_issue_probe2 pk_fma_dist / pk_mul_dist")),
    ("valu_fma3", (5.14, "v_fma_f32 with three distinct sources: r03 fma_dist3")),
    ("valu_plain", (4.13, "v_mul / v_add / v_fmac / v_mov / v_accvgpr / compares: r03 mul_dist2, fmac_dist")),
    ("valu_trans", (8.1, "v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp / v_log: r02 probe, two slots")),
    ("valu_cndmask_vcc", (19.0, "v_cndmask_b32_e32 (VCC): r02 probe")),
    ("salu", (4.1, "any scalar instruction between VALU ops: r02 probe")),
    ("waitcnt", (4.1, "s_waitcnt: its slot; the stall behind it is NOT modelled")),
    ("lds_read", (13.0, "ds_read between VALU ops: r02 probe")),
    ("lds_write", (17.0, "ds_write between VALU ops: r02 probe")),
    ("vmem_load", (42.0, "buffer_load (256-512 B per wave): r02 probe 37-47")),
    ("vmem_store", (33.0, "buffer_store: r02 probe")),
    ("smem", (4.1, "s_load: a scalar slot")),
    ("branch", (4.1, "s_cbranch / s_branch")),
])
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")


def classify(op):
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store", "ds_add")):
        return "lds_write"
    if op.startswith(("buffer_load", "global_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("buffer_store", "global_store", "flat_store", "buffer_atomic")):
        return "vmem_store"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_cndmask_b32_e32"):
        return "valu_cndmask_vcc"
    if op.startswith(("v_fma_", "v_fmaak", "v_fmamk", "v_mad_")):
        return "valu_fma3"
    if op.startswith("v_"):
        return "valu_plain"
    raise ValueError(f"unclassified instruction {op!r}")


def kernel_body(asm, substr):
    lines = open(asm).read().splitlines()
    start = next(i for i, l in enumerate(lines)
                 if re.match(r"^_Z\S*:", l) and substr in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start + 1:end + 1]


def blocks_of(body):
    """[(label, loop header or None, [mnemonics])] in program order."""
    out, cur = [], ["entry", None, []]
    for l in body:
        m = re.match(r"^\.?(LBB\d+_\d+):\s*(;.*)?$", l) or \
            re.match(r"^; %(bb\.\d+):\s*(;.*)?$", l)
        if m:
            out.append(tuple(cur))
            note = m.group(2) or ""
            hdr = None
            if "Loop Header" in note:
                hdr = m.group(1)[1:]
            mm = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
            if mm:
                hdr = mm.group(1)
            cur = [m.group(1), hdr, []]
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur[2].append(t.split()[0])
    out.append(tuple(cur))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--kernel", default="wing_rollout_pk_kernelILi1E")
    ap.add_argument("--trips", nargs="*", default=["BB4_73=20", "BB4_93=5", "BB4_4=0"])
    # blocks behind a condition of the launch: factor on their loop's trip count
    # (defaults: the states_out stores of the forward step - no states are asked
    # for in the bench launch; the checkpoint stash - every 4th forward step)
    ap.add_argument("--scale", nargs="*", default=["bb.77=0", "LBB4_75=0.25"])
    ap.add_argument("--ghz", type=float, default=2.25)
    ap.add_argument("--measured-cycles", type=float, default=123616.0)
    ap.add_argument("--json")
    a = ap.parse_args()
    asm = a.asm
    if asm is None:
        asm = "/tmp/apg_wing_model.s"
        subprocess.check_call(
            ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
             "-I" + os.path.join(HERE, "..", "include"), "-S", "--cuda-device-only",
             os.path.join(SRC, "wing.hip"), "-o", asm], stderr=subprocess.DEVNULL)
    trips = dict(t.split("=") for t in a.trips)
    trips = {k: float(v) for k, v in trips.items()}
    scale = {k: float(v) for k, v in (t.split("=") for t in a.scale)}
    blocks = blocks_of(kernel_body(asm, a.kernel))
    per_class = collections.Counter()
    per_loop = collections.Counter()
    counts = collections.Counter()
    static = collections.Counter()
    seen_loops = set()
    for label, hdr, ops in blocks:
        mult = 1.0 if hdr is None else trips.get(hdr)
        if hdr is not None:
            seen_loops.add(hdr)
        if mult is None:
            sys.exit(f"loop {hdr} has no trip count (--trips {hdr}=N)")
        mult *= scale.get(label, 1.0)
        for op in ops:
            c = classify(op)
            static[c] += 1
            counts[c] += mult
            per_class[c] += mult * COST[c][0]
            per_loop[hdr or "straight-line"] += mult * COST[c][0]
    total = sum(per_class.values())
    res = {
        "kernel": a.kernel, "trips": trips, "block_scale": scale,
        "loops_found": sorted(seen_loops),
        "static_instructions": dict(static), "dynamic_instructions": dict(counts),
        "cycles_per_class": {k: round(v, 1) for k, v in per_class.items()},
        "cycles_per_loop": {k: round(v, 1) for k, v in per_loop.items()},
        "issue_cycles_total": round(total, 1),
        "us_at_clock": round(total / (a.ghz * 1e3), 2), "GHz": a.ghz,
        "measured_wave_cycles": a.measured_cycles,
        "model_over_measured": round(total / a.measured_cycles, 3),
        "cost_table": {k: v[0] for k, v in COST.items()},
    }
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
