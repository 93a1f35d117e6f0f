"""Instruction counts of each variant's loop body in tools/transposition_probe.hip
(from the compiler's assembly: hipcc ... -save-temps=obj; no GPU needed).
usage: python tools/transposition_probe_counts.py tools/exp/transposition_probe-hip-amdgcn-amd-amdhsa-gfx950.s
Prints one JSON line per kernel: instructions by class in the largest loop (the
layer loop) - matrix, VALU (without matrix), LDS (ds_*; the transposing reads and
the permlane swaps listed on their own), VMEM, SALU, waits."""
import collections
import json
import re
import sys

text = open(sys.argv[1]).read().splitlines()
kern = None
bodies = collections.OrderedDict()
for ln in text:
    m = re.match(r"^(_Z\w+):", ln)
    if m and "probe_kernel" in m.group(1):
        kern = m.group(1)
        bodies[kern] = []
        continue
    if kern and ln.startswith(".Lfunc_end"):
        kern = None
    if kern:
        bodies[kern].append(ln)


def classify(op):
    if op.startswith("v_mfma"):
        return "matrix"
    if op.startswith("ds_read_b64_tr"):
        return "lds_tr_read"
    if op.startswith("v_permlane"):
        return "permlane_swap"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    if "_dpp" in op:
        return "valu_dpp"
    if op.startswith("v_"):
        return "valu"
    return "other"


for name, lines in bodies.items():
    labels = {}
    ins = []
    for ln in lines:
        s = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        ins.append(s.split(";")[0].strip())
    best = None
    for i, s in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            span = (labels[m.group(1)], i + 1)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    body = ins[best[0]:best[1]] if best else ins
    cnt = collections.Counter()
    for s in body:
        cnt[classify(s.split()[0])] += 1
    v = re.search(r"ILi(\d)E", name)
    out = {"variant": int(v.group(1)) if v else name, "loop_instructions": len(body)}
    out.update(sorted(cnt.items()))
    print(json.dumps(out))
