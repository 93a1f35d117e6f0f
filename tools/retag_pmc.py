"""Re-tag profiles/pmc_traffic.json entries to the current kernel sources -
ONLY after proving that the device code of the kernel's translation unit is
unchanged since the revision the counters were taken on.

    python tools/retag_pmc.py <git-rev> quad|wing "<why>"

Compiles csrc/<unit>.hip of <git-rev> (with that revision's headers) and of the
working tree with `hipcc -S --cuda-device-only` and compares the assembly with
comments, .file / .loc / .ident lines and the compilation-unit id removed.  If
they are equal, the entry's `kernel_build` moves to bench.kernel_build_id() of
the working tree and the step is appended to `kernel_build_history`."""
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CSRC = os.path.join(REPO, "apg_trajectory_tracking_amd", "csrc")
UNITS = {"quad": ("quad.hip", "quad_B65536_H10_packed", "KERNEL_SOURCES",
                  ["-mllvm", "-amdgpu-kernarg-preload-count=16"]),
         "wing": ("wing.hip", "wing_B131072_H20_soa", "WING_SOURCES", [])}


def device_asm(csrc, inc, unit, extra):
    out = subprocess.run(
        ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
         "-fno-slp-vectorize", *extra, "-I", inc, "-I", csrc, "-S",
         "--cuda-device-only", os.path.join(csrc, unit), "-o", "-"],
        check=True, capture_output=True, text=True).stdout
    keep = []
    for line in out.splitlines():
        line = re.sub(r";.*", "", line).rstrip()
        if not line or re.match(r"\s*\.(file|loc|ident)\b", line):
            continue
        keep.append(re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", line))
    return keep


def main():
    rev, which, why = sys.argv[1], sys.argv[2], sys.argv[3]
    unit, key, sources, extra = UNITS[which]
    with tempfile.TemporaryDirectory() as tmp:
        old_csrc, old_inc = os.path.join(tmp, "csrc"), os.path.join(tmp, "include")
        os.makedirs(old_csrc), os.makedirs(old_inc)
        names = subprocess.run(["git", "ls-tree", "--name-only", rev,
                                "apg_trajectory_tracking_amd/csrc/", "include/"],
                               cwd=REPO, check=True, capture_output=True,
                               text=True).stdout.split()
        for n in names:
            if n.endswith((".h", ".hip")):
                dst = os.path.join(old_inc if n.startswith("include/") else old_csrc,
                                   os.path.basename(n))
                with open(dst, "w") as f:
                    f.write(subprocess.run(["git", "show", f"{rev}:{n}"], cwd=REPO,
                                           check=True, capture_output=True,
                                           text=True).stdout)
        old = device_asm(old_csrc, old_inc, unit, extra)
    new = device_asm(CSRC, os.path.join(REPO, "include"), unit, extra)
    if old != new:
        diff = sum(a != b for a, b in zip(old, new)) + abs(len(old) - len(new))
        raise SystemExit(f"{unit}: device code differs from {rev} ({diff} lines): "
                         "take a new PMC pass instead of re-tagging")
    import bench
    build = bench.kernel_build_id(getattr(bench, sources))
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    with open(path) as f:
        d = json.load(f)
    e = d[key]
    if e["kernel_build"] != build:
        e.setdefault("kernel_build_history", []).append(
            {"from": e["kernel_build"], "to": build, "device_code_equal_to": rev,
             "why": why})
        e["kernel_build"] = build
        with open(path, "w") as f:
            json.dump(d, f, indent=1)
    print(f"{unit}: device code equal to {rev}; {key}.kernel_build = {build}")


if __name__ == "__main__":
    main()
