"""Build libapg_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the source tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libapg_hip.so")
SOURCES = ["common.hip", "quad.hip", "wing.hip", "cartpole.hip", "lstm.hip",
           "mlp_rollout.hip", "mlp_concurrent.hip", "mlp_wing.hip", "wing_learnt.hip",
           "linear_wgrad.hip", "planes_gemm.hip"]
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring f32 ops into
# v_pk_fma/mul/add_f32; on gfx950 a packed op issues no faster than two plain
# ones here and needs v_mov shuffles to form register pairs - measured on
# MI355X: quad rollout 9.7 -> 9.0 us, wing rollout 126 -> 94 us
COMMON_FLAGS = ["-fno-slp-vectorize"]
# kernarg preload: the leading scalar arguments of a kernel arrive in SGPRs
# with the wave (gfx950 firmware feature) - the packed-rows rollout issues its
# first loads without waiting for an s_load round trip
EXTRA_FLAGS = {"quad.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}
def _headers():
    """Every header a source may include: all of csrc/*.h plus the C ABI (a
    hand-kept list once missed policy_mfma16.h, so `python -m ...build` did
    not see edits to it)."""
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(REPO, "include", "apg.h")]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(
            os.path.join(CSRC, "kernel_resources.json")):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _extra_flags():
    """APG_HIPCC_FLAGS: extra compiler flags for the product build (-g,
    -save-temps ...).  Kernel-variant macros are refused here - they belong to
    tools/build_variant.py, which builds a SEPARATE library - so that a stray
    -DAPG_... in the environment cannot change the shipped kernels (the
    sources #error on them as well, apg_device.h)."""
    flags = os.environ.get("APG_HIPCC_FLAGS", "").split()
    bad = [f for f in flags if f.startswith(("-DAPG_", "-UAPG_"))]
    if bad:
        raise RuntimeError(
            f"APG_HIPCC_FLAGS must not define kernel macros ({' '.join(bad)}): "
            "build variants with tools/build_variant.py")
    return flags


RESOURCES = os.path.join(CSRC, "kernel_resources.json")


def _kernel_resources(remarks):
    """{demangled-ish kernel name: {vgprs, agprs, sgprs, scratch, occupancy}} out of
    hipcc's -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs",
            "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill"}
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in keys:
            cur[keys[m.group(1)]] = int(m.group(2))
    return out


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not _stale():
        return LIB
    extra = _extra_flags()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               *COMMON_FLAGS, *EXTRA_FLAGS.get(s, []),
               *extra, "-Rpass-analysis=kernel-resource-usage",
               "-I", os.path.join(REPO, "include"), "-I", CSRC, "-c", src,
               "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(
            cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    resources = {}
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        resources.update(_kernel_resources(out))
    # registers, scratch and waves per SIMD of every kernel as the compiler
    # reports them: tests/test_host_cpu.py holds the kernels that are written for
    # two waves per SIMD to that (five more live registers once cost the LSTM
    # forward sweep its second wave: 94 -> 109 us)
    import json
    with open(RESOURCES, "w") as f:
        json.dump(resources, f, indent=0, sort_keys=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stdout}")
    return LIB


LIB_CPU = os.path.join(CSRC, "libapg_cpu.so")


def build_cpu(force=False, verbose=False):
    """libapg_cpu.so (include/apg_cpu.h): the per-trajectory headers of the
    kernels compiled for the HOST behind the `..._cpu` twins of the dynamics
    entry points.  A separate library that the package never loads."""
    src = os.path.join(CSRC, "cpu_twins.hip")
    deps = [src, os.path.join(REPO, "include", "apg_cpu.h"), __file__] + _headers()
    if (not force and os.path.exists(LIB_CPU)
            and os.path.getmtime(LIB_CPU) >= max(os.path.getmtime(d) for d in deps)):
        return LIB_CPU
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # (-no-hip-rt: the library depends on libstdc++ / libm only - it loads on
    # a machine without ROCm)
    cmd = [hipcc, "--cuda-host-only", "-no-hip-rt", "-O2", "-std=c++17", "-fPIC",
           "-shared",
           "-I", os.path.join(REPO, "include"), "-I", CSRC, "-o", LIB_CPU, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc (host) failed on cpu_twins.hip:\n{r.stdout}")
    return LIB_CPU


LIB_PLANES = os.path.join(CSRC, "libapg_planes.so")


def build_planes(force=False, verbose=False):
    """libapg_planes.so (include/apg_planes.h): the plane-writing reverse kernels
    of rounds 1-4 behind their C entry points - a TEST library (tests/
    plane_path.py loads it; the package never does): the product has one reverse
    kernel per training mode.  mlp_planes.hip + common.hip (error plumbing, the
    loss reduction), self-contained (-Bsymbolic: its own copies of the shared
    helpers whatever else the process has loaded)."""
    srcs = [os.path.join(CSRC, s) for s in ("mlp_planes.hip", "common.hip")]
    deps = srcs + [os.path.join(REPO, "include", "apg_planes.h"), __file__] + _headers()
    if (not force and os.path.exists(LIB_PLANES)
            and os.path.getmtime(LIB_PLANES) >= max(os.path.getmtime(d) for d in deps)):
        return LIB_PLANES
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *COMMON_FLAGS,
           "-shared", "-Wl,-Bsymbolic", "-I", os.path.join(REPO, "include"), "-I", CSRC,
           "-o", LIB_PLANES, *srcs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on mlp_planes.hip:\n{r.stdout}")
    return LIB_PLANES


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cpu(force="--force" in sys.argv, verbose=True))
    print(build_planes(force="--force" in sys.argv, verbose=True))
