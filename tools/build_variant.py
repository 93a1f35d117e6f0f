"""Build an experimental variant of libapg_hip.so: quad.hip recompiled with
extra flags, linked with the other (shipped) objects -> tools/exp/libapg_<name>.so
    python tools/build_variant.py <name> [-DAPG_QX=5 -DAPG_HW_TRIG ...]
Also dumps the ISA statistics of the bench kernel instance."""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from apg_trajectory_tracking_amd import build as B  # noqa: E402

KERNELS = ["quad_rollout_reg_kernelILi0ELi10ELb0ELb1E",
           "quad_rollout_rows_kernelILi10ELb0E"]


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    B.build()
    out = os.path.join(REPO, "tools", "exp")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(B.CSRC, "quad.hip")
    obj = os.path.join(out, f"quad_{name}.o")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
            "-fPIC", *B.COMMON_FLAGS, "-DAPG_EXPERIMENT_BUILD", *flags, "-I",
            os.path.join(REPO, "include"), "-I", B.CSRC]
    subprocess.run(base + ["-c", src, "-o", obj], check=True)
    objs = [os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES
            if s != "quad.hip"] + [obj]
    lib = os.path.join(out, f"libapg_{name}.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared",
                    "-fPIC", "-o", lib] + objs, check=True)
    asm = os.path.join(out, f"quad_{name}.s")
    r = subprocess.run(base + ["-S", "--cuda-device-only", src, "-o", asm,
                               "-Rpass-analysis=kernel-resource-usage"],
                       stderr=subprocess.PIPE, text=True)
    txt = open(asm).read()
    for KERNEL in KERNELS:
        report(name, KERNEL, txt, r.stderr)


def report(name, KERNEL, txt, stderr):
    class r:  # noqa: N801
        pass
    r.stderr = stderr
    m = re.search(r"^_ZN3apg12_GLOBAL__N_1\d+%s.*?:\n(.*?)s_endpgm" % KERNEL, txt,
                  re.S | re.M)
    body = m.group(1) if m else ""
    ins = [l.split()[0] for l in body.splitlines()
           if l.startswith("\t") and l.strip()
           and not l.strip().startswith((".", ";"))]
    cnt = lambda p: sum(1 for i in ins if re.match(p, i))
    regs = re.search(r"%s.*?VGPRs: (\d+).*?AGPRs: (\d+)" % KERNEL, r.stderr, re.S)
    print(f"{name} {KERNEL[13:24]}: instr {len(ins)} valu {cnt('v_')} accvgpr {cnt('v_accvgpr')} "
          f"trans {cnt('v_(sin|cos|rcp|sqrt|exp|log)')} salu {cnt('s_') - cnt('s_waitcnt')} "
          f"waitcnt {cnt('s_waitcnt')} vmem_ld {cnt('buffer_load')} "
          f"vmem_st {cnt('buffer_store')} ds {cnt('ds_')} "
          f"vgpr/agpr {regs.group(1) if regs else '?'}/{regs.group(2) if regs else '?'}")


if __name__ == "__main__":
    main()
