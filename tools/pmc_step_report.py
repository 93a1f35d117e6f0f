"""The concurrent step's PMC counters as the derived figures DESIGN.md quotes
(profiles/r05_pmc_concurrent_step.txt).  Input: the per-pass summaries of
tools/pmc_summary.py (one CSV per --pmc pass), concatenated.
usage: python tools/pmc_step_report.py [concurrent|autoregressive|LSTM] summary1.csv ... > report.txt"""
import collections
import csv
import sys

KERNELS = {
    "concurrent": (("mlp_concurrent_bwd_tm_kernel", "bwd_tm"), ("mlp_concurrent_fwd_kernel", "fwd")),
    "autoregressive": (("mlp_rollout_bwd_tm_kernel", "ar_bwd_tm"), ("mlp_rollout_fwd_kernel", "ar_fwd")),
    "LSTM": (("lstm_rollout_bwd_kernel", "lstm_bwd"), ("lstm_rollout_fwd_kernel", "lstm_fwd"),
             ("lstm_gate_wgrad_kernel", "lstm_gate_wgrad"), ("planes_gemm", "planes_gemm")),
}
args = sys.argv[1:]
mode = args.pop(0) if args and args[0] in KERNELS else "concurrent"
vals = collections.defaultdict(dict)
for path in args:
    for r in csv.DictReader(open(path)):
        if r.get("Kernel") in (None, "Kernel"):
            continue
        vals[r["Kernel"]][r["Counter_Name"]] = float(r["mean"])
WAVES, SIMDS, XCDS = 2048, 1024, 8
for name, short in KERNELS[mode]:
    ks = [k for k in vals if name in k]
    if not ks:
        continue
    # (several instantiations share a name: the one with the most cycles)
    ks.sort(key=lambda k: -vals[k].get("SQ_WAVE_CYCLES", vals[k].get("SQ_INSTS_VALU", 0)))
    c = vals[ks[0]]
    g = lambda k: c.get(k, float("nan"))
    cyc = g("GRBM_GUI_ACTIVE") / XCDS
    print(f"{short}: kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs) {cyc:,.0f}")
    mf = g("SQ_VALU_MFMA_BUSY_CYCLES") / SIMDS
    print(f"  matrix pipe busy per SIMD  = SQ_VALU_MFMA_BUSY_CYCLES / 1024 = {mf:,.0f} cycles = {100 * mf / cyc:.0f}% of the kernel")
    vw = g("SQ_INSTS_VALU") / WAVES
    print(f"  VALU instructions per wave = SQ_INSTS_VALU / 2048 = {vw:,.0f} (x 4 issue cycles x 2 waves per SIMD = {8 * vw:,.0f} cycles = {100 * 8 * vw / cyc:.0f}% of the kernel)")
    print(f"  LDS instructions per wave  = {g('SQ_INSTS_LDS') / WAVES:,.0f}; bank-conflict cycles / LDS active = {100 * g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_ACTIVE_INST_LDS'), 1):.0f}%; VMEM reads per wave {g('SQ_INSTS_VMEM_RD') / WAVES:,.0f}; SALU per wave {g('SQ_INSTS_SALU') / WAVES:,.0f}")
    print(f"  waiting: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.0f}%; SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = {100 * g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES'):.1f}%")
    if "SQ_INSTS_MFMA" in c:
        print(f"  matrix instructions per wave = SQ_INSTS_MFMA / 2048 = {g('SQ_INSTS_MFMA') / WAVES:,.0f}; "
              f"VMEM writes per wave {g('SQ_INSTS_VMEM_WR') / WAVES:,.0f}; "
              f"issuing: SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {100 * g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES'):.0f}%, "
              f"issue-stalled: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {100 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.0f}%")
    print("  raw: " + ", ".join(f"{k}={v:,.0f}" for k, v in sorted(c.items())))
    print()
