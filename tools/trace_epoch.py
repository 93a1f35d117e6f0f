"""Where an epoch's time goes beyond its batches' kernels, out of a rocprofv3
kernel_trace.csv of tools/time_run_epoch.py: the last epoch's span, the sum of
its kernels, every idle gap over 3 us and the time between epochs.
usage: python tools/trace_epoch.py <kernel_trace.csv> <first kernel of a batch> <batches per epoch>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker, nb = sys.argv[2], int(sys.argv[3])
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
first = idx[-nb]                    # first batch of the last epoch
prev_last = idx[-nb - 1]            # last batch of the epoch before
# the epoch's own prologue (permutation, index copy) sits between the previous
# epoch's last step kernel and `first`
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
N = lambda r: r["Kernel_Name"].replace("apg::(anonymous namespace)::", "")[:60]
end = len(rows)
span = E(rows[end - 1]) - S(rows[first])
busy = sum(E(r) - S(r) for r in rows[first:end])
print(f"last epoch: {nb} batches, first step kernel -> last kernel {span / 1e3:.1f} us "
      f"({span / 1e3 / nb:.2f} per batch), kernels {busy / 1e3:.1f} us ({busy / 1e3 / nb:.2f} per batch)")
per = defaultdict(lambda: [0, 0])
for r in rows[first:end]:
    per[N(r)][0] += 1
    per[N(r)][1] += E(r) - S(r)
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t / 1e3 / nb:8.2f} us/batch  x{n / nb:5.2f}  {k}")
gaps = [(S(rows[i + 1]) - E(rows[i]), i) for i in range(first, end - 1)]
big = [(g, i) for g, i in gaps if g > 3000]
print(f"idle inside the epoch: {sum(max(g, 0) for g, _ in gaps) / 1e3:.1f} us, "
      f"{len(big)} gaps over 3 us: " + " ".join(f"{g / 1e3:.0f}@{N(rows[i])[:18]}" for g, i in big[:12]))
# between epochs: previous epoch's last step kernel .. this epoch's first
j = prev_last
while j + 1 < first and marker not in rows[j + 1]["Kernel_Name"]:
    j += 1
print("between the epochs:")
t_prev = E(rows[prev_last])
for r in rows[prev_last:first + 1]:
    print(f"  {(S(r) - t_prev) / 1e3:9.1f} {(E(r) - S(r)) / 1e3:8.1f} {N(r)}")
