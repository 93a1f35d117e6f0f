"""Timeline of one training step out of a rocprofv3 kernel_trace.csv: the
kernels between two consecutive launches whose name contains <marker> - the last
two, or with <back> = n the pair n launches before the end (an epoch's last batch
shares the device with the next epoch's permutation sort: use 4 there).
usage: python tools/trace_step.py <kernel_trace.csv> <marker> [back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].replace("apg::(anonymous namespace)::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {name}")
span = int(rows[i1]["Start_Timestamp"]) - t0
print(f"step span {span / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us")
