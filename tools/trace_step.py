"""Timeline of one training step out of a rocprofv3 kernel_trace.csv: the
kernels between the last two launches whose name contains <marker>.
usage: python tools/trace_step.py <kernel_trace.csv> <marker>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
i0, i1 = idx[-2], idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    name = r["Kernel_Name"].replace("apg::(anonymous namespace)::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {name}")
span = int(rows[i1]["Start_Timestamp"]) - t0
print(f"step span {span / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us")
