"""Per-kernel means of a rocprofv3 --pmc counter_collection.csv (the raw file
is tens of MB; this summary is what goes into profiles/).
usage: python tools/pmc_summary.py <counter_collection.csv> <label> >> out.csv"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    rows[(r["Kernel_Name"][:90], r["Grid_Size"], r["Counter_Name"])].append(
        float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["command", "Kernel", "Grid_Size", "Counter_Name", "mean", "min", "max", "count"])
for (k, g, c), v in sorted(rows.items()):
    w.writerow([sys.argv[2], k, g, c, sum(v) / len(v), min(v), max(v), len(v)])
