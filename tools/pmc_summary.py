"""Aggregate rocprofv3 counter_collection CSVs: mean counter value per dispatch, grouped by kernel name.
usage: python tools/pmc_summary.py <dir-or-csv> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if flt and flt not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
for k in sorted(acc):
    print(k[:110])
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("    %-36s %18.1f   (dispatches %d)" % (c, s / n, n))
