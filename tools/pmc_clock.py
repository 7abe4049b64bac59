"""Effective shader clock per kernel from a rocprofv3 --pmc pass that holds GRBM_GUI_ACTIVE (+ --kernel-trace): the chip clocks to
its power budget (MI355X_MICROARCH.md, "DVFS give-back"), effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time.  Joins the
counter rows with the kernel trace of the SAME pass by dispatch id; prints mean clock, mean duration and, where the pass holds them,
matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_WAVE_CYCLES x 4 / waves per SIMD) per kernel name.
usage: python tools/pmc_clock.py <pass-dir> [name-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2:]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not cc:
    sys.exit("no counter_collection.csv under " + d)
rows = list(csv.DictReader(open(cc[0])))
cols = rows[0].keys()
did = "Dispatch_Id" if "Dispatch_Id" in cols else [c for c in cols if "ispatch" in c][0]
dur = {}
if "Start_Timestamp" in cols and "End_Timestamp" in cols:
    for r in rows:
        dur[r[did]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
elif kt:
    for r in csv.DictReader(open(kt[0])):
        k = r.get("Dispatch_Id") or r.get("dispatch_id")
        dur[k] = int(r.get("End_Timestamp") or r.get("end_timestamp")) - int(r.get("Start_Timestamp") or r.get("start_timestamp"))
else:
    sys.exit("no timestamps: columns " + ", ".join(cols))
per = defaultdict(lambda: defaultdict(dict))
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and not any(f in name for f in flt):
        continue
    per[name][r[did]][r["Counter_Name"]] = float(r["Counter_Value"])
print("%-64s %6s %10s %10s %9s" % ("kernel", "n", "mean us", "clock MHz", "GUI/8/us"))
for name in sorted(per):
    clk, us, n = 0.0, 0.0, 0
    for k, c in per[name].items():
        if "GRBM_GUI_ACTIVE" not in c or k not in dur or dur[k] <= 0:
            continue
        clk += c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[k] * 1e-3)
        us += dur[k] * 1e-3
        n += 1
    if n:
        print("%-64s %6d %10.1f %10.0f" % (name[:64], n, us / n, clk / n))
