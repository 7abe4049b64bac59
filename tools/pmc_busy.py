"""Matrix-pipe busy fraction of the recurrence kernels from a rocprofv3 --pmc pass that holds SQ_VALU_MFMA_BUSY_CYCLES and
SQ_WAVE_CYCLES (tools/final_pass.sh, pass `sq`): busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES) -- the first counts
per clock and SIMD, the second once per 4 clocks per wave, one wave per SIMD in these kernels.  Written to
profiles/pipe_busy.json, each entry stamped with bench.kernel_source_sha() of the CURRENT tree and with the profiles/ file the
counters are kept in; bench.py reports roofline.pipe_busy only from entries whose stamp matches the tree it runs on.
usage: python tools/pmc_busy.py <pass-dir> <profiles-file-name>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        if not name.startswith("gru_"):
            continue
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
path = os.path.join(root, "profiles", "pipe_busy.json")
try:
    table = json.load(open(path))
except Exception:
    table = {}
for name, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "SQ_WAVE_CYCLES" not in c:
        continue
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    wv = c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
    table[name] = {"pipe_busy": mf / (4.0 * wv), "mfma_busy_cycles": mf, "wave_cycles": wv, "dispatches": c["SQ_WAVE_CYCLES"][1],
                   "source": "profiles/" + sys.argv[2], "source_sha": bench.kernel_source_sha()}
    print(name, table[name])
json.dump(table, open(path, "w"), indent=1)
