"""HBM traffic per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs) of the
same command: mean per dispatch grouped by (kernel, grid size).  bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- gfx950's
FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads (MI355X_MICROARCH.md, HBM section).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [name-filter ...]
       python tools/pmc_traffic.py --stamp <json-from-above> [kernel-name ...]
           merges the entries (keys reduced to the kernel name; of several grid sizes the one with the most launches)
           into profiles/hbm_traffic.json, each stamped with bench.kernel_source_sha() of the CURRENT tree -- bench.py
           reports roofline.traffic only from entries whose stamp matches the tree it runs on."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


if len(sys.argv) > 2 and sys.argv[1] == "--stamp":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    new = json.load(open(sys.argv[2]))
    want = sys.argv[3:]
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    table = json.load(open(path))
    best = {}
    for key, v in new.items():
        name = key.split("|")[0].split("<")[0]
        if want and name not in want:
            continue
        if name not in best or v["launches_averaged"] > best[name]["launches_averaged"]:
            best[name] = v
    for name, v in best.items():
        table[name] = dict(v, source_sha=bench.kernel_source_sha())
        print("stamped", name, table[name])
    json.dump(table, open(path, "w"), indent=1)
    sys.exit(0)


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            a = acc[(name, row.get("Grid_Size", ""))]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
flt = sys.argv[3:]
out = {}
for key in sorted(set(fetch) | set(write)):
    if flt and not any(f in key[0] for f in flt):
        continue
    f, w = fetch.get(key, [0, 0]), write.get(key, [0, 0])
    fk = f[0] / f[1] if f[1] else 0.0
    wk = w[0] / w[1] if w[1] else 0.0
    out["%s|grid=%s" % key] = {"fetch_size_kb_raw": round(fk, 1), "write_size_kb": round(wk, 1),
                               "bytes_per_launch": int((2 * fk + wk) * 1024), "launches_averaged": max(f[1], w[1])}
print(json.dumps(out, indent=1))
