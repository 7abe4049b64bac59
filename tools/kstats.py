"""Print the top rows of a rocprofv3 *_kernel_stats.csv (names contain commas: use the csv module, not awk)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.3f ms" % (tot / 1e6))
for r in rows[:n]:
    print("%-64s %6s %9.3f ms %5.1f%% avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
