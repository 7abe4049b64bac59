cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6s18
cd /tmp && export TMPDIR=/tmp
for cfg in "timit 1" "slibri 1"; do
  set -- $cfg
  rm -rf /tmp/tri_$1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tri_$1 -o p -- python $GRAFT_REPO_ROOT/tools/infer_profile.py $1 $2 > $GRAFT_REPO_ROOT/gpurun_out/r6s18/$1.log 2>&1
  f=$(find /tmp/tri_$1 -name "*kernel_trace.csv" | head -1)
  python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r6s18/$1_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows))
# the last call: kernels after the last-but-one decode kernel
names = [e[2] for e in ev]
marks = [i for i, n in enumerate(names) if "ctc_beam" in n or "decode" in n.lower()]
last = marks[-1]
prev = max([m for m in marks if m < last - 5] or [0])
seg = ev[prev + 1:last + 1]
t0 = seg[0][0]
print("call: %d kernels, %.3f ms" % (len(seg), (seg[-1][1] - t0) / 1e6))
le = {}
for s, e, n, q in seg:
    gap = (s - le[q]) / 1e3 if q in le else 0.0
    le[q] = e
    print("%9.1f us +%8.1f us q%-2s gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]))
PY
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/r6s18/$1.log
done
