"""Timeline of the LAST train step in a rocprofv3 --kernel-trace csv: which kernels ran when, on which queue.
    python tools/trace_timeline.py kernel_trace.csv [marker-kernel-substring] [min_us]
The step boundary is the last-but-one launch of the marker kernel (default: clip_sgd_kernel).  Prints every kernel of at
least min_us (default 30) with its start offset, duration and queue, the busy time per queue, and the main queue's idle
gaps -- what the critical path of a step with side-stream work (bidirectional stacks) looks like."""
import csv
import sys

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "clip_sgd_kernel"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
rows = list(csv.DictReader(open(path)))
key_s = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "start_timestamp"
key_e = "End_Timestamp" if "End_Timestamp" in rows[0] else "end_timestamp"
key_n = "Kernel_Name" if "Kernel_Name" in rows[0] else "kernel_name"
key_q = "Queue_Id" if "Queue_Id" in rows[0] else "queue_id"
ev = sorted(((int(r[key_s]), int(r[key_e]), r[key_n], r[key_q]) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if marker in e[2]]
lo, hi = marks[-2] + 1, marks[-1] + 1
step = ev[lo:hi]
t0 = step[0][0]
print("step: %d kernels, %.3f ms" % (len(step), (step[-1][1] - t0) / 1e6))
busy = {}
for s, e, n, q in step:
    busy[q] = busy.get(q, 0) + (e - s)
print("busy per queue (ms):", {q: round(v / 1e6, 3) for q, v in busy.items()})
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
last_end = {}
for s, e, n, q in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    if (e - s) / 1e3 >= min_us or gap >= min_us:
        print("%9.1f us  +%8.1f us  q%-3s gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, gap, short(n)))
