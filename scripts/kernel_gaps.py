"""Reads a rocprofv3 --kernel-trace CSV: per kernel name, count / mean duration; and for the LAST `n` dispatches in
start order, start, end, gap to the previous end, queue.   python scripts/kernel_gaps.py <kernel_trace.csv> [n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in by.items():
    print("%-72s n=%5d mean %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
t0 = int(rows[-n]["Start_Timestamp"])
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("q%-3s %-40s start %8.1f end %8.1f dur %6.1f gap_from_prev_end %6.1f" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:40], s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev is not None else 0))
    prev = e
# device-side span of the filter loop: first start to last end of the step kernels of the LAST run in the trace
steps = [r for r in rows if "pf_step_kernel" in r["Kernel_Name"] or "metropolis" in r["Kernel_Name"]]
if steps:
    half = steps[len(steps) // 2:]  # (the script runs the filter twice: second run)
    span = int(half[-1]["End_Timestamp"]) - int(half[0]["Start_Timestamp"])
    per_queue = collections.Counter(r.get("Queue_Id", "?") for r in half)
    nsteps = max(per_queue.values())
    print("loop span (second run): %.2f ms over %d steps = %.1f us per step; queues %s" % (span / 1e6, nsteps, span / 1e3 / nsteps, dict(per_queue)))
