#!/usr/bin/env python3
"""Condenses a scripts/profile.sh output directory into a short markdown summary
(kernel names are truncated: torch's generator kernels have 4 KB names)."""
import csv
import glob
import sys


def short(n):
    return n if len(n) < 90 else n[:87] + "..."


def main(d):
    print("# rocprofv3 summary (%s)\n" % d.rstrip("/").split("/")[-1])
    for f in glob.glob(d + "/kt/**/*kernel_stats.csv", recursive=True):
        print("## --kernel-trace --stats (python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu: the driver's command)\n")
        print("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) < 0.05:
                continue
            print("| %s | %s | %.1f | %.1f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    # the 20 TIMED dispatches themselves: bench.py runs 595 settle + 5 warm-up launches of the headline kernel,
    # then the timed 20 (then the strong-scaling leg's 800); rocprof's duration excludes the ~2.5 us between launches
    for f in glob.glob(d + "/kt/**/*kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "cusmc_logpdf_nb4_asm" in r["Kernel_Name"] or "logpdf_mfma_kernel<4, true, false, 0, 1, false>" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        if len(rows) >= 620:
            win = rows[600:620]
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in win]
            span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e3 / len(win)
            print("\nThe 20 timed dispatches (#601-620 of the headline kernel): kernel duration avg %.2f us (min %.2f, max %.2f); "
                  "start-to-end span / 20 = %.2f us per launch including the gaps between launches -- the quantity "
                  "bench.py's HIP events measure (`roofline.kernel_ms`)." % (sum(dur) / len(dur), min(dur), max(dur), span))
            alld = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
            print("All %d dispatches of that kernel in the run: avg %.2f us." % (len(alld), sum(alld) / len(alld)))
    for f in glob.glob(d + "/configs/**/*kernel_stats.csv", recursive=True):
        print("\n## --kernel-trace --stats (python3 scripts/bench_configs.py: the other BASELINE configs)\n")
        print("| kernel | calls | avg us | min us | max us |\n|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            if "cusmc" not in r["Name"]:
                continue
            print("| %s | %s | %.1f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(d + "/%s/**/*counter_collection.csv" % counter.split("_")[0].lower(), recursive=True):
            vals = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and "cusmc" in r["Kernel_Name"]:
                    vals.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
            print("\n## --pmc %s (KiB per dispatch, raw counter)\n" % counter)
            for k, v in vals.items():
                print("- `%s`: mean %.1f KiB over %d dispatches = %.1f MB" % (k, sum(v) / len(v), len(v), sum(v) / len(v) * 1024 / 1e6))
    print("\nFETCH_SIZE on gfx950 reports half the bytes of a wide (16 B/lane) streaming read "
          "(MI355X_MICROARCH.md, HBM): double it before comparing with the algorithmic 512 MB; WRITE_SIZE is exact.")


if __name__ == "__main__":
    main(sys.argv[1])
