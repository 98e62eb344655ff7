"""Top kernels of a rocprofv3 --kernel-trace --stats run, per training step.
   python3 scripts/kstats_top.py <dir or kernel_stats.csv> <steps> [rows]"""
import csv, os, sys
p, steps = sys.argv[1], float(sys.argv[2])
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
if os.path.isdir(p):
    p = [os.path.join(d, f) for d, _, fs in os.walk(p) for f in fs if f.endswith("kernel_stats.csv")][0]
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# %s: %.2f ms of kernel time per step over %.0f steps" % (p, tot / steps / 1e6, steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:rows_n]:
    print("%7.3f ms/step %7.1f launches/step  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, float(r["Calls"]) / steps,
                                                               float(r["AverageNs"]) / 1e3, r["Name"].replace("wcmc::", "")[:110]))
