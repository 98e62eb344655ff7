"""Kernels before / after every launch whose name contains <pattern> in a rocprofv3 kernel trace (last step only).
   python3 scripts/trace_around.py <dir> <pattern> [context]"""
import csv, os, sys
d, pat = sys.argv[1], sys.argv[2]
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
f = [os.path.join(r, x) for r, _, fs in os.walk(d) for x in fs if x.endswith("kernel_trace.csv")][0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-700:]
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        for j in range(max(0, i - ctx), min(len(rows), i + ctx + 1)):
            q = rows[j]
            print("%s %8.1f us  %s" % ("->" if j == i else "  ", (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e3, q["Kernel_Name"].replace("wcmc::", "")[:100]))
        print()
