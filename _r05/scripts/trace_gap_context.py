"""The idle gaps of one graphed step with the kernels around them (rocprofv3 kernel trace of bench.py).
   python3 scripts/trace_gap_context.py <dir> [min gap us]"""
import sys, csv, glob
root = sys.argv[1]
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("wcmc::", ""), r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
steps = [i for i, r in enumerate(rows) if "nchw_split_kernel" in r[2]]
a, b = steps[-4], steps[-3]
seg = rows[a:b + 1]
t0 = seg[0][0]
busy_until = seg[0][1]
last = seg[0]
for r in seg[1:]:
    if r[0] > busy_until:
        gap = (r[0] - busy_until) / 1e3
        if gap >= ming:
            print("%7.2f ms  gap %5.1f us   after %-60s (%5.1f us, q %s)   before %-60s (q %s)" % ((busy_until - t0) / 1e6, gap, last[2][:60], (last[1] - last[0]) / 1e3, last[3], r[2][:60], r[3]))
    if r[1] > busy_until:
        busy_until = r[1]; last = r
