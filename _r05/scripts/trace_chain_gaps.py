"""Dispatch gaps inside the graphed step, from a rocprofv3 kernel trace (see trace_gaps.py): for every kernel the time between
the end of the previous kernel ON THE SAME QUEUE and its own start, over six steps: median / mean / sum per step and queue."""
import sys, csv, glob, collections, statistics
root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
steps = [i for i, r in enumerate(rows) if "nchw_split_kernel" in r[2]]
print("columns: Queue_Id / Stream_Id of the first rows:", rows[0][3:], "| kernels", len(rows), "steps", len(steps))
a, b = steps[8], steps[14]
seg = rows[a:b]
byq = collections.defaultdict(list)
for s, e, nm, q, st in seg:
    byq[(q, st)].append((s, e, nm))
for key, ks in sorted(byq.items()):
    gaps = [ks[i][0] - ks[i - 1][1] for i in range(1, len(ks))]
    pos = [g for g in gaps if 0 <= g < 100000]
    neg = sum(1 for g in gaps if g < 0)
    if not pos:
        continue
    print("queue %s stream %s: %d kernels in 6 steps, busy %.2f ms/step; gaps to the previous kernel of the queue: median %.1f us, mean %.1f us, sum %.2f ms/step, "
          "%d overlapping (started before the previous one ended); <3us: %d, 3-6: %d, 6-12: %d, >12: %d" %
          (key[0], key[1], len(ks), sum(e - s for s, e, _ in ks) / 6e6, statistics.median(pos) / 1e3, statistics.mean(pos) / 1e3, sum(pos) / 6e6, neg,
           sum(g < 3000 for g in pos), sum(3000 <= g < 6000 for g in pos), sum(6000 <= g < 12000 for g in pos), sum(g >= 12000 for g in pos)))
# when each queue starts and ends inside a step (ms from the step's first kernel), and what the other queue still runs after it
for a_, b_ in list(zip(steps[:-1], steps[1:]))[8:11]:
    seg = rows[a_:b_]; t0 = seg[0][0]
    qs = collections.defaultdict(list)
    for s, e, nm, q, st in seg:
        qs[q].append((s, e, nm))
    ends = {q: max(e for _, e, _ in v) for q, v in qs.items()}
    print("step: " + " | ".join("queue %s: %d kernels, first start %.3f, last end %.3f ms" % (q, len(v), (min(s for s, _, _ in v) - t0) / 1e6, (ends[q] - t0) / 1e6)
                                for q, v in sorted(qs.items())))
    first_done = min(ends.values())
    tail = [(s, e, nm) for s, e, nm, q, st in seg if e > first_done]
    print("  after the first queue is done: %d kernels, %.3f ms: %s" % (len(tail), (max(e for _, e, _ in tail) - first_done) / 1e6,
          ", ".join("%s %.0f" % (nm.split("(")[0].split("::")[-1][:28], (e - s) / 1e3) for s, e, nm in tail[:40])))
a_ = steps[9]
print("the first kernels of a step (start offset us, duration us, queue, name):")
for s, e, nm, q, st in rows[a_:a_ + 44]:
    print("  %8.1f %7.1f  q%s  %s" % ((s - rows[a_][0]) / 1e3, (e - s) / 1e3, q, nm[:100]))
# per kernel name: launches per step, mean duration, ms per step (over the six steps above)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, nm, q, st in rows[steps[8]:steps[14]]:
    agg[nm][0] += 1; agg[nm][1] += e - s
print("kernels of the step by time (launches per step, mean us, ms per step):")
for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:48]:
    print("  %5.1f x %7.1f us = %6.3f ms  %s" % (c / 6, t / c / 1e3, t / 6e6, nm[:120]))
