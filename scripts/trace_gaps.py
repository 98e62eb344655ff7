"""Idle time and overlap inside the graphed step, from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline
   python3 scripts/trace_gaps.py OUT
Takes the last replays of the captured step (recognised by their period), prints per step: wall time, union of the kernel
intervals (= time with at least one kernel running), the idle gaps, and the time with two or more kernels in flight."""
import sys, csv, glob, collections
root = sys.argv[1]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# steps: the captured step begins with the one nchw_split launch (the shared `paths` conversion)
steps = [i for i, r in enumerate(rows) if "nchw_split_kernel" in r[2]]
print("kernels %d, steps found %d" % (len(rows), len(steps)))
for a, b in list(zip(steps[:-1], steps[1:]))[8:14]:
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]                      # first kernel of this step -> first kernel of the next
    ev = []
    for s, e, *_ in seg:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    ev.append((t1, 0))
    busy = multi = 0; depth = 0; last = ev[0][0]; gaps = []
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: multi += t - last
        if depth == 0 and t > last: gaps.append((t - last, last - t0))
        depth += d; last = t
    gaps.sort(reverse=True)
    ksum = sum(e - s for s, e, *_ in seg)
    print("step: %d kernels, wall %.2f ms, busy (>=1 kernel) %.2f ms, idle %.2f ms in %d gaps (largest: %s us), >=2 kernels in flight %.2f ms, sum of kernel times %.2f ms" %
          (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), ", ".join("%.0f@%.1fms" % (g / 1e3, at / 1e6) for g, at in gaps[:6]), multi / 1e6, ksum / 1e6))
# what runs ALONE (one kernel in flight) and for how long, per kernel name, over the same steps: the serial part of the schedule
solo = collections.Counter(); tot = collections.Counter(); nsteps = 0
for a, b in list(zip(steps[:-1], steps[1:]))[8:14]:
    seg = rows[a:b]; nsteps += 1
    ev = []
    for i, (s, e, *_) in enumerate(seg):
        ev.append((s, 1, i)); ev.append((e, -1, i))
        tot[seg[i][2]] += e - s
    ev.sort()
    live = set(); last = ev[0][0]
    for t, d, i in ev:
        if len(live) == 1:
            solo[seg[next(iter(live))][2]] += t - last
        (live.add if d > 0 else live.discard)(i); last = t
if nsteps:
    print("time with exactly ONE kernel in flight, by kernel (ms per step alone | ms per step in total), %d steps:" % nsteps)
    for nm, v in solo.most_common(14):
        print("  %6.3f | %6.3f  %s" % (v / nsteps / 1e6, tot[nm] / nsteps / 1e6, nm[:110]))
    print("  %6.3f alone in total" % (sum(solo.values()) / nsteps / 1e6))
    cnt = collections.Counter()
    for a, b in list(zip(steps[:-1], steps[1:]))[8:14]:
        for r in rows[a:b]:
            cnt[r[2]] += 1
    print("every kernel of the step (launches per step, ms per step, mean us), %d steps; under the profiler the two half-step graphs run in series:" % nsteps)
    for nm, v in tot.most_common():
        print("  %5.1f  %6.3f  %7.1f  %s" % (cnt[nm] / nsteps, v / nsteps / 1e6, v / cnt[nm] / 1e3, nm[:120]))
    print("  %5.1f  %6.3f  in total" % (sum(cnt.values()) / nsteps, sum(tot.values()) / nsteps / 1e6))
if len(steps) > 12:
    a, b = steps[10], steps[11]
    a = max(0, b - 30)
    print("the last kernels of one step and the first of the next (start offset us, duration us, name):")
    for s_, e_, nm, *_ in rows[a:a + 60]:
        print("  %8.1f %7.1f  %s" % ((s_ - rows[a][0]) / 1e3, (e_ - s_) / 1e3, nm[:90]))
