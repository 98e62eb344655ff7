"""Copy the summaries of a scripts/refresh_profiles.sh run (gpurun_out/<tag>/) into profiles/<tag>_*.
   python3 scripts/collect_profiles.py r04"""
import glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")

def last_json_line(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith('{"metric"')]
    assert lines, path
    return json.loads(lines[-1])

for log, out in (("bench_line.log", "bench_line.json"), ("bench_line_bf16x3.log", "bench_line_bf16x3.json"),
                 ("bench_line_fp32.log", "bench_line_fp32.json"), ("bench_under_rocprof.log", "bench_under_rocprof.json"),
                 ("bench_eager_under_rocprof.log", "bench_eager_under_rocprof.json")):
    with open(os.path.join(dst, "%s_%s" % (tag, out)), "w") as f:
        json.dump(last_json_line(os.path.join(src, log)), f, indent=1)
        f.write("\n")
for sub, out in (("prof_graph", "bench_kernel_stats.csv"), ("prof_eager", "bench_eager_kernel_stats.csv")):
    got = sorted(glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")), key=os.path.getmtime)
    assert got, sub                                       # (gpurun merges into gpurun_out/: an earlier call's file may still be there)
    shutil.copy(got[-1], os.path.join(dst, "%s_%s" % (tag, out)))
for name in ("pmc_summary.json", "step_trace_gaps.txt", "bench_config_parity_device.txt", "bench_runs.txt"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, "%s_%s" % (tag, name)))
line = last_json_line(os.path.join(src, "bench_line.log"))
print("%s: %.1f patches/s, roofline %s frac %.4f" % (tag, line["value"], line["roofline"]["class"], line["roofline"]["frac"]))
