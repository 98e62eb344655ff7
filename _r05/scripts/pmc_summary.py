"""Per-kernel means of the rocprofv3 --pmc passes collected by scripts/pmc.sh.
   python3 scripts/pmc_summary.py gpurun_out/pmc_<tag> > profiles/rNN_pmc_summary.json

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section); both counters are in KB."""
import sys, os, csv, glob, json, collections

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values (per dispatch)
for f in glob.glob(os.path.join(root, "p*", "*", "*counter_collection.csv")):
    per = collections.defaultdict(dict)
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"]
            if "wcmc::" not in k:
                continue
            per[(k, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            per[(k, r["Dispatch_Id"])]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (k, _), d in per.items():
        for c, v in d.items():
            acc[k][c].append(v)

def mean(x):
    return sum(x) / len(x) if x else None

out = {}
for k, d in sorted(acc.items()):
    m = {c: mean(v) for c, v in d.items()}
    g = lambda c: m.get(c)
    e = {"launch_us_avg": g("_us"), "FETCH_SIZE_KB_raw": g("FETCH_SIZE"), "WRITE_SIZE_KB_raw": g("WRITE_SIZE")}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        e["hbm_bytes_per_launch_corrected"] = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        e["L2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CYCLES"):
        # MFMA-busy cycles are summed over the 1024 SIMDs; SQ_BUSY_CYCLES over the 32 shader engines' SQs
        e["mfma_busy_cycles_per_simd"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0
    if g("SQ_WAVE_CYCLES"):
        e["wave_wait_any_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") is not None else None
        e["wave_wait_inst_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_INST_ANY") is not None else None
    if g("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA") if g("SQ_INSTS_VALU") is not None else None
        e["lds_insts_per_mfma"] = g("SQ_INSTS_LDS") / g("SQ_INSTS_MFMA") if g("SQ_INSTS_LDS") is not None else None
        e["vmem_rd_insts_per_mfma"] = g("SQ_INSTS_VMEM_RD") / g("SQ_INSTS_MFMA") if g("SQ_INSTS_VMEM_RD") is not None else None
    if g("GRBM_GUI_ACTIVE"):
        e["gui_active_cycles"] = g("GRBM_GUI_ACTIVE")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            pass
    out[k] = e
# MFMA-busy fraction needs GRBM_GUI_ACTIVE from another pass: combine here
for k, e in out.items():
    if e.get("mfma_busy_cycles_per_simd") is not None and e.get("gui_active_cycles"):
        e["mfma_busy_frac"] = e["mfma_busy_cycles_per_simd"] / (e["gui_active_cycles"] / 8.0)   # GUI_ACTIVE sums the 8 XCDs
json.dump(out, sys.stdout, indent=1)
