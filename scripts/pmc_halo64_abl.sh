# instruction counts of the KPCN 5x5 forward kernel's parts: rocprofv3 --pmc on the debug library's ablation instances
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export WCMC_DEBUG_LIB=1
for ab in 0 32 1 8; do
  export WCMC_DEBUG_ABLATE=$ab
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_h64/ab$ab -- python3 $R/scripts/one_halo64.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_h64/ab$ab/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_halo64" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print("ablate=$ab", k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
done
