set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_i
python3 bench.py --steps 30 --warmup 5 > $R/gpurun_out/bench_r1_final8.log 2>&1
tail -1 $R/gpurun_out/bench_r1_final8.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1r -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_r1r.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1s -- python3 $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_prof_r1s.log 2>&1
cd $R
bash scripts/pmc.sh i > $R/gpurun_out/pmc_i/run.log 2>&1
tail -3 $R/gpurun_out/pmc_i/run.log
find $R/gpurun_out/prof_r1r $R/gpurun_out/prof_r1s -name "*kernel_stats.csv" | head
