# Evidence behind profiles/: the bench lines, the two rocprofv3 kernel-stats runs (graphed, eager) and the PMC passes.
#   gpurun -- 'bash scripts/refresh_profiles.sh r05'      (then copy the summaries from gpurun_out/<tag>/ into profiles/)
set -x
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O/pmc
# the PMC passes first: the bench line reads its `traffic` from profiles/<tag>_pmc_summary.json
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc/p$i -- python3 $R/scripts/bench_kernels.py 3 > $O/pmc/log$i.txt 2>&1 || echo "pass $i failed"
done
cd $R
python3 scripts/pmc_summary.py $O/pmc > $O/pmc_summary.json
cp $O/pmc_summary.json $R/profiles/${TAG}_pmc_summary.json
python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_line.log 2>&1
tail -1 $O/bench_line.log | cut -c1-200
python3 $R/bench.py --steps 20 --warmup 5 --precision bf16x3 --no-cpu-baseline > $O/bench_line_bf16x3.log 2>&1
tail -1 $O/bench_line_bf16x3.log | cut -c1-200
python3 $R/bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline > $O/bench_line_fp32.log 2>&1
tail -1 $O/bench_line_fp32.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager -- python3 $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_eager_under_rocprof.log 2>&1
cd $R
python3 scripts/trace_gaps.py $O/prof_graph > $O/step_trace_gaps.txt 2>&1
python3 -m pytest tests/test_gpu_bench_config.py -q -m gpu > $O/bench_config_test.log 2>&1; tail -1 $O/bench_config_test.log
cp gpurun_out/bench_config_parity_device.txt $O/bench_config_parity_device.txt
set +x
for i in 1 2 3; do python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['class'], d['roofline']['frac'], 'long', d['value_long']['value'], 'c2', d['c2']['value'], {k: v['value'] for k, v in d['other_precisions'].items()}, 'multi_rank_path', d['multi_rank_path']['value'], d['multi_rank_path']['tail_ms'], 'overlap_allreduce', d['multi_rank_path']['overlap_allreduce']['value'])"; done > $O/bench_runs.txt 2>&1
set -x
find $O/prof_graph $O/prof_eager -name "*kernel_stats.csv" | head
# keep the merge under the 64 MiB limit: only the stats / counter CSVs go home
find $O -name "*kernel_trace.csv" -delete
find $O/pmc -name "*counter_collection.csv" -size +8M -delete
du -sh $O
