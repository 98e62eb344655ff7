# Same-box evidence for DESIGN 7.1 (what moves the two-stream step and what does not):
#   gpurun -- 'bash scripts/evidence_step_ab.sh'      -> gpurun_out/r06_step_ab.txt, gpurun_out/r06_step_kernels.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
{
echo "# two-stream captured step, U-Net 3x3 kernel of round 6 off / on (debug library, alternating order)"
python3 scripts/ab_step_switch.py ENV:WCMC_HALO3 3 2>/dev/null | grep -v FeatureMSE
echo "# the same with the halves in series on one stream"
AB_SERIAL=1 python3 scripts/ab_step_switch.py ENV:WCMC_HALO3 2 2>/dev/null | grep -v FeatureMSE
echo "# wave-private final-chain forward, 16 pixels per wave (100 VGPRs) against the shipped kernel"
python3 scripts/ab_step_switch.py ENV:WCMC_F2W=0,1 2 2>/dev/null | grep -v FeatureMSE
echo "# ... 32 pixels per wave (192 VGPRs)"
python3 scripts/ab_step_switch.py ENV:WCMC_F2W=0,2 2 2>/dev/null | grep -v FeatureMSE
echo "# clock and power while the step replays; all-zero parameters against the drawn ones"
python3 scripts/probe_power.py 2>/dev/null | grep -v FeatureMSE
if [ -d _r05 ]; then
echo "# round 5's tree against this one, alternating processes (value, ms per step, long run, configs[1])"
bash scripts/ab_trees.sh _r05 . 3
fi
} > $O/r06_step_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_full -- python3 $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline > $O/trace_full.log 2>&1
cd $R
python3 scripts/trace_gaps.py $O/trace_full > $O/r06_step_kernels.txt 2>&1
rm -rf $O/trace_full
tail -12 $O/r06_step_ab.txt
