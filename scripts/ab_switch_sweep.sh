# In-step re-test of the kernel switches whose defaults were chosen on ISOLATED launch times in rounds 2-5 (debug library; off = "0"):
#   gpurun -- 'bash scripts/ab_switch_sweep.sh'  -> gpurun_out/r06_switch_sweep.txt
R=$GRAFT_REPO_ROOT; cd $R
for sw in WCMC_HALO64_CS32 WCMC_WGRAD_ROWS8 WCMC_WGRAD_ROWS_3X3 WCMC_WGRAD_ROWS_1X1 WCMC_IGEMM_PW WCMC_PW_TAIL WCMC_WGRAD_ROWS8_XE WCMC_HALO_NB=3,2 WCMC_WGRAD_ROWS8_PRIO=8,0 WCMC_KA_TILE=0,1; do
  python3 scripts/ab_step_switch.py ENV:$sw 2 2>/dev/null | grep -E "^WCMC" 
done > gpurun_out/r06_switch_sweep.txt 2>&1
cat gpurun_out/r06_switch_sweep.txt
