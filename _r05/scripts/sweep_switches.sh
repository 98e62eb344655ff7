# The library's A/B switches re-measured on the graphed benchmark step (their defaults were chosen when the halves' backward passes ran in series):
#   gpurun -- 'bash scripts/sweep_switches.sh'   -> one line per setting (scripts/time_step_env.py), the default first and last
cd $GRAFT_REPO_ROOT
for e in "X=1" "WCMC_HALO64_PT3=0" "WCMC_HALO64_CS32=0" "WCMC_HALO_TH8_5X5=0" "WCMC_WGRAD_ROWS8=0" "WCMC_WGRAD_ROWS8_PRIO=0" "WCMC_WGRAD_ROWS_3X3=0" \
         "WCMC_FUSE_BIAS_GRAD=0" "WCMC_GATE_MASK=0" "WCMC_PACK_CHAIN=0" "WCMC_KA_TILE=1" "WCMC_DGRAD_AP1=0" "WCMC_FUSE_EMBED=0" "WCMC_FUSE_FINAL=0" "X=2"; do
  env $e python3 scripts/time_step_env.py $e 2>&1 | tail -1 | cut -c1-110
done
