export WCMC_HALO64_MIX=2
for h in 92 100 104 108 116 120 124; do
  python3 scripts/ab_step_switch.py ENV:WCMC_HALO64_NOMIX=0,$h 2 2>/dev/null | grep "WCMC_HALO64_NOMIX:" | sed "s/^/h=$h  /"
done
