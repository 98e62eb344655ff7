# Same-box comparison of two source trees' bench.py (alternating processes): bash scripts/ab_trees.sh <treeA> <treeB> [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for T in $A $B; do
    ( cd $T && python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', d['value'], d['ms_per_step'], 'long', d['value_long']['value'], 'c2', d['c2']['value'])" )
  done
done
