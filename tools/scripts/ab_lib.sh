# same-box A/B of two builds of the library on the whole step: bash tools/scripts/ab_lib.sh <old .so> [rounds]
OLD=$1; R=${2:-3}
cd $GRAFT_REPO_ROOT
for i in $(seq 1 $R); do
  for v in old new; do
    if [ $v = old ]; then export CSTS_HIP_LIB=$OLD; else unset CSTS_HIP_LIB; fi
    python bench.py --steps 20 --warmup 5 --median-steps 30 --no-cpu-baseline --no-roofline --no-segments --no-loss-check 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['median_step_ms'])"
  done
done
