set -euo pipefail
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 20 --warmup 5 --median-steps 30 --no-cpu-baseline --no-roofline --no-segments --no-loss-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['median_step_ms'])"; }
for i in 1 2; do
  unset CSTS_HIP_LIB; run base
  for v in v1 v2 v3; do export CSTS_HIP_LIB=tools/diag/lib_$v.so; run $v; done
done
