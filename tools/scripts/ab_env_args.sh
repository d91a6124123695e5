# same-box A/B of an environment switch with extra bench.py arguments: bash tools/scripts/ab_env_args.sh VAR a b rounds -- <bench args>
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
VAR=$1; A=$2; B=$3; N=$4; shift 5
cd "$R"
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --median-steps 30 --no-cpu-baseline --no-roofline --no-segments --no-loss-check "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['median_step_ms'])"
  done
done
