# same-box sweep of an environment switch over several values: bash tools/scripts/ab_values.sh VAR rounds v1 v2 v3 ...
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
VAR=$1; N=$2; shift 2
cd "$R"
for i in $(seq 1 $N); do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --median-steps 30 --no-cpu-baseline --no-roofline --no-segments --no-loss-check 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['median_step_ms'])"
  done
done
