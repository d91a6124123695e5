# same-box A/B of two builds of the library on the whole step: bash tools/scripts/ab_lib.sh <old .so> [rounds]
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
OLD=$1; N=${2:-3}
cd "$R"
# csts_amd/lib.py refuses a library whose csts_abi_version() differs from the binding's: an older build that reads the
# argument structs with another layout fails loudly here instead of running wrong
for i in $(seq 1 $N); do
  for v in old new; do
    if [ $v = old ]; then export CSTS_HIP_LIB=$OLD; else unset CSTS_HIP_LIB; fi
    python bench.py --steps 20 --warmup 5 --median-steps 30 --no-cpu-baseline --no-roofline --no-segments --no-loss-check 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['median_step_ms'])"
  done
done
