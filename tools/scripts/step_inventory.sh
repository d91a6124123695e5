# per-step kernel inventory of the replayed HIP graph (launch counts and time by kernel name, last 3 replayed steps only -- the
# whole-process kernel_stats.csv also holds warm-up, capture-pass and one-time work):  bash tools/scripts/step_inventory.sh <tag>
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${1:-step_inventory}"
mkdir -p "$O"
cd "$R"
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o x -- python3 bench.py --steps 6 --warmup 2 --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check --no-roofline > "$O/prof_bench.txt" 2>&1
python tools/step_kernel_counts.py "$O/prof" 200 > "$O/step_inventory.txt"
rm -rf "$O/prof"
head -40 "$O/step_inventory.txt"
