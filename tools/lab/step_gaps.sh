set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${1:-step_gaps}"
mkdir -p "$O"
cd "$R"
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o x -- python3 bench.py --steps 6 --warmup 2 --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check --no-roofline > "$O/prof_bench.txt" 2>&1
python tools/step_gaps.py "$O/prof" 2.0 > "$O/step_gaps.txt"
head -3 "$(find $O/prof -name '*kernel_trace.csv' | head -1)" > "$O/trace_head.txt"
rm -rf "$O/prof"
cat "$O/step_gaps.txt"
