# per-shape GEMM durations inside the graph replay (one stream): order file from the instrumented pass + kernel trace
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${1:-gemm_replay}"
mkdir -p $O
cd "$R"
timeout -k 10 600 python bench.py --one-stream --steps 5 --warmup 3 --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check --dump-gemm-order $O/order.json > $O/bench_one_stream.json 2> $O/bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o x -- python3 bench.py --one-stream --steps 4 --warmup 2 --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/prof_bench.txt 2>&1
python tools/trace_gemm_map.py $O/prof $O/order.json $O/gemm_replay_shapes.txt --steps 3
python tools/sum_kernel_trace.py $O/prof _kernel > $O/kernel_by_grid.txt
rm -rf $O/prof
head -30 $O/gemm_replay_shapes.txt
