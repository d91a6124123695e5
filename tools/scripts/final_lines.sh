# the bench lines DESIGN section 6 quotes, from the current build:  bash tools/scripts/final_lines.sh <tag>
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${1:-final}"
mkdir -p $O
cd "$R"
timeout -k 10 900 python bench.py --op-breakdown $O/op_breakdown.txt --dump-gemm $O/gemm_shapes.txt > $O/bench.json 2> $O/bench.err
echo "default done"; tail -c 300 $O/bench.json
timeout -k 10 300 python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --frames 32 --batch-per-gpu 2 --no-cpu-baseline > $O/bench_T32_b2.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --crop 224 --no-cpu-baseline > $O/bench_crop224.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --rehearse-dist --no-cpu-baseline > $O/bench_rehearse_dist.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --rehearse-dist --trunk-cut 0 --no-cpu-baseline > $O/bench_rehearse_dist_cut0.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --compute fp16 --no-cpu-baseline --no-roofline > $O/bench_fp16.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --compute fp16 --frames 32 --batch-per-gpu 2 --no-cpu-baseline --no-roofline > $O/bench_fp16_T32_b2.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --compute fp16 --mode fwd --no-cpu-baseline > $O/bench_fp16_fwd.json 2>> $O/bench.err
timeout -k 10 400 python bench.py --compute fp32 --no-cpu-baseline --no-roofline > $O/bench_fp32.json 2>> $O/bench.err
for f in bench_fwd bench_T32_b2 bench_crop224 bench_rehearse_dist bench_rehearse_dist_cut0 bench_fp16 bench_fp16_T32_b2 bench_fp16_fwd bench_fp32; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['unit'], d['ms_per_step'])"; done
