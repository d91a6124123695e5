set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/r2v2"
mkdir -p $O
cd "$R"
timeout -k 10 900 python bench.py --op-breakdown $O/op_breakdown.txt --dump-gemm $O/gemm_shapes.txt > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x -- python3 bench.py --steps 4 --warmup 2 --median-steps 0 --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/prof_bench.txt 2>&1
python tools/sum_kernel_trace.py $O/prof _kernel > $O/kernel_by_grid.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
echo "kernel stats done"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o x -- python3 bench.py --steps 2 --warmup 1 --median-steps 0 --no-graph --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/pmc_f.txt 2>&1
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o x -- python3 bench.py --steps 2 --warmup 1 --median-steps 0 --no-graph --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/pmc_w.txt 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic_eager.json
rm -rf $O/pmc_f $O/pmc_w
echo "eager pmc done"
CSTS_GROUP_WGRADS=1 timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o x -- python3 bench.py --steps 2 --warmup 1 --median-steps 0 --no-graph --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/pmc_f2.txt 2>&1
CSTS_GROUP_WGRADS=1 timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o x -- python3 bench.py --steps 2 --warmup 1 --median-steps 0 --no-graph --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/pmc_w2.txt 2>&1
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic_grouped.json
rm -rf $O/pmc_f $O/pmc_w
echo "grouped pmc done"
