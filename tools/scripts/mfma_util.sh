# MFMA utilisation / stall / HBM-traffic counters of the ten kernels with the most time in a step:
#   bash tools/scripts/mfma_util.sh <tag>        (on the GPU box; separate --pmc passes, no tracing combined with counters)
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${1:-mfma_util}"
mkdir -p $O
cd "$R"
B="python3 bench.py --one-stream --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check"
python tools/step_kernel_counts.py --help > /dev/null 2>&1 || true
# graph replay: kernel durations + the ordered GEMM list (algorithmic bytes / flops per kernel name)
timeout -k 10 600 $B --steps 5 --warmup 3 --dump-gemm-order $O/order.json > $O/bench.json 2> $O/bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o x -- $B --steps 4 --warmup 2 --no-roofline > $O/trace.txt 2>&1
echo "trace done"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  CSTS_GROUP_WGRADS=1 timeout -k 10 500 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o x -- $B --steps 1 --warmup 1 --no-graph --no-roofline > $O/pmc$i.txt 2>&1 || { tail -5 $O/pmc$i.txt; echo "pmc pass $i failed"; }
  echo "pmc pass $i done"
done
python tools/mfma_util.py $O/trace $O/order.json $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > $O/mfma_util.txt
python tools/pmc_traffic.py $O/p4 $O/p5 $O/pmc_traffic.json --steps-profiled 2 > $O/pmc_traffic_top.txt      # 1 warm-up + 1 timed eager step per pass
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/step_kernel_counts.py $O/trace 200 > $O/step_inventory.txt
python tools/sum_kernel_trace.py $O/trace _kernel > $O/kernel_by_grid.txt
rm -rf $O/trace $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
cat $O/mfma_util.txt
