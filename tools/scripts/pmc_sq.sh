# SQ counters of one eager bench pass, per (kernel, grid): bash tools/scripts/pmc_sq.sh <match> (on the GPU box)
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"   # the repo root: gpurun exports it; else derived from this script's path
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/pmc_sq"
mkdir -p $O
cd "$R"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o x -- python3 bench.py --steps 1 --warmup 1 --median-steps 0 --no-graph --quick-cpu-baseline --no-segments --no-loss-check --no-roofline > $O/log$i.txt 2>&1 || { tail -5 $O/log$i.txt; echo "pass $i failed"; }
done
python tools/pmc_summary.py $O/p1 $O/p2 $O/p3 $O/p4 --match "$1" > $O/summary_$1.txt
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
cat $O/summary_$1.txt
