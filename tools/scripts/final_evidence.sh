# The whole evidence set of the final build in TWO GPU calls (a gpurun call is capped at 20 minutes):
#   bash tools/scripts/final_evidence.sh counters [tag]   kernel trace + the five --pmc passes + per-shape GEMM replay trace
#   (copy gpurun_out/<tag>/pmc_traffic.json to profiles/pmc_traffic.json in the build container)
#   bash tools/scripts/final_evidence.sh lines [tag]      every bench line of DESIGN section 6; the default line reads that traffic file
# Run `python tools/build_stamp.py` in the build container first (the stamp file travels with the snapshot).  Everything lands under
# gpurun_out/<tag>/ with the names profiles/ uses (r5_final_*), each text file headed by the stamp (source commit + library sha256).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
PHASE="${1:-counters}"
TAG="${2:-r5_final}"
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd "$R"
STAMP=$(python tools/build_stamp.py | tail -1)
echo "stamp: $STAMP"
if [ "$PHASE" = counters ]; then
bash tools/scripts/mfma_util.sh ${TAG}_m > "$O/mfma_util.log" 2>&1 || echo "mfma_util.sh failed"
echo "counters done"
bash tools/scripts/gemm_replay_trace.sh ${TAG}_g > "$O/gemm_replay.log" 2>&1 || echo "gemm_replay_trace.sh failed"
echo "gemm replay done"
M="$R/gpurun_out/${TAG}_m"; G="$R/gpurun_out/${TAG}_g"
head1() { { echo "# build stamp: $STAMP"; cat "$1"; } > "$2"; }
[ -f "$M/kernel_stats.csv" ] && cp "$M/kernel_stats.csv" "$O/${TAG}_kernel_stats.csv"
[ -f "$M/mfma_util.txt" ] && head1 "$M/mfma_util.txt" "$O/${TAG}_mfma_util.txt"
[ -f "$M/pmc_traffic_top.txt" ] && head1 "$M/pmc_traffic_top.txt" "$O/${TAG}_pmc_traffic_top.txt"
[ -f "$M/step_inventory.txt" ] && head1 "$M/step_inventory.txt" "$O/${TAG}_step_inventory.txt"
[ -f "$M/kernel_by_grid.txt" ] && head1 "$M/kernel_by_grid.txt" "$O/${TAG}_kernel_by_grid.txt"
[ -f "$M/pmc_traffic.json" ] && cp "$M/pmc_traffic.json" "$O/pmc_traffic.json"
[ -f "$G/gemm_replay_shapes.txt" ] && head1 "$G/gemm_replay_shapes.txt" "$O/${TAG}_gemm_replay_shapes.txt"
ls "$O"
exit 0
fi
# phase "lines": the default line reads profiles/pmc_traffic.json collected by phase "counters" (same build, same launch population)
head1() { { echo "# build stamp: $STAMP"; cat "$1"; } > "$2"; }
bash tools/scripts/final_lines.sh ${TAG}_l > "$O/final_lines.log" 2>&1 || echo "final_lines.sh failed"
Lr="$R/gpurun_out/${TAG}_l"
for f in bench bench_fwd bench_T32_b2 bench_crop224 bench_rehearse_dist bench_rehearse_dist_cut0 bench_fp16 bench_fp16_T32_b2 bench_fp16_fwd bench_fp32; do
  [ -f "$Lr/$f.json" ] && cp "$Lr/$f.json" "$O/${TAG}_$f.json"
done
[ -f "$Lr/op_breakdown.txt" ] && head1 "$Lr/op_breakdown.txt" "$O/${TAG}_op_breakdown.txt"
tail -12 "$O/final_lines.log"
ls "$O"
