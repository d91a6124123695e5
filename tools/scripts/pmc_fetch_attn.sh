# FETCH_SIZE of the attention kernels for the current library and for CSTS_HIP_LIB=$1 (eager single-stream step, rocprofv3 --pmc)
set -euo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
O="$R/gpurun_out/${2:-pmc_attn}"; mkdir -p "$O"; cd "$R"
B="python3 bench.py --one-stream --median-steps 0 --no-cpu-baseline --no-segments --no-loss-check --steps 1 --warmup 1 --no-graph --no-roofline"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/new -o x -- $B > $O/new.txt 2>&1
python tools/pmc_summary.py $O/new --match attn_ > $O/new_fetch.txt
export CSTS_HIP_LIB=$1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/old -o x -- $B > $O/old.txt 2>&1
python tools/pmc_summary.py $O/old --match attn_ > $O/old_fetch.txt
rm -rf $O/new $O/old
paste $O/old_fetch.txt $O/new_fetch.txt | grep -v "^$" | head -60
