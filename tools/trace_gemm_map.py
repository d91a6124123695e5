"""Per-shape GEMM launch durations INSIDE THE HIP-GRAPH REPLAY: zip the ordered launch list of one step
(bench.py --one-stream --dump-gemm-order order.json) with a rocprofv3 --kernel-trace of `bench.py --one-stream` (one stream,
so the trace's start order is the issue order).

    python tools/trace_gemm_map.py <trace dir> <order.json> [<out.txt>]

A step in the trace = the kernels between two opt_adamw_kernel launches; the last `--steps` steps are graph replays.  Per
(layout, M, N, K, kernel): launches per step, median / min duration in the replay, the same launch timed by HIP events in the
eager instrumented pass, and the median idle gap in front of it (start - previous kernel's end)."""
import collections
import csv
import glob
import json
import re
import statistics
import sys


def short(n):
    m = re.search(r"(gemm\w*_kernel)\s*(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")).replace(" ", "") if m else None


def main():
    tdir, order_path = sys.argv[1], sys.argv[2]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
    f = glob.glob(tdir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    order = json.load(open(order_path))
    want = [o["kernel"].replace(" ", "") for o in order]
    # cut into steps at the optimizer kernel
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "opt_adamw_kernel" in r["Kernel_Name"]:
            steps.append(cur)
            cur = []
    per = collections.defaultdict(lambda: {"us": [], "gap": [], "eager": [], "n": 0})
    used = 0
    step_ms, gemm_ms, gap_ms = [], [], []
    for st in steps[-nsteps:]:
        seq = []
        prev_end = None
        for r in st:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            nm = short(r["Kernel_Name"])
            if nm is not None and "finish" not in r["Kernel_Name"]:
                seq.append((nm, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0.0))
            prev_end = e if prev_end is None else max(prev_end, e)
        if [n for n, _, _ in seq] != want:
            print(f"step skipped: {len(seq)} GEMM launches in the trace, {len(want)} in the order file "
                  f"(first mismatch at {next((i for i, (a, b) in enumerate(zip([n for n, _, _ in seq], want)) if a != b), None)})", file=sys.stderr)
            continue
        used += 1
        step_ms.append((int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])) / 1e6)
        gemm_ms.append(sum(d for _, d, _ in seq) / 1e3)
        gap_ms.append(sum(g for _, _, g in seq) / 1e3)
        for (nm, d, g), o in zip(seq, order):
            k = (o["layout"], o["M"], o["N"], o["K"], o["split"], nm, o["flop"], o["bytes"] + o["epilogue_bytes"])
            per[k]["us"].append(d)
            per[k]["gap"].append(g)
            per[k]["eager"].append(o["us_eager_events"])
    if not used:
        sys.exit("no replay step matched the order file")
    for k in per:
        per[k]["n"] = len(per[k]["us"]) // used
    out.write(f"graph replay, one stream: step {statistics.median(step_ms):.3f} ms; GEMM kernels {statistics.median(gemm_ms):.3f} ms per step over "
              f"{len(want)} launches; idle gaps in front of GEMM launches {statistics.median(gap_ms):.3f} ms per step ({used} replayed steps)\n")
    out.write("layout M N K split calls/step  replay_med_us replay_min_us eager_event_us gap_us  TFLOP/s GB/s(A+B+C+epilogue operands)  ms/step  kernel\n")
    for k, v in sorted(per.items(), key=lambda kv: -statistics.median(kv[1]["us"]) * kv[1]["n"]):
        lay, M, N, K, sp, nm, fl, by = k
        med = statistics.median(v["us"])
        out.write(f"{lay} {M} {N} {K} {sp} {v['n']:3d}  {med:8.1f} {min(v['us']):8.1f} {statistics.median(v['eager']):8.1f} {statistics.median(v['gap']):5.1f}  "
                  f"{fl / med / 1e6:7.1f} {by / med / 1e3:7.0f}  {med * v['n'] / 1e3:6.3f}  {nm}\n")


main()
