"""Idle gaps of one replayed step: python tools/step_gaps.py <trace dir> [min_gap_us]
For the last replayed step: the device-wide busy time (union of all kernel intervals), and every interval longer than min_gap_us
in which NO kernel runs on any stream, with the kernels before and after it."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ming = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "opt_adamw_factored" in r["Kernel_Name"] or ("opt_adamw_kernel" in r["Kernel_Name"] and not any("opt_adamw_factored" in x["Kernel_Name"] for x in rows)):
        steps.append(cur); cur = []
st = steps[-2]
name = lambda r: re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"])))[:60]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r), r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in st)
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
busy, end, gaps = 0, iv[0][0], []
last = iv[0]
for s, e, n, q in iv:
    if s > end:
        if (s - end) / 1e3 >= ming:
            gaps.append(((s - end) / 1e3, (end - t0) / 1e6, last[2], n))
        busy += 0
        end = e; last = (s, e, n, q)
    elif e > end:
        end = e; last = (s, e, n, q)
tot_gap = sum(g[0] for g in gaps)
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(iv)} launches, queues {sorted(set(q for *_, q in iv))}; gaps >= {ming} us: {len(gaps)}, {tot_gap / 1e3:.3f} ms in all")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]:7.1f} us at {g[1]:7.3f} ms   after {g[2]}   before {g[3]}")
# per-queue busy time
per = {}
for s, e, n, q in iv:
    per[q] = per.get(q, 0) + (e - s)
print({q: round(v / 1e6, 3) for q, v in per.items()})
