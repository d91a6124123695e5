"""Summarise a rocprofv3 kernel_trace.csv: per-kernel buckets and the last step's per-launch detail.

usage: python tools/prof_summary.py <kernel_trace.csv> [--steps N] [--detail SUBSTR]
"""
import csv, sys, re, collections

def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]

def main():
    path = sys.argv[1]
    steps = 6
    detail = None
    if "--steps" in sys.argv: steps = int(sys.argv[sys.argv.index("--steps") + 1])
    if "--detail" in sys.argv: detail = sys.argv[sys.argv.index("--detail") + 1]
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        agg[k][0] += 1; agg[k][1] += d
    tot = sum(v[1] for v in agg.values())
    print(f"total kernel time {tot/1e3:.2f} ms over {steps} steps -> {tot/1e3/steps:.2f} ms/step")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{t/1e3/steps:8.3f} ms/step  {n/steps:7.1f} calls  {t/n:8.1f} us  {k}")
    if detail:
        sel = [r for r in rows if detail in r["Kernel_Name"]]
        per = len(sel) // steps
        g = collections.defaultdict(lambda: [0, 0.0])
        for r in sel[-per:]:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
            g[key][0] += 1; g[key][1] += d
        for k, (n, t) in sorted(g.items(), key=lambda kv: -kv[1][1]):
            print(f"{t:9.1f} us  x{n:3d}  {t/n:8.1f} us  {k}")

main()
