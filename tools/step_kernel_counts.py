"""Kernel launches and time per replayed step, by short kernel name: python tools/step_kernel_counts.py <trace dir> [top]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "opt_adamw_kernel" in r["Kernel_Name"]:
        steps.append(cur); cur = []
use = steps[-3:]
agg = collections.defaultdict(lambda: [0, 0.0])
for st in use:
    for r in st:
        n = re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        n = re.sub(r"\(.*$", "", n)[:90]
        agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"launches per step {sum(v[0] for v in agg.values()) / len(use):.0f}, kernel time per step {sum(v[1] for v in agg.values()) / len(use) / 1e3:.2f} ms, "
      f"step span {sum(int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp']) for s in use) / len(use) / 1e6:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{c / len(use):7.1f} x {t / c:8.1f} us = {t / len(use) / 1e3:7.3f} ms  {n}")
