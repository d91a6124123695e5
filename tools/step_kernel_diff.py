"""Per-kernel device time of the LAST replayed step of two rocprofv3 kernel traces, side by side (ms, launches), sorted by the difference.
usage: step_kernel_diff.py <trace dir A> <trace dir B>"""
import csv, glob, re, sys, collections


def last_step(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    opt = [i for i, r in enumerate(rows) if "opt_adamw_kernel" in r[2]]
    t, n = collections.Counter(), collections.Counter()
    for s, e, k in rows[opt[-2] + 1:opt[-1] + 1]:
        k = re.sub(r"\(anonymous namespace\)::", "", k)
        k = re.sub(r"^void ", "", k).split("(")[0]
        t[k] += (e - s) / 1e6
        n[k] += 1
    return t, n


ta, na = last_step(sys.argv[1])
tb, nb = last_step(sys.argv[2])
print(f"total kernel ms: A {sum(ta.values()):.3f} ({sum(na.values())} launches)   B {sum(tb.values()):.3f} ({sum(nb.values())} launches)")
for k in sorted(set(ta) | set(tb), key=lambda k: -abs(tb[k] - ta[k]))[:40]:
    print(f"{tb[k] - ta[k]:+8.3f} ms   A {ta[k]:7.3f} x{na[k]:<4d} B {tb[k]:7.3f} x{nb[k]:<4d} {k[:90]}")
