"""Per-(kernel, grid) launch statistics of a rocprofv3 --kernel-trace CSV: sum_kernel_trace.py <dir> <substring of the kernel name>."""
import csv, sys, collections, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if pat in n:
        m = re.search(r"(\w+_kernel)(<[^(]*>)?", n)
        key = ((m.group(1) + (m.group(2) or "")) if m else n[:40], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k}: n={len(v)} median {v2[len(v2)//2]:.1f} us  sum {sum(v)/1e3:.2f} ms")
    tot += sum(v)
print("total ms", tot / 1e3)
