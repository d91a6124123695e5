"""Idle gaps on the device inside one replayed step of a rocprofv3 kernel trace: the union of kernel intervals of the LAST step (between the
last two opt_adamw_kernel launches), the gaps > thresh us with the kernels on either side.  usage: gap_report.py <trace dir> [thresh_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
th = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
opt = [i for i, r in enumerate(rows) if "opt_adamw_kernel" in r[2]]
lo, hi = opt[-2] + 1, opt[-1] + 1
step = rows[lo:hi]
span = (step[-1][1] - step[0][0]) / 1e3
busy, cur_e, gaps, last = 0.0, step[0][0], [], step[0]
for s, e, n in step:
    if s > cur_e:
        gaps.append(((s - cur_e) / 1e3, last[2][:60], n[:60], (s - step[0][0]) / 1e3))
        busy += 0
    if e > cur_e:
        busy += (e - max(s, cur_e)) / 1e3
        cur_e, last = e, (s, e, n)
print(f"step span {span:.1f} us, device busy {busy:.1f} us, idle {span - busy:.1f} us in {len(gaps)} gaps; {len(step)} launches")
for g, a, b, at in sorted(gaps, reverse=True)[:40]:
    if g >= th:
        print(f"{g:8.1f} us at {at:9.1f}   after {a}   before {b}")
