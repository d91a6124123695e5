"""Timeline of the end-of-backward tail of one replayed step from a rocprofv3 kernel trace: start / end (us, relative to the first listed kernel)
of the grouped weight-gradient launches and their neighbours.  usage: tail_timeline.py <trace dir> [pattern]"""
import csv, glob, re, sys
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"wgrad|reduce_rows_batched|opt_|factored")
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last step: everything after the second-to-last opt_adamw kernel
opt = [i for i, r in enumerate(rows) if r[2].startswith("opt_adamw_kernel") or "opt_adamw_kernel" in r[2]]
lo = opt[-2] + 1 if len(opt) >= 2 else 0
sel = [r for r in rows[lo:] if pat.search(r[2])]
t0 = sel[0][0]
for s, e, n in sel:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  {n[:70]}")
