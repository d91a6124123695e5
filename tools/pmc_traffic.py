"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: they do not fit one TCC pass).

Corrections applied (guide, section HBM): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, so it is DOUBLED; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json] [--steps-profiled N]
--steps-profiled: eager steps each --pmc pass ran (warm-up included): every kernel entry then carries launches_per_step, and the
file a "stamp" (tools/build_stamp.py: library sha256 + source commit) -- bench.py prints the stamp in roofline.traffic_source and
reports traffic only when the library it runs and the launch count it times are the ones profiled here.
Writes {"kernels": {"<kernel name as rocprofv3 prints it, without 'void (anonymous namespace)::' and the argument list>":
        {"launches": n, "fetch_kib": mean, "write_kib": mean, "hbm_bytes_per_launch": 2*fetch*1024 + write*1024}}}
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def main():
    steps = None
    if "--steps-profiled" in sys.argv:
        i = sys.argv.index("--steps-profiled")
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                    "(gfx950 FETCH_SIZE counts 128-B requests as 64 B)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        out["kernels"][k] = {"launches": max(len(f), len(w)), "fetch_kib": round(fm, 1), "write_kib": round(wm, 1),
                             "hbm_bytes_per_launch": round((2 * fm + wm) * 1024)}
        if steps:
            out["kernels"][k]["launches_per_step"] = max(len(f), len(w)) / steps
    if steps:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import build_stamp
        out["stamp"] = dict(build_stamp.current(), steps_profiled=steps,
                            command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-graph ...")
    path = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    json.dump(out, open(path, "w"), indent=1)
    top = sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
    for k, v in top:
        print(f"{v['launches']:5d} x {v['hbm_bytes_per_launch']/1e6:9.2f} MB  {k[:90]}")


main()
