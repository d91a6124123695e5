"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per (kernel, grid).

usage: python tools/pmc_summary.py <dir-or-csv>... [--match SUBSTR]
"""
import csv, sys, os, re, collections, glob

def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)[:60]

def main():
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    files = []
    for a in sys.argv[1:]:
        if a.startswith("--") or a == match: continue
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            if match not in r["Kernel_Name"]: continue
            key = (short(r["Kernel_Name"]), r["Grid_Size"])
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in sorted(acc.items()):
        print(key)
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")

main()
