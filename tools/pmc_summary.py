#!/usr/bin/env python3
"""tools/pmc_summary.py DIR [kernel-substring] -- per-kernel mean of every counter found in rocprofv3
*counter_collection.csv files under DIR (one row per kernel/counter, mean over dispatches)."""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]

def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if pat and pat not in k: continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-42s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))

if __name__ == "__main__":
    main()
