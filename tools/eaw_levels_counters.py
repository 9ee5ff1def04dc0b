#!/usr/bin/env python3
"""tools/eaw_levels_counters.py DIR -- the five a-trous levels of LeveledEAWFilter ONE BY ONE from the rocprofv3 passes of
tools/profile_eaw_levels.sh: duration, FETCH_SIZE, WRITE_SIZE and HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE: the gfx950 correction of
MI355X_MICROARCH.md) per template instantiation k_wavelet_tiled<STEP, ...> (a level of step s is the instantiation STEP = s), against the
algorithmic 44 B per pixel and level."""
import csv, glob, os, re, sys
from collections import defaultdict

W, H = 1920, 1080
ALGO = 44 * W * H


def step_of(name):
    m = re.search(r"k_wavelet_tiled<\s*(\d+)", name)
    if m:
        return int(m.group(1))
    m = re.search(r"k_wavelet_tiledILi(\d+)E", name)          # mangled
    return int(m.group(1)) if m else None


def main():
    root = sys.argv[1]
    dur, cnt = defaultdict(list), defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = step_of(r["Kernel_Name"])
            if s:
                dur[s].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = step_of(r["Kernel_Name"])
            if s:
                cnt[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# algorithmic bytes per level launch: %.1f MB (44 B x %d x %d)" % (ALGO / 1e6, W, H))
    print("step  launches   avg us   FETCH_SIZE KB   WRITE_SIZE KB   HBM MB (F x 2 + W)   x algorithmic   achieved TB/s (algorithmic / time)")
    tot_t = tot_b = 0.0
    for s in sorted(dur):
        us = sum(dur[s]) / len(dur[s]) / 1e3
        f = cnt[s].get("FETCH_SIZE"); w = cnt[s].get("WRITE_SIZE")
        fk = sum(f) / len(f) if f else float("nan"); wk = sum(w) / len(w) if w else float("nan")
        hbm = fk * 1024 * 2 + wk * 1024
        tot_t += us; tot_b += hbm
        print("%4d  %8d  %7.1f  %14.0f  %14.0f  %18.1f  %14.2f  %10.2f" % (s, len(dur[s]), us, fk, wk, hbm / 1e6, hbm / ALGO, ALGO / us / 1e6))
    print("all five: %.1f us, %.1f MB = %.2f x algorithmic" % (tot_t, tot_b / 1e6, tot_b / (5 * ALGO)))


if __name__ == "__main__":
    main()
