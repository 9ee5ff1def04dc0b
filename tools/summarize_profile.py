#!/usr/bin/env python3
"""tools/summarize_profile.py PROF_DIR OUT_PREFIX -- condense the output of tools/profile.sh:
  OUT_PREFIX_kernel_stats.csv    per-kernel calls / total / average / min / max (ns) from the kernel trace (one stream)
  OUT_PREFIX_kernel_stats_overlapped.csv   the same with frames overlapped on the auxiliary streams
  OUT_PREFIX_hbm_counters.json   per-kernel FETCH_SIZE / WRITE_SIZE (KB per launch, as reported) and HBM bytes per
                                 launch with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x 2)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name.split("(")[0][-48:]


def kernel_stats(root, sub="trace"):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return rows


def counter(root, sub, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == name:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    ks = kernel_stats(root)
    with open(out + "_kernel_stats.csv", "w") as fh:
        fh.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
        for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
            fh.write("%s,%d,%d,%.1f,%d,%d\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    ko = kernel_stats(root, "trace_overlap")
    if ko:
        with open(out + "_kernel_stats_overlapped.csv", "w") as fh:
            fh.write("# the same command with frames overlapped on the auxiliary streams (the timed region's mode): kernels share the CUs\n")
            fh.write("kernel,calls,total_ns,average_ns,min_ns,max_ns\n")
            for k, v in sorted(ko.items(), key=lambda kv: -sum(kv[1])):
                fh.write("%s,%d,%d,%.1f,%d,%d\n" % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    fetch, write = counter(root, "fetch", "FETCH_SIZE"), counter(root, "write", "WRITE_SIZE")
    res = {"units": "FETCH_SIZE / WRITE_SIZE in KB per launch as reported by rocprofv3 (separate --pmc passes); "
                    "hbm_bytes_per_launch = FETCH_SIZE*1024*2 (gfx950 reports half of wide coalesced reads) + WRITE_SIZE*1024",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
                             "average_ns": (sum(ks[k]) / len(ks[k])) if k in ks else None}
    with open(out + "_hbm_counters.json", "w") as fh:
        json.dump(res, fh, indent=1)
    print(open(out + "_kernel_stats.csv").read())
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
