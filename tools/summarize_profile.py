#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (gpurun_out/<tag>_kt, <tag>_fetch, <tag>_write) into the
small summaries kept under profiles/: the kernel-trace --stats table of the cvec kernels and the HBM
traffic of the dominant kernel.

    python tools/summarize_profile.py r01 [vectors_per_gpu]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per
128-byte request of a wide coalesced read, i.e. half the bytes actually fetched, so the read side is
doubled before it is compared with a byte count; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avg(tag, counter, kernel_substr):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    vals = collections.defaultdict(float)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(vals.values()) / len(vals) if vals else None


def main():
    tag = sys.argv[1]
    nvec = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_kt", "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(stats))]
    out_csv = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if "cvec::" in r["Name"]:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"], r["StdDev"]])
    print("wrote", out_csv)
    fetch_kib = counter_avg(tag + "_fetch", "FETCH_SIZE", "k_sections")
    write_kib = counter_avg(tag + "_write", "WRITE_SIZE", "k_sections")
    if fetch_kib is not None and write_kib is not None:
        traffic = {"kernel": "cvec::k_sections<f32, filter>", "vectors_per_gpu": nvec,
                   "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
                   "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE x1",
                   "hbm_bytes_per_launch": int(2 * fetch_kib * 1024 + write_kib * 1024)}
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
        json.dump(traffic, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
        print(traffic)


if __name__ == "__main__":
    main()
