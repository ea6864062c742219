#!/usr/bin/env python3
"""Condenses what tools/profile_r02.sh left under gpurun_out/ into the summaries kept under profiles/:

    python tools/summarize_r02.py r02

  profiles/<tag>_<cfg>_kernel_stats.csv   the kernel-trace --stats rows of the cvec kernels of one configuration
  profiles/<tag>_kernel_stats.csv         the same for bench.py
  profiles/<tag>_traffic.json             per configuration and kernel: HBM bytes per launch from the counters
                                          (FETCH_SIZE x 2 + WRITE_SIZE, both reported in KiB; on gfx950 FETCH_SIZE
                                          tallies a 128-byte read request at 64 bytes), the average launch time
                                          and, for the dominant kernel of each operation, the algorithmic bytes
  profiles/roofline_traffic.json          the headline kernel's traffic, read by bench.py"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5sin", "cfg5mix", "bench"]
MODES = {"0": "filter (one-of / float zero)", "1": "decode", "2": "filter + masks", "3": "filter (bit pattern)",
         "4": "filter + masks (bit pattern)"}


def short(name):
    """cvec::k_sections<18, 3, 1, 18> -> readable"""
    m = re.match(r"(?:void )?cvec::(k_\w+)<([^>]*)>", name)
    if not m:
        m2 = re.match(r"(?:void )?cvec::(k_\w+)", name)
        return m2.group(1) if m2 else name[:60]
    k, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    et = {"16": "u64", "17": "u32", "18": "f32"}.get(args[0].replace("(int)", ""), args[0])
    if k in ("k_sections", "k_vstream") and len(args) >= 3:
        mode = MODES.get(args[1].replace("(int)", ""), args[1])
        shape = ", %s KiB%s" % (args[3].replace("(int)", ""), ", reference tail rule" if len(args) > 4 and args[4].replace("(int)", "") == "1" else "") \
            if len(args) > 3 and k == "k_sections" else ""
        if k == "k_sections" and len(args) > 5:
            shape += {"1": ", wide stagings", "2": ", paired stores"}.get(args[5].replace("(int)", ""), "")
        return "%s<%s, %s, %s values%s>" % (k, et, mode, args[2].replace("(int)", ""), shape)
    return "%s<%s>" % (k, ", ".join([et] + args[1:]))


def counter_by_kernel(tag_dir, counter):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", tag_dir, "**", "*counter_collection.csv"), recursive=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in files:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "cvec::" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    tag = sys.argv[1]
    traffic = {}
    for cfg in CFGS:
        stats = glob.glob(os.path.join(ROOT, "gpurun_out", "%s_%s_kt" % (tag, cfg), "**", "*kernel_stats.csv"), recursive=True)
        if not stats:
            continue
        rows = [r for r in csv.DictReader(open(stats[0])) if "cvec::" in r["Name"]]
        out_csv = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg) if cfg != "bench" else "%s_kernel_stats.csv" % tag)
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        avg = {}
        for r in rows:
            avg[short(r["Name"])] = float(r["AverageNs"])
        fetch = counter_by_kernel("%s_%s_fetch" % (tag, cfg), "FETCH_SIZE")
        write = counter_by_kernel("%s_%s_write" % (tag, cfg), "WRITE_SIZE")
        info = {}
        log = os.path.join(ROOT, "gpurun_out", "%s_%s_kt.log" % (tag, cfg))
        if os.path.exists(log):
            for line in open(log):
                if line.startswith("{") and '"algorithmic_bytes"' in line:
                    info = json.loads(line)
        ent = {"uncompressed_bytes": info.get("uncompressed_bytes"), "compressed_bytes": info.get("compressed_bytes"),
               "algorithmic_bytes_per_operation": info.get("algorithmic_bytes"), "kernels": {}}
        for k in sorted(set(avg) | set(fetch) | set(write)):
            f_kib, w_kib = fetch.get(k), write.get(k)
            e = {"avg_launch_us": round(avg[k] / 1e3, 2) if k in avg else None}
            if f_kib is not None and w_kib is not None:
                e["FETCH_SIZE_KiB"] = round(f_kib, 1)
                e["WRITE_SIZE_KiB"] = round(w_kib, 1)
                e["hbm_bytes_per_launch"] = int(2 * f_kib * 1024 + w_kib * 1024)
                if k in avg and avg[k] > 0:
                    e["hbm_GBs"] = round(e["hbm_bytes_per_launch"] / avg[k], 1)
            ent["kernels"][k] = e
        traffic[cfg] = ent
        print(cfg, "->", out_csv)
    traffic["_correction"] = "hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): gfx950 tallies a 128-byte read request at 64 bytes"
    json.dump(traffic, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
    b = traffic.get("bench", {}).get("kernels", {})
    for k, e in b.items():
        if k.startswith("k_sections<f32") and "hbm_bytes_per_launch" in e:
            json.dump({"kernel": "cvec::" + k, "vectors_per_gpu": 131072, "FETCH_SIZE_KiB_per_launch": e["FETCH_SIZE_KiB"],
                       "WRITE_SIZE_KiB_per_launch": e["WRITE_SIZE_KiB"],
                       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE x1",
                       "hbm_bytes_per_launch": e["hbm_bytes_per_launch"]},
                      open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=1)
            print("roofline_traffic.json <-", k, e)


if __name__ == "__main__":
    main()
