"""Condenses the counter CSVs that tools/pmc_probe.sh left under gpurun_out/<tag>_*: per kernel name the
average of every counter per dispatch (the 20 largest kernels by dispatch count are printed)."""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, tag + "_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0][:70]
            c = agg[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
for k, cs in sorted(agg.items(), key=lambda kv: -max(c[1] for c in kv[1].values())):
    if "cvec" not in k:
        continue
    print(k)
    for name, (tot, n) in sorted(cs.items()):
        print("   %-28s %16.0f per dispatch (%d dispatches)" % (name, tot / n, n))
