"""Summary of tools/gpu_pmc_ab.sh: per library and config, every counter of the trace kernel averaged per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
table = defaultdict(dict)
for path in sorted(glob.glob(os.path.join(root, "*", "*", "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    rel = os.path.relpath(path, root).split(os.sep)
    lib, cfg = rel[0], rel[1]
    acc, disp = defaultdict(float), defaultdict(set)
    for row in csv.DictReader(open(path)):
        if "k_trace" not in row["Kernel_Name"] or "ILi0E" in row["Kernel_Name"].split("k_trace")[1][:8]:
            continue  # (the AO / ray kernels only: the primary pass that feeds them is not what is compared)
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        disp[row["Counter_Name"]].add(row["Dispatch_Id"])
    for k, v in acc.items():
        table[(cfg, k)][lib] = v / max(len(disp[k]), 1)
libs = sorted({l for v in table.values() for l in v})
print("%-14s %-30s " % ("config", "counter") + " ".join("%14s" % l for l in libs) + "   ratio(last/first)")
for (cfg, k), v in sorted(table.items()):
    vals = [v.get(l, float("nan")) for l in libs]
    print("%-14s %-30s " % (cfg, k) + " ".join("%14.0f" % x for x in vals) + "   %.3f" % (vals[-1] / vals[0] if vals[0] else float("nan")))
