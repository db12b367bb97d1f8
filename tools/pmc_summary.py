"""Averages rocprofv3 --pmc counter_collection.csv files per launch of the traversal kernel (development aid).
usage: python tools/pmc_summary.py <dir-or-csv> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

tot = defaultdict(lambda: [0.0, 0])
dur = [0.0, 0]
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        seen = set()
        for row in csv.DictReader(open(f)):
            if "k_trace<0" not in row["Kernel_Name"]:
                continue
            c = tot[row["Counter_Name"]]
            c[0] += float(row["Counter_Value"])
            c[1] += 1
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                dur[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
                dur[1] += 1
for k in sorted(tot):
    print("%-44s %16.1f  (avg of %d launches)" % (k, tot[k][0] / tot[k][1], tot[k][1]))
if dur[1]:
    print("kernel duration under the profiler: %.3f ms avg" % (dur[0] / dur[1]))
