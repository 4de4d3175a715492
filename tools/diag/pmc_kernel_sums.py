#!/usr/bin/env python
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name and counter: pmc_kernel_sums.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").split("(")[0][:60]
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row.get("Dispatch_Id"))
    print("==", d)
    for k in sorted(sums):
        n = max(1, len(launches[k]))
        print("%-60s launches=%d" % (k, n))
        for c in sorted(sums[k]):
            print("    %-28s %.6g per launch" % (c, sums[k][c] / n))
