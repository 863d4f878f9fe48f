#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs: python tools/pmc_summary.py <dir>..."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(lambda: collections.defaultdict(int))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            k = k[5:] if k.startswith("void ") else k
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
        for k in sorted(acc):
            if not k.startswith("k_"):
                continue
            print(k, " ".join("%s=%.4g" % (c, acc[k][c] / n[k][c]) for c in sorted(acc[k])), "(n=%d)" % max(n[k].values()))
