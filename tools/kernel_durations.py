"""Distribution of one kernel's durations in a rocprofv3 --kernel-trace CSV: python tools/kernel_durations.py <dir> <kernel name prefix>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sys.argv[2] in r["Kernel_Name"]])
print("n=%d mean=%.1f min=%.1f p10=%.1f p50=%.1f p90=%.1f max=%.1f" % (len(d), d.mean(), d.min(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max()))
k = 25
for i in range(0, len(d), k):
    print("launches %3d..%3d: mean %.1f  max %.1f" % (i, min(i + k, len(d)) - 1, d[i:i + k].mean(), d[i:i + k].max()))
