"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace CSV: python tools/step_timeline.py <dir> [step_from_end] [nsteps]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pyr_pad" in r["Kernel_Name"]]
i0, i1 = idx[-back], idx[-back + nst]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-20s q%-3s start %8.1f end %8.1f dur %7.1f" % (r["Kernel_Name"].split("(")[0].replace("void ", "")[:20], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
