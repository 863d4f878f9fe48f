#!/bin/bash
# kernel + memory-copy timeline of the host-streaming loop (on the GPU box): tools/e2e_timeline.sh
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_e2e
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/prof_e2e -- python3 $GRAFT_REPO_ROOT/tools/e2e_probe.py 24 > $out/e2e_probe.log 2>&1
python3 - <<PY
import csv, glob
k = glob.glob("$out/prof_e2e/*/*kernel_trace.csv")[0]
m = glob.glob("$out/prof_e2e/*/*memory_copy_trace.csv")[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], "q" + r.get("Queue_Id", "?")))
for r in csv.DictReader(open(m)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Kind", "?"))[:14], ""))
ev.sort()
# the last ~3 steps before the end
pads = [i for i, e in enumerate(ev) if "k_pyr_pad" in e[2]]
i0 = pads[-5]
t0 = ev[i0][0]
with open("$out/e2e_timeline.txt", "w") as f:
    for s, e, n, q in ev[i0:pads[-2] + 1]:
        f.write("%-24s %-4s start %9.1f end %9.1f dur %8.1f\n" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
rm -f $out/prof_e2e/*/*kernel_trace.csv
cat $out/e2e_probe.log | tail -2
