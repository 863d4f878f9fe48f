#!/bin/bash
# kernel + memory-copy timeline of ONE frame of tools/tracking_loop_probe.py (device-resident chain), on the GPU box
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_track
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/prof_track -- python3 $GRAFT_REPO_ROOT/tools/tracking_loop_probe.py 16 ${1:-device} > $out/tracking_probe_prof.log 2>&1
python3 - <<PY
import csv, glob
k = glob.glob("$out/prof_track/*/*kernel_trace.csv")[0]
m = glob.glob("$out/prof_track/*/*memory_copy_trace.csv")[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:26]))
for r in csv.DictReader(open(m)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?").replace("MEMORY_COPY_", "") + " %s B" % r.get("Size", r.get("Bytes", "?"))))
ev.sort()
pads = [i for i, e in enumerate(ev) if "k_pyr_pad<true>" in e[2] or "k_pyr_pad_rows" in e[2]]
i0, i1 = pads[-3], pads[-2]
while i0 > 0 and ev[i0 - 1][2].startswith("COPY") and ev[i0][0] - ev[i0 - 1][1] < 200000: i0 -= 1
while i1 > 0 and ev[i1 - 1][2].startswith("COPY") and ev[i1][0] - ev[i1 - 1][1] < 200000: i1 -= 1
t0 = ev[i0][0]
busy = sum(e - s for s, e, n in ev[i0:i1])
with open("$out/tracking_timeline_${1:-device}.txt", "w") as f:
    f.write("# one frame of the tracking chain, backend ${1:-device} (1241x376, 2000 features / camera): kernels and copies, us\n")
    prev = t0
    for s, e, n in ev[i0:i1]:
        f.write("%-34s start %8.1f end %8.1f dur %6.1f gap %6.1f\n" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
        prev = e
    f.write("# frame period %.1f us, GPU busy %.1f us (%d kernels / copies)\n" % ((ev[i1][0] - t0) / 1e3, busy / 1e3, i1 - i0))
PY
rm -f $out/prof_track/*/*kernel_trace.csv
tail -3 $out/tracking_probe_prof.log
