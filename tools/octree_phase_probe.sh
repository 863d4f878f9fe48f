#!/bin/bash
# k_octree_pyr phase times on a single image: tools/octree_phase_probe.sh W H NFEAT   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for stop in 1 2 3 4 0; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/octp_$stop
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/octp_$stop -- python3 $GRAFT_REPO_ROOT/tools/octree_phase_probe.py $1 $2 $3 $stop > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/octp_$stop/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("k_octree_pyr"):
            print("stop $stop: k_octree_pyr min %.1f us  median-ish avg %.1f us  (%s calls)" % (float(r["MinNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
