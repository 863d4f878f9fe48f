#!/bin/bash
# usage: tools/prof_run.sh <tag> [quick_gpu args]   (on the GPU box): rocprofv3 kernel-trace stats
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/tools/quick_gpu.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith(('k_', 'void k_')):
            print("%-18s calls=%4s avg=%9.1f us min=%8.1f max=%8.1f total=%8.2f ms %5s%%" % (r['Name'].split('(')[0][:18], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
