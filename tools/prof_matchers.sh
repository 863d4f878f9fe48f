#!/bin/bash
# rocprofv3 kernel stats of tools/bench_matchers.py (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_match
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_match -- python3 $GRAFT_REPO_ROOT/tools/bench_matchers.py > $GRAFT_REPO_ROOT/gpurun_out/prof_match.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_match/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name'].split('(')[0]
        print("%-34s calls=%5s avg=%8.1f us min=%8.1f max=%8.1f" % (n[:34], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
