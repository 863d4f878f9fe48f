cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof1080
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1080 -- python3 $GRAFT_REPO_ROOT/bench.py --workload mono_1920x1080_4000feat --batch 32 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1080.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof1080/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Name'].startswith('k_'):
            print("%-18s calls=%4s avg=%9.1f us min=%8.1f max=%8.1f" % (r['Name'].split('(')[0][:18], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
