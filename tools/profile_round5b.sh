#!/bin/bash
# refresh of the round-5 files that depend on the latency path's final form (bench line incl. tracking_front_end, tracking timeline + probe)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/r05_bench.json 2> $out/r05_bench.err
python3 bench.py --steps 20 --warmup 5 > $out/r05_bench_20_steps.json 2> $out/r05_bench_20_steps.err
python3 bench.py --force-gather --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end > $out/r05_bench_force_gather_rccl.json 2> $out/r05_bench_force_gather.err
bash tools/tracking_timeline.sh view > $out/r05_tracking_timeline.log 2>&1
cp $out/tracking_timeline_view.txt $out/r05_tracking_timeline.txt
python3 tools/tracking_loop_probe.py 40 2>&1 | grep -v amdgpu > $out/r05_tracking_probe.txt
echo done
