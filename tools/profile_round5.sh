#!/bin/bash
# Round-5 evidence (on the GPU box): headline bench line, kernel stats + HBM traffic + SQ counters + step timeline of the headline and of
# BASELINE config 4's per-GPU share, the one-rank RCCL leg, the single-frame tracking timeline.  Files land in gpurun_out/ as r05_*;
# copy the ones to keep into profiles/.  Part 1 and part 2 are separate gpurun calls (each well inside a call's time limit).
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
if [ "$1" = "1" ]; then
python3 bench.py > $out/r05_bench.json 2> $out/r05_bench.err
python3 bench.py --steps 20 --warmup 5 > $out/r05_bench_20_steps.json 2> $out/r05_bench_20_steps.err
python3 bench.py --force-gather --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end > $out/r05_bench_force_gather_rccl.json 2> $out/r05_bench_force_gather.err
PMC=1 SQ=1 PMC_WORKLOAD=kitti_stereo_1241x376_1000feat PMC_BATCH=64 PMC_IMAGES=128 bash tools/profile_workload.sh r05_headline --workload kitti_stereo_1241x376_1000feat
else
PMC=1 SQ=1 PMC_WORKLOAD=mono_1920x1080_4000feat PMC_BATCH=64 PMC_IMAGES=64 bash tools/profile_workload.sh r05_fullhd --workload mono_1920x1080_4000feat --batch 64
bash tools/tracking_timeline.sh view > $out/r05_tracking_timeline.log 2>&1
cp $out/tracking_timeline_view.txt $out/r05_tracking_timeline.txt
python3 tools/tracking_loop_probe.py 40 > $out/r05_tracking_probe.txt 2>&1
ORBX_LIB=$GRAFT_REPO_ROOT/orb_slam2v2-1_amd/lib/liborbx_hip_dev.so python3 tools/octree_pass_time_probe.py fullhd4000 > $out/r05_fullhd_octree_pass_times.txt 2>&1
fi
echo done
