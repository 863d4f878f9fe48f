#!/bin/bash
# Round-4 evidence (on the GPU box): headline bench line + kernel stats + HBM traffic + SQ counters + step timeline, and the same
# for BASELINE configs 2 and 4.  Files land in gpurun_out/ as r04_*; copy the ones to keep into profiles/.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/r04_bench.json 2> $out/r04_bench.err
PMC=1 SQ=1 PMC_WORKLOAD=kitti_stereo_1241x376_1000feat PMC_BATCH=64 PMC_IMAGES=128 bash tools/profile_workload.sh r04_headline --workload kitti_stereo_1241x376_1000feat
PMC=1 SQ=1 PMC_WORKLOAD=kitti_stereo_1241x376_2000feat PMC_BATCH=64 PMC_IMAGES=128 bash tools/profile_workload.sh r04_kitti2000 --workload kitti_stereo_1241x376_2000feat
PMC=1 SQ=1 PMC_WORKLOAD=mono_1920x1080_4000feat PMC_BATCH=64 PMC_IMAGES=64 bash tools/profile_workload.sh r04_fullhd --workload mono_1920x1080_4000feat --batch 64
PMC=1 SQ=1 PMC_WORKLOAD=kitti_stereo_natural_1241x376_1000feat PMC_BATCH=64 PMC_IMAGES=128 bash tools/profile_workload.sh r04_natural --workload kitti_stereo_natural_1241x376_1000feat
echo done
