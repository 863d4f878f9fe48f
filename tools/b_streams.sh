cd $GRAFT_REPO_ROOT
C="--steps 60 --warmup 10 --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end"
for wl in "kitti_stereo_1241x376_1000feat --batch 64" "mono_1920x1080_4000feat --batch 32" "kitti_stereo_1241x376_2000feat --batch 64"; do
for S in 1 2 3; do
python bench.py $C --workload $wl --streams $S > gpurun_out/r05_bs.json 2>gpurun_out/r05_bs.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bs.json')); print('$wl S=$S', d['value'], d['ms_per_step'], d['verified'], d['roofline']['kernel_ms'])"
done; done
