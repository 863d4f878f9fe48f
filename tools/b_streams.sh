# whole steps overlapped on S handles / S streams (bench.py --streams S) against the single-handle pipelined step, same box
cd $GRAFT_REPO_ROOT
C="--steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end"
WL=${WL:-kitti_stereo_1241x376_1000feat:64 mono_1920x1080_4000feat:64 kitti_stereo_1241x376_2000feat:64 euroc_stereo_752x480_1000feat:64 mono_640x480_1000feat:64}
for rep in 1 2; do for W in $WL; do w=${W%%:*}; b=${W##*:}; for S in 1 2 3; do
python bench.py $C --workload $w --batch $b --streams $S > gpurun_out/ab.json 2>gpurun_out/ab.err && python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$w B=$b S=$S', d['value'], d['ms_per_step'], d.get('verified'))"
done; done; done
