cd $GRAFT_REPO_ROOT
C="--workload mono_1920x1080_4000feat --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end --no-verify"
for opt in "" "--handle-options 26=1" "--handle-options 26=1,11=1" "--handle-options 11=1"; do for B in 32; do
for rep in 1 2; do
python bench.py $C --batch $B $opt > gpurun_out/r05_b1080.json 2>gpurun_out/r05_b1080.err; python -c "
import json; d=json.load(open('gpurun_out/r05_b1080.json')); print('B=$B opt=[$opt]', d['value'], d['ms_per_step'], d['stage_ms_per_call']['quadtree'], d['stage_ms_per_call']['fast'])"
done; done; done
