# A/B of trees under _ab_* (worktrees built beforehand) against the current one on the SAME box
cd $GRAFT_REPO_ROOT
C="--steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end --no-verify"
TREES="${TREES:-_ab_r04 .}"
for rep in 1 2; do for W in ${WL:-"mono_1920x1080_4000feat:64"}; do w=${W%%:*}; b=${W##*:}; for tree in $TREES; do
( cd $tree && python bench.py $C --workload $w --batch $b > $GRAFT_REPO_ROOT/gpurun_out/ab.json 2>$GRAFT_REPO_ROOT/gpurun_out/ab.err ) && python -c "
import json; d=json.load(open('gpurun_out/ab.json')); s=d['stage_ms_per_call']; print('tree=$tree $w B=$b', d['value'], d['ms_per_step'], {k: s[k] for k in ('pyramid','fast','quadtree','describe')})" 
done; done; done
