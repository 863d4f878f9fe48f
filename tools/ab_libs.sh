# A/B of library variants (ORBX_LIB) on the SAME box.  LIBS="path ..." ('-' = the tree's own library), WL="workload:batch ...", OPTS="k=v,k=v ..." ('-' = none)
cd $GRAFT_REPO_ROOT
C="--steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-tracking --no-end-to-end ${VERIFY:---no-verify}"
for rep in 1 2; do for W in ${WL:-"mono_1920x1080_4000feat:64"}; do w=${W%%:*}; b=${W##*:}; for lib in $LIBS; do for o in ${OPTS:--}; do
if [ "$lib" = "-" ]; then unset ORBX_LIB; else export ORBX_LIB=$GRAFT_REPO_ROOT/$lib; fi
if [ "$o" = "-" ]; then HO=""; else HO="--handle-options $o"; fi
python bench.py $C --workload $w --batch $b $HO > gpurun_out/ab.json 2>gpurun_out/ab.err && python -c "
import json; d=json.load(open('gpurun_out/ab.json')); s=d['stage_ms_per_call']; print('lib=$lib opts=$o $w B=$b', d['value'], d['ms_per_step'], d.get('verified'), 'fast_in_step', d['roofline']['kernel_ms'], {k: s[k] for k in ('pyramid','fast','quadtree','describe')})" 
done; done; done; done
