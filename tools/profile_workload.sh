#!/bin/bash
# usage (on the GPU box via gpurun): tools/profile_workload.sh <tag> <bench.py args...>
# kernel stats + one step's timeline of `bench.py <args>` (a non-headline workload), into gpurun_out/<tag>_*.
# With PMC=1 also the two HBM-traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only).
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
( while sleep 45; do date >> $out/heartbeat.log; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --no-other-workloads --no-verify --no-end-to-end --no-tracking --gen-workers 1"
rm -rf $out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $common --steps 40 --warmup 10 --ramp-steps 60 "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cp $(ls $out/prof_$tag/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $out/prof_$tag 15 > $out/${tag}_step_timeline.txt
rm -f $out/prof_$tag/*/*kernel_trace.csv
if [ -n "$PMC" ]; then
    rm -rf $out/pmc_fetch_$tag $out/pmc_write_$tag
    timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $common --steps 4 --warmup 1 --ramp-steps 0 "$@" > $out/pmc_fetch_$tag.log 2>&1
    timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $common --steps 4 --warmup 1 --ramp-steps 0 "$@" > $out/pmc_write_$tag.log 2>&1
    python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/pmc_fetch_$tag $out/pmc_write_$tag > $out/${tag}_pmc_traffic.json
    rm -rf $out/pmc_fetch_$tag $out/pmc_write_$tag
fi
if [ -n "$SQ" ]; then   # SQ instruction / wait counters of the same command, two passes (the counters do not fit one)
    for pass in a b; do
        if [ $pass = a ]; then ctr="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS";
        else ctr="SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; fi
        rm -rf $out/pmc_sq${pass}_$tag
        timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_sq${pass}_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $common --steps 3 --warmup 1 --ramp-steps 0 "$@" > $out/pmc_sq${pass}_$tag.log 2>&1
    done
    python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_sqa_$tag $out/pmc_sqb_$tag > $out/${tag}_sq_counters.txt
    rm -rf $out/pmc_sqa_$tag $out/pmc_sqb_$tag
fi
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/${tag}_kernel_stats.csv")):
    if r['Name'].startswith(('k_', 'void k_')):
        print("%-22s calls=%4s avg=%9.1f us max=%8.1f total=%8.2f ms %5s%%" % (r['Name'].split('(')[0].replace('void ','')[:22], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
