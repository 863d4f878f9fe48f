#!/bin/bash
# second half of profile_round.sh alone (kernel stats + PMC passes), for when the bench line is already there
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
( while sleep 45; do date >> $out/heartbeat.log; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_bench
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-workloads --no-verify --gen-workers 1 > $out/prof_bench.log 2>&1
cp $(ls $out/prof_bench/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $out/prof_bench 3 > $out/step_timeline.txt
rm -f $out/prof_bench/*/*kernel_trace.csv
PMC_NO_HEARTBEAT=1 bash $GRAFT_REPO_ROOT/tools/pmc_traffic.sh > /dev/null
bash $GRAFT_REPO_ROOT/tools/pmc_run.sh a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
bash $GRAFT_REPO_ROOT/tools/pmc_run.sh b SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_a $out/pmc_b > $out/sq_counters.txt
echo done
