#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call (run on the GPU box):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic (two PMC passes), SQ counters (two passes)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
( while sleep 45; do date >> $out/heartbeat.log; done ) &   # PMC passes are slow and silent; gpurun kills silent runs
hb=$!
trap "kill $hb 2>/dev/null" EXIT
cd $GRAFT_REPO_ROOT
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-other-workloads --no-verify > $out/prof_bench.log 2>&1
cp $(ls $out/prof_bench/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
PMC_NO_HEARTBEAT=1 bash $GRAFT_REPO_ROOT/tools/pmc_traffic.sh > /dev/null
bash $GRAFT_REPO_ROOT/tools/pmc_run.sh a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
bash $GRAFT_REPO_ROOT/tools/pmc_run.sh b SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_a $out/pmc_b > $out/sq_counters.txt
tail -n 1 $out/bench.json
