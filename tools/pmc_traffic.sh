#!/bin/bash
# HBM traffic of the bench command per kernel launch (run on the GPU box through gpurun):
# two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950),
# each with --kernel-trace only, as MI355X_MICROARCH.md "HBM" prescribes.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
if [ -z "$PMC_NO_HEARTBEAT" ]; then   # PMC passes are slow and silent; gpurun kills silent runs
    ( while sleep 45; do date >> $out/heartbeat.log; done ) &
    hb=$!
    trap "kill $hb 2>/dev/null" EXIT
fi
cd /tmp && export TMPDIR=/tmp
rm -rf $out/pmc_fetch $out/pmc_write
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --ramp-steps 0 --no-cpu-baseline --no-other-workloads --no-verify --gen-workers 1 > $out/pmc_fetch.log 2>&1
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --ramp-steps 0 --no-cpu-baseline --no-other-workloads --no-verify --gen-workers 1 > $out/pmc_write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write > $out/pmc_traffic.json
cat $out/pmc_traffic.json
