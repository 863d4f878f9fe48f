#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration (tools/pmc_calibrate.hip) -> gpurun_out/pmc_calibration.json.  Run on the GPU box.
set -e
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/cal_fetch $out/cal_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/cal_fetch -- $GRAFT_REPO_ROOT/tools/_build/pmc_calibrate > $out/cal_expected.json 2> $out/cal_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/cal_write -- $GRAFT_REPO_ROOT/tools/_build/pmc_calibrate > /dev/null 2> $out/cal_write.log
python3 $GRAFT_REPO_ROOT/tools/pmc_calibrate.py $out/cal_fetch $out/cal_write $out/cal_expected.json > $out/pmc_calibration.json
cat $out/pmc_calibration.json
