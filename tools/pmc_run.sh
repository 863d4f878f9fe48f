#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...>   (run on the GPU box via gpurun; one PMC pass)
# writes gpurun_out/pmc_<tag>/ ; per-kernel averages are printed by tools/pmc_summary.py
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/quick_gpu.py 128 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
