#!/bin/bash
# every stress tool on the final build (GPU box); logs under gpurun_out/r05_stress_*.txt
cd $GRAFT_REPO_ROOT
python tools/stress_parity.py ${1:-1500} 777 > gpurun_out/r05_stress_parity.txt 2>&1 && tail -1 gpurun_out/r05_stress_parity.txt &&
python tools/stress_batch.py ${2:-300} 778 > gpurun_out/r05_stress_batch.txt 2>&1 && tail -1 gpurun_out/r05_stress_batch.txt &&
python tools/stress_stereo.py 200 > gpurun_out/r05_stress_stereo.txt 2>&1 && tail -1 gpurun_out/r05_stress_stereo.txt &&
python tools/stress_matchers.py 20 > gpurun_out/r05_stress_matchers.txt 2>&1 && tail -1 gpurun_out/r05_stress_matchers.txt
