#!/bin/bash
# a second, longer pass of the stress tools with other seeds (GPU box)
cd $GRAFT_REPO_ROOT
python tools/stress_parity.py 3000 4242 > gpurun_out/r05_stress_parity_seed4242.txt 2>&1; tail -1 gpurun_out/r05_stress_parity_seed4242.txt
python tools/stress_batch.py 500 4243 > gpurun_out/r05_stress_batch_seed4243.txt 2>&1; tail -1 gpurun_out/r05_stress_batch_seed4243.txt
