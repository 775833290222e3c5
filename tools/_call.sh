#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/final3
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "state_store or read_back" > gpurun_out/final3/pytest.txt 2>&1; echo rc=$?; tail -4 gpurun_out/final3/pytest.txt
