#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4g
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4g/pytest.txt 2>&1; rc=$?
tail -5 gpurun_out/r4g/pytest.txt
[ $rc -eq 0 ] && timeout -k 10 300 python3 bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err && cat gpurun_out/r4g/bench.json
