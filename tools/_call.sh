#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4r
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4r/pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/r4r/pytest.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4r/smoke.txt 2>&1; tail -5 gpurun_out/r4r/smoke.txt
