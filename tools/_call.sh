#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5b
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -x -q -m gpu > gpurun_out/r5b/pytest.txt 2>&1; rc=$?
tail -4 gpurun_out/r5b/pytest.txt
[ $rc -eq 0 ] || exit 1
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5b_c4 3 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5b_k512 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
