#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4j
timeout -k 10 300 tools/microbench/sqrt_check > gpurun_out/r4j/sqrt_check.txt 2>&1; rc=$?
cat gpurun_out/r4j/sqrt_check.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4j/pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/r4j/pytest.txt
[ $rc -eq 0 ] || exit 1
export BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg"
timeout -k 10 300 bash tools/ab_bench.sh r4j 3 -- "new=X=1"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4j_c3 2 -- "new=X=1"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4j_c4 2 -- "new=X=1"
