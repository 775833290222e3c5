#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5s
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5s/pytest.txt 2>&1; rc=$?
tail -3 gpurun_out/r5s/pytest.txt
[ $rc -eq 0 ] || exit 1
BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5s 3 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5s_c4 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5s_c3 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
