#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5o 3 -- "loop=X=1" "unrolled=CCV_MPPI_LIB=$R/_abl/lib_noloop.so"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5o_c3 3 -- "loop=X=1" "unrolled=CCV_MPPI_LIB=$R/_abl/lib_noloop.so"
