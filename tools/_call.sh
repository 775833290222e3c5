#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4v 3 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4v_c3 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4v_c4 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r4v_k512 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
