#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5p_c4 3 -- "base=X=1" "loop=CCV_MPPI_LIB=$R/_abl/lib_sololoop.so"
