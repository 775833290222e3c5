#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=("base=X=1")
for n in maxilp bias0 trackers nopostsched; do V+=("$n=CCV_MPPI_LIB=$R/_abl/lib_$n.so"); done
BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 400 bash tools/ab_bench.sh r5g 2 -- "${V[@]}"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 400 bash tools/ab_bench.sh r5g_c4 2 -- "${V[@]}"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 400 bash tools/ab_bench.sh r5g_c3 1 -- "${V[@]}"
