#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for K in 81920 98304 131072 196608; do
  echo "K=$K"
  BENCH_ARGS="--samples-per-gpu $K --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5h_$K 2 -- "default=X=1" "r4=CCV_MPPI_KERNEL=r4"
done
echo C3
BENCH_ARGS="--workload C3 --samples-per-gpu 131072 --steps 100 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5h_c3 2 -- "default=X=1" "r4=CCV_MPPI_KERNEL=r4"
