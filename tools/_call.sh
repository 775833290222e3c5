#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5e
timeout -k 10 600 python3 tools/fuzz_kernels.py 100 11 > gpurun_out/r5e/fuzz.txt 2>&1; echo fuzz rc=$?
tail -2 gpurun_out/r5e/fuzz.txt; grep -c "nan=1" gpurun_out/r5e/fuzz.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5e/pytest.txt 2>&1; echo pytest rc=$?
tail -3 gpurun_out/r5e/pytest.txt
BENCH_ARGS="--steps 400 --warmup 20 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5e 3 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5e_k512 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline --no-closed-loop-leg" timeout -k 10 300 bash tools/ab_bench.sh r5e_c4 2 -- "new=X=1" "prev=CCV_MPPI_LIB=$R/_abl/lib_prev.so"
