set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1 || { tail -30 gpurun_out/r02a/pytest.log; exit 1; }
tail -3 gpurun_out/r02a/pytest.log
bash tools/ab_bench.sh r02a 3 -- "r01=CCV_MPPI_LIB=$GRAFT_REPO_ROOT/_abl/lib_r01.so" "new=X=1" "new_noprune=CCV_MPPI_PRUNE=0"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline" bash tools/ab_bench.sh r02a_c3 2 -- "r01=CCV_MPPI_LIB=$GRAFT_REPO_ROOT/_abl/lib_r01.so" "new=X=1" "new_prune=CCV_MPPI_PRUNE=1"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02a_c4 2 -- "r01=CCV_MPPI_LIB=$GRAFT_REPO_ROOT/_abl/lib_r01.so" "new=X=1"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02a_k512k 2 -- "r01=CCV_MPPI_LIB=$GRAFT_REPO_ROOT/_abl/lib_r01.so" "new=X=1"
