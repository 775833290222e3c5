cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02n/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r02n/pytest.log
[ $rc = 0 ] || exit $rc
L=$GRAFT_REPO_ROOT/_abl/lib_head.so
bash tools/ab_bench.sh r02n 3 -- "head=CCV_MPPI_LIB=$L" "zstore=X=1"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline" bash tools/ab_bench.sh r02n_c3 2 -- "head=CCV_MPPI_LIB=$L" "zstore=X=1"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02n_c4 2 -- "head=CCV_MPPI_LIB=$L" "zstore=X=1"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02n_k512k 2 -- "head=CCV_MPPI_LIB=$L" "zstore=X=1"
