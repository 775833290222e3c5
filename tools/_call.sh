cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02o/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r02o/pytest.log
[ $rc = 0 ] || exit $rc
L=$GRAFT_REPO_ROOT/_abl/lib_head.so
bash tools/ab_bench.sh r02o 2 -- "before_zstore=CCV_MPPI_LIB=$L" "now=X=1"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline" bash tools/ab_bench.sh r02o_c3 3 -- "before_zstore=CCV_MPPI_LIB=$L" "now=X=1"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02o_c4 2 -- "before_zstore=CCV_MPPI_LIB=$L" "now=X=1"
