cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02g/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r02g/pytest.log
[ $rc = 0 ] || exit $rc
L=$GRAFT_REPO_ROOT/_abl/lib_head.so
bash tools/ab_bench.sh r02g 3 -- "head=CCV_MPPI_LIB=$L" "flags=X=1"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline" bash tools/ab_bench.sh r02g_c3 2 -- "head=CCV_MPPI_LIB=$L" "flags=X=1"
