cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/_abl
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02c_stag 2 -- "new=X=1" "stag2=CCV_MPPI_LIB=$L/lib_stag2.so" "stag4=CCV_MPPI_LIB=$L/lib_stag4.so" "stag8=CCV_MPPI_LIB=$L/lib_stag8.so"
BENCH_ARGS="--samples-per-gpu 131072 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02c_stag_dd 2 -- "new=X=1" "stag2=CCV_MPPI_LIB=$L/lib_stag2.so" "stag4=CCV_MPPI_LIB=$L/lib_stag4.so"
