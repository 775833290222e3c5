cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/_abl/lib_head.so
bash tools/ab_bench.sh r02i 3 -- "head=CCV_MPPI_LIB=$L" "pts2=X=1"
BENCH_ARGS="--workload C4 --steps 100 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02i_c4 3 -- "head=CCV_MPPI_LIB=$L" "pts2=X=1"
BENCH_ARGS="--samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline" bash tools/ab_bench.sh r02i_k512k 3 -- "head=CCV_MPPI_LIB=$L" "pts2=X=1"
BENCH_ARGS="--workload C3 --steps 200 --warmup 20 --no-cpu-baseline" bash tools/ab_bench.sh r02i_c3 2 -- "head=CCV_MPPI_LIB=$L" "pts2=X=1"
