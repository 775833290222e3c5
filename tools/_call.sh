cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02b/pytest3.log 2>&1; rc=$?
tail -5 gpurun_out/r02b/pytest3.log
[ $rc = 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r02b/bench_default.json 2> gpurun_out/r02b/bench_default.err; echo "bench rc $?"; tail -3 gpurun_out/r02b/bench_default.err
CCV_BENCH_DEVICE=0 CCV_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/r02b/bench_2rank_rehearsal.json 2> gpurun_out/r02b/bench_2rank.err; echo "2-rank rc $?"; tail -3 gpurun_out/r02b/bench_2rank.err
