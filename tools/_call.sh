#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/final5
export HSA_ENABLE_IPC_MODE_LEGACY=0
CCV_BENCH_DEVICE=0 CCV_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/final5/x2.json 2> gpurun_out/final5/x2.err; echo rc=$?
python3 -c "
import json; d=json.loads(open('gpurun_out/final5/x2.json').read()); print(d['value'], d['ms_per_step'], d['config']['exchange'])"
timeout -k 10 500 python3 tests/closed_loop_eval.py > gpurun_out/final5/closed_loop_eval.txt 2>&1; echo rc=$?; tail -12 gpurun_out/final5/closed_loop_eval.txt
