#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5c
for i in 1 2 3 4 5; do
  for v in 1 0; do
    CCV_BENCH_POLL=$v timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-closed-loop-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('poll=$v  %.2f us/step  kernel %.2f' % (1e3*d['ms_per_step'], d['roofline']['kernel_avg_us']))"
  done
done | tee gpurun_out/r5c/poll.txt
