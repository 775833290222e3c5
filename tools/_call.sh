cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
./tools/microbench/dist_loop > gpurun_out/r02e/dist_loop.txt 2>&1
grep -v "^$" gpurun_out/r02e/dist_loop.txt | grep "V7\|V8\|V9\|V10\|V6"
for k in pc solo; do CCV_MPPI_KERNEL=$k python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$k', d['roofline']['kernel_avg_us'], 1e3*d['ms_per_step'])"; done
