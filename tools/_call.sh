#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/final4
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/final4/pytest.txt 2>&1; echo pytest rc=$?
tail -3 gpurun_out/final4/pytest.txt
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final4/smoke.txt 2>&1; echo smoke rc=$?; tail -1 gpurun_out/final4/smoke.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/final4/bench_driver.json 2> gpurun_out/final4/bench.err; echo bench rc=$?
python3 -c "
import json; d=json.loads(open('gpurun_out/final4/bench_driver.json').read()); r=d['roofline']
print('%.3e rollouts/s  %.2f us/step  kernel %.2f us frac %.3f traffic %s closed loop %.1f' % (d['value'], 1e3*d['ms_per_step'], r['kernel_avg_us'], r['frac'], r['traffic'], d['closed_loop']['us_per_tick']))"
timeout -k 10 600 python3 tests/soak_determinism.py 3000 > gpurun_out/final4/soak.txt 2>&1; tail -4 gpurun_out/final4/soak.txt
