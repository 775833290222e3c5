cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r02j/pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r02j/pytest.log
[ $rc = 0 ] || exit $rc
L=$GRAFT_REPO_ROOT/_abl/lib_head.so
for i in 1 2; do
for v in head new; do
  if [ $v = head ]; then export CCV_MPPI_LIB=$L; else unset CCV_MPPI_LIB; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v default-line closed_loop us_per_tick %.2f  rms %.4f' % (d['closed_loop']['us_per_tick'], d['closed_loop']['path_error_rms_m']))"
  python bench.py --closed-loop --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v --closed-loop us/tick %.2f' % (1e3*d['ms_per_step']))"
done; done
