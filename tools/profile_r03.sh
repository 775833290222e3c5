# Round-3 measurement set (profiles/<tag>_*): bash tools/profile_r03.sh <tag>
#   bench lines: the default run (C2 + C3 / C4 + reference defaults + CPU baselines), the driver's --steps 20 --warmup 5, K = 524 288
#   on one device, the closed loop, dt beyond the small-turn gate
#   rocprofv3 --kernel-trace --stats of the C2 / C3 / C4 / closed-loop / fb-default bench commands
#   rocprofv3 --pmc passes (each counter set its own run): WRITE_SIZE, FETCH_SIZE and an SQ set for C2 / C3 / C4 / fb default
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r03a}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-other-workloads --no-defaults-leg --no-live-traffic"
python3 $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/c2_bench_driver_args.json 2>> $O/c2_bench.err
python3 $R/bench.py --samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline > $O/c2_K524288_bench.json 2>/dev/null
python3 $R/bench.py --closed-loop --no-cpu-baseline > $O/c2_closed_loop_bench.json 2>/dev/null
python3 $R/bench.py --dt 0.41 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg > $O/c2_dt0.41_bench.json 2>/dev/null
for w in C3 C4 fb_default dd_default; do python3 $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $O/${w}_bench.json 2>/dev/null; done
echo "bench lines done"
for w in C2 C3 C4 fb_default; do
  st=200; [ $w = C4 ] && st=100
  # (--no-closed-loop-leg: the resident loop launches the same kernel on another workload -- pose and window from the frame in HBM,
  #  full windows only -- and has a kernel trace of its own below; with it in, 4 % of the launches averaged here ran 6 us longer)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$w -- python3 $R/bench.py --workload $w --steps $st --warmup 20 $Q --no-closed-loop-leg > $O/${w}_bench_under_rocprof.json 2> $O/ktrace_$w.err
  find $O/ktrace_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
  echo "ktrace $w done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_cl -- python3 $R/bench.py --closed-loop --steps 400 --warmup 20 --no-cpu-baseline > $O/c2_closed_loop_under_rocprof.json 2> $O/ktrace_cl.err
find $O/ktrace_cl -name "*kernel_stats.csv" -exec cp {} $O/c2_closed_loop_kernel_stats.csv \;
for w in C2 C3 C4 fb_default; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 40 --warmup 5 $Q --no-kernel-events --no-closed-loop-leg > /dev/null 2> $O/pmc_${w}_$c.err
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_${w}_sq -- python3 $R/bench.py --workload $w --steps 40 --warmup 5 $Q --no-kernel-events --no-closed-loop-leg > /dev/null 2> $O/pmc_${w}_sq.err
  echo "pmc $w done"
done
cd $R
for w in C2 C3 C4 fb_default; do
  for d in pmc_${w}_WRITE_SIZE pmc_${w}_FETCH_SIZE pmc_${w}_sq; do [ -d $O/$d ] && python3 tools/pmc_summary.py $O/$d rollout; done > $O/${w}_pmc_summary.txt 2>&1 || true
done
python3 tools/pmc_json.py $O $T || true
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo done
