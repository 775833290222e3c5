# Round-2 measurement set (profiles/<tag>_*): bash tools/profile_r02.sh <tag>
#   bench lines (default run, the driver's --steps 20 --warmup 5, C3, C4, K = 524 288 on one device, closed loop)
#   rocprofv3 --kernel-trace --stats of the C2 / C3 / C4 bench commands
#   rocprofv3 --pmc passes (each counter set its own run): WRITE_SIZE, FETCH_SIZE for C2 / C3 / C4; SQ sets for C2 and C4
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r02a}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/c2_bench_driver_args.json 2>> $O/c2_bench.err
python3 $R/bench.py --workload C3 --steps 100 --warmup 10 --no-cpu-baseline > $O/C3_bench.json 2>/dev/null
python3 $R/bench.py --workload C4 --steps 100 --warmup 10 --no-cpu-baseline > $O/C4_bench.json 2>/dev/null
python3 $R/bench.py --samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline > $O/c2_K524288_bench.json 2>/dev/null
python3 $R/bench.py --path straight --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_straight_bench.json 2>/dev/null
# the plain kernel the host falls back to when a measured loop period crosses |w|max dt = pi/4 (C2 limits: 0.3927 s)
python3 $R/bench.py --dt 0.41 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg > $O/c2_dt0.41_fallback_bench.json 2>/dev/null
python3 $R/bench.py --dt 0.39 --steps 200 --warmup 20 --no-cpu-baseline --no-closed-loop-leg > $O/c2_dt0.39_bench.json 2>/dev/null
echo "bench lines done"
for w in C2 C3 C4; do
  st=200; [ $w = C4 ] && st=100
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$w -- python3 $R/bench.py --workload $w --steps $st --warmup 20 --no-cpu-baseline > $O/${w}_bench_under_rocprof.json 2> $O/ktrace_$w.err
  find $O/ktrace_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
  echo "ktrace $w done"
done
for w in C2 C3 C4; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python3 $R/bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --no-closed-loop-leg > /dev/null 2> $O/pmc_${w}_$c.err
  done
  echo "pmc traffic $w done"
done
for w in C2 C4; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_${w}_sq -- python3 $R/bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --no-closed-loop-leg > /dev/null 2> $O/pmc_${w}_sq.err
  echo "pmc sq $w done"
done
cd $R
for w in C2 C3 C4; do
  for d in pmc_${w}_WRITE_SIZE pmc_${w}_FETCH_SIZE pmc_${w}_sq; do [ -d $O/$d ] && python3 tools/pmc_summary.py $O/$d rollout; done > $O/${w}_pmc_summary.txt 2>&1 || true
done
python3 tools/pmc_json.py $O $T || true
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo done
