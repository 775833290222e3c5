# kernel-trace summaries of the C2 and C4 bench runs (profiles/<tag>_*): bash tools/profile_extra.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r01i}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
python3 $R/bench.py --workload C3 --steps 100 --warmup 10 --no-cpu-baseline > $O/C3_bench.json 2>/dev/null
python3 $R/bench.py --workload C4 --steps 100 --warmup 10 --no-cpu-baseline > $O/C4_bench.json 2>/dev/null
python3 $R/bench.py --samples-per-gpu 524288 --steps 60 --warmup 10 --no-cpu-baseline > $O/c2_K524288_bench.json 2>/dev/null
python3 $R/bench.py --closed-loop --no-cpu-baseline > $O/c2_closed_loop_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c2 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_bench_under_rocprof.json 2> $O/ktrace_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c4 -- python3 $R/bench.py --workload C4 --steps 100 --warmup 10 --no-cpu-baseline > $O/C4_bench_under_rocprof.json 2> $O/ktrace_c4.err
find $O/ktrace_c2 -name "*kernel_stats.csv" -exec cp {} $O/c2_kernel_stats.csv \;
find $O/ktrace_c4 -name "*kernel_stats.csv" -exec cp {} $O/C4_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo done
