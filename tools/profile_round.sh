set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r01e}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
for w in C3 C4; do python3 $R/bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $O/${w}_bench.json 2>/dev/null; done
python3 $R/bench.py --closed-loop --no-cpu-baseline > $O/c2_closed_loop_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_bench_under_rocprof.json 2> $O/ktrace.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_s -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_s.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_t -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_t.err || true
cd $R
for d in pmc_w pmc_f pmc_s pmc_t; do python3 tools/pmc_summary.py $O/$d rollout; python3 tools/pmc_summary.py $O/$d finalize; done > $O/c2_pmc_summary.txt 2>&1 || true
find $O/ktrace -name "*kernel_stats.csv" -exec cp {} $O/c2_kernel_stats.csv \;
# keep only small files
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
echo done
