# VALU / wait / instruction-fetch counters of the C4 rollout kernel: bash tools/pmc_c4.sh <tag> [workload] [K]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_c4}
W=${2:-C4}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
K=${3:+--samples-per-gpu $3}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 --output-format csv -d $O/pmc_s -- python3 $R/bench.py --workload $W $K --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_s.err
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_i -- python3 $R/bench.py --workload $W $K --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/pmc_i.err
cd $R
for d in pmc_s pmc_i; do python3 tools/pmc_summary.py $O/$d rollout; done > $O/pmc_summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
echo done
