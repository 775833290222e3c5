# A/B of library builds on one box, alternating: bash tools/ab_bench.sh <tag> <reps> -- "<name>=<env assignments>" ...
# every variant runs `bench.py $BENCH_ARGS` per repetition; one JSON line per run in gpurun_out/<tag>/ab.jsonl
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; REPS=$2; shift 3
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}
    line=$(env $envs python3 $R/bench.py ${BENCH_ARGS:---steps 400 --warmup 20 --no-cpu-baseline} 2>>$O/ab.err || echo '{"error": true}')
    echo "{\"variant\": \"$name\", \"rep\": $rep, \"bench\": $line}" >> $O/ab.jsonl
    echo "$name rep $rep: $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d.get("roofline",{}); print("%.1f us/iter  kernel %.2f us  frac %.3f" % (1e3*d.get("ms_per_step",0), r.get("kernel_avg_us",0), r.get("frac") or 0))' 2>/dev/null)"
  done
done
