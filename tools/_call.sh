cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout -k 10 900 python -m pytest tests -m gpu -q -s -k "resident or world_8 or leak or memory" > gpurun_out/r02b/pytest2.log 2>&1; rc=$?
tail -30 gpurun_out/r02b/pytest2.log
exit $rc
