cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout -k 10 500 python tests/soak_determinism.py 20000 2>&1 | tee gpurun_out/r02k/soak_determinism.txt
