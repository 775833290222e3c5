cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
./tools/microbench/hbm_write | tee gpurun_out/r02f/hbm_write.txt
