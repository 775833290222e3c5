cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh r02m2 2 -- "reread=CCV_MPPI_KEEP=0" "keep=X=1" "keeploop_reread_epi=CCV_MPPI_LIB=$GRAFT_REPO_ROOT/_abl/lib_keeploop.so"
