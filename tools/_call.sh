#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5n
STAMP_SET=epi timeout -k 10 300 python3 tools/stamps_r4.py C2 > gpurun_out/r5n/epi_c2.txt 2>&1; cat gpurun_out/r5n/epi_c2.txt
