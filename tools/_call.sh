#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 1100 bash tools/profile_r02.sh r02x
