#!/bin/bash
# Quick loop on the GPU box: extractor parity tests, then the per-kernel averages of 64 EuRoC-sized frames.
#   bash tools/r2_quick.sh <tag>   -> gpurun_out/<tag>/p_kernel_stats.csv
set -e -o pipefail
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_extract.py -x -q --timeout 120 --timeout-method thread > gpurun_out/${TAG}_test.log 2>&1 || { tail -30 gpurun_out/${TAG}_test.log; exit 1; }
tail -1 gpurun_out/${TAG}_test.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o p -- python3 $R/tools/prof_extract.py 64 5 > /dev/null 2>&1
python3 $R/tools/kstats.py $R/gpurun_out/$TAG/p_kernel_stats.csv
