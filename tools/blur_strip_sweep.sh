#!/bin/bash
# k_blur_levels time against the strip height: bash tools/blur_strip_sweep.sh "<frames> <batches> <w> <h> <nfeatures>" rows...
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CFG=$1; shift
export ORBX_BLUR_MODE=level
for rows in "$@"; do
  export ORBX_BLUR_STRIP_ROWS=$rows
  d=$O/strip_$rows
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/tools/prof_extract.py $CFG > /dev/null 2>&1
  echo "strip rows $rows ($CFG)"
  python3 $R/tools/kstats.py $d/p_kernel_stats.csv | grep -E "k_blur|k_describe"
done
