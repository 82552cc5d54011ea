#!/bin/bash
# Per-patch blur inside the descriptor kernel against the whole-level blur kernel, per configuration:
#   bash tools/blur_modes.sh [tag]     (on the GPU box; writes gpurun_out/<tag>_blur_modes.txt)
# Each run is a rocprofv3 kernel trace of tools/prof_extract.py: <frames> <batches> <width> <height> <nfeatures>.
set -e -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
OUT=$O/${TAG}_blur_modes.txt
: > $OUT
for cfg in "64 6 752 480 1000" "64 6 752 480 5000" "16 6 1920 1080 5000" "64 4 1920 1080 5000"; do
  for mode in patch level; do
    export ORBX_BLUR_MODE=$mode
    d=$O/${TAG}_blur_$(echo $cfg | tr ' ' '_')_$mode
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 $R/tools/prof_extract.py $cfg > /dev/null 2>&1
    echo "== frames batches width height nfeatures = $cfg   ORBX_BLUR_MODE=$mode ${ORBX_BLUR_STRIP_ROWS:+strip rows $ORBX_BLUR_STRIP_ROWS}" >> $OUT
    python3 $R/tools/kstats.py $d/p_kernel_stats.csv | grep -E "k_describe|k_blur|k_fast|k_resize|k_quadtree|k_layout|k_copy" >> $OUT
  done
done
cat $OUT
