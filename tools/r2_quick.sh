#!/bin/bash
# Quick measurement pass on the GPU box: GPU tests, then kernel stats of the one-stream extraction at both sizes.
# bash tools/r2_quick.sh <tag>
set -e -o pipefail
TAG=${1:-q}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/${TAG}_tests.log 2>&1 || { tail -30 $O/${TAG}_tests.log; exit 1; }
tail -2 $O/${TAG}_tests.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_euroc64 -o p -- python3 $R/tools/prof_extract.py 64 5 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_1080p -o p -- python3 $R/tools/prof_extract.py 32 4 1920 1080 5000 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob
for d in ("${TAG}_euroc64","${TAG}_1080p"):
    f=glob.glob("$O/"+d+"/**/p_kernel_stats.csv",recursive=True)[0]
    print(d)
    tot=0
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0].replace("void ","")[:40]
        if n.startswith("k_"):
            print("  %-40s calls %4s avg %9.2f us" % (n, r["Calls"], float(r["AverageNs"])/1e3))
PY
