#!/bin/bash
# k_fast with n extra scalar moves per quick-reject row step (diagnostic builds: make -C orb_slam3_amd/csrc extra-salu):
# kernel time and scalar / vector instruction counts per launch of 64 EuRoC frames.   bash tools/extra_salu.sh
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for n in 0 4 8 16; do
  if [ $n = 0 ]; then unset ORBX_LIB; else export ORBX_LIB=$R/build/liborbx_salu$n.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/salu_t$n -o p -- python3 $R/tools/prof_extract.py 64 6 > /dev/null 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/salu_c$n -o p -- python3 $R/tools/prof_extract.py 64 1 > /dev/null 2>&1
  echo "extra scalar moves per row step: $n"
  python3 $R/tools/kstats.py $O/salu_t$n/p_kernel_stats.csv | grep k_fast
  python3 $R/tools/pmc_summary.py $O/salu_c$n/p_counter_collection.csv | grep k_fast
done
