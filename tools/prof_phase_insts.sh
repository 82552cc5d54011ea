#!/bin/bash
# Instruction counters of k_fast / k_describe cut after each phase (make -C orb_slam3_amd/csrc phase-stops first):
#   bash tools/prof_phase_insts.sh     -> gpurun_out/phase_insts/{f,d}{0..4}/   (0 = the product library)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/phase_insts
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in f1 f2 f3 f4 d1 d2 d3 d4 p0; do
  case $v in
    f*) export ORBX_LIB=$R/build/liborbx_fstop${v#f}.so ;;
    d*) export ORBX_LIB=$R/build/liborbx_dstop${v#d}.so ;;
    p0) unset ORBX_LIB ;;
  esac
  echo $v
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_LDS --output-format csv -d $O/$v -o p -- python3 $R/tools/prof_extract.py 64 1 > /dev/null 2>&1
done
echo done
