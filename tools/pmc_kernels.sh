#!/bin/bash
# Instruction and stall counters of every extractor kernel for 64 EuRoC-sized frames (product library):
#   bash tools/pmc_kernels.sh <tag>  -> gpurun_out/<tag>_a, <tag>_b ; python3 tools/pmc_table.py gpurun_out/<tag>
set -e -o pipefail
TAG=${1:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${TAG}_a -o p -- python3 $R/tools/prof_extract.py 64 1 > /dev/null 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/${TAG}_b -o p -- python3 $R/tools/prof_extract.py 64 1 > /dev/null 2>&1
echo done
