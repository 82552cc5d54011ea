#!/bin/bash
# Collects one measurement round on the GPU box: bash tools/run_profiles.sh r02   (then: python tools/collect_profiles.py r02)
# Needs the diagnostic libraries: make -C orb_slam3_amd/csrc stamps phase-stops
# Every step has its own time limit; the steps are chained so that a failure ends the run.
set -e -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
echo bench; timeout -k 10 420 python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo bench-prof; timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --cpu-pairs 0 --no-host-path > /dev/null 2>&1
echo euroc; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_euroc64 -o p -- python3 $R/tools/prof_extract.py 64 5 > /dev/null 2>&1
echo 1080p; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_1080p -o p -- python3 $R/tools/prof_extract.py 32 4 1920 1080 5000 > /dev/null 2>&1
echo pmc; timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o p -- python3 $R/tools/prof_extract.py 64 2 > /dev/null 2>&1
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o p -- python3 $R/tools/prof_extract.py 64 2 > /dev/null 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU --output-format csv -d $O/${TAG}_sq1 -o p -- python3 $R/tools/prof_extract.py 64 2 > /dev/null 2>&1
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_sq2 -o p -- python3 $R/tools/prof_extract.py 64 2 > /dev/null 2>&1
echo phases; ORBX_LIB=$R/build/liborbx_stamps.so timeout -k 10 200 python3 $R/tools/prof_fast_phases.py 64 > $O/${TAG}_phase_cycles.txt 2>&1
bash $R/tools/prof_phase_insts.sh > /dev/null 2>&1
python3 $R/tools/phase_insts_table.py $O/phase_insts > $O/${TAG}_phase_insts.txt 2>&1
echo host; timeout -k 10 200 python3 $R/tools/host_path_rate.py > $O/host_rate.txt 2>&1
timeout -k 10 100 python3 $R/tools/prof_stage.py 64 > $O/${TAG}_stage.txt 2>&1
echo collected $TAG
