#!/bin/bash
# rocprofv3 evidence for mrt_stage_intersect's kernels (teatime at 400x400: 160 000 rays x 4034
# objects): kernel trace + stats, then one --pmc pass per counter group (never combined with a
# trace).  Usage (repo root, GPU box): bash tools/profile_stages.sh <tag>; results in
# gpurun_out/prof_stages_<tag>/.  Stops at the first pass that fails.
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_stages_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/time_stages.py teatime"
step() {
    local name=$1; shift
    "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name exit $rc"
    if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.log"; exit $rc; fi
}
step trace timeout -k 10 120 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $CMD
step pmc_valu timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d "$OUT/pmc_valu" -o pmc -- $CMD
step pmc_lds timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY -f csv -d "$OUT/pmc_lds" -o pmc -- $CMD
step pmc_fetch timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o pmc -- $CMD
step pmc_write timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o pmc -- $CMD
step pmc_grbm timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -f csv -d "$OUT/pmc_grbm" -o pmc -- $CMD
echo "all passes done"
