#!/bin/bash
# Collects the rocprofv3 evidence for the render kernels on the GPU box:
#   1. kernel trace + stats (per-kernel durations),
#   2. hardware counters, one --pmc pass per run (never combined with traces).
# Usage (from the repo root, on the GPU box):  bash tools/profile.sh <tag>
# Results land in gpurun_out/prof_<tag>/; copy the CSV summaries into profiles/.
set -u
TAG=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"

timeout -s KILL 240 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
echo "trace exit $?"

run_pmc() {
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc -- $CMD > "$OUT/pmc_$name.log" 2>&1
    echo "pmc $name exit $?"
}
run_pmc valu SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls -R "$OUT" | head -60
