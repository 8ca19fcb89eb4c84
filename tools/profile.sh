#!/bin/bash
# Collects the rocprofv3 evidence for the render kernels on the GPU box:
#   1. kernel trace + stats (per-kernel durations),
#   2. hardware counters, one --pmc pass per run (never combined with traces).
# Usage (from the repo root, on the GPU box):  bash tools/profile.sh <tag>
# Results land in gpurun_out/prof_<tag>/; tools/summarize_profile.py condenses them into profiles/.
# Every pass runs under its own time limit and the script stops at the first pass that fails:
# nothing else is started on the GPU after a fault, an abort or a timeout.
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary"

step() {
    local name=$1; shift
    echo "== $name"
    "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name exit $rc"
    if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.log"; exit $rc; fi
}

step trace timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $CMD
run_pmc() {
    local name=$1; shift
    step "pmc_$name" timeout -k 10 240 rocprofv3 --pmc "$@" -f csv -d "$OUT/pmc_$name" -o pmc -- $CMD
}
run_pmc valu SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
# the accelerated entry points (grid build + grid / tile-cull / block-reject frames), kernel trace only
step trace_accel timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_accel" -o trace -- python3 $ROOT/tools/time_accel.py teatime4k soup16k
echo "all passes done"
