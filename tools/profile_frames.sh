#!/bin/bash
# rocprofv3 evidence for the other graded frames: the 16384^2 soup and teatime at 7680x4320, default
# kernel (flags 0) and MRT_FLAG_FULL_TEST (0x20): kernel trace + stats, then one --pmc pass of the
# VALU counters (never combined with a trace).  Usage (repo root, GPU box):
#   bash tools/profile_frames.sh <tag>     -> gpurun_out/prof_frames_<tag>/{soup16k,teatime8k}/...
# Stops at the first pass that fails.
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_frames_$TAG
cd /tmp && export TMPDIR=/tmp
step() {
    local name=$1; shift
    "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name exit $rc"
    if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.log"; exit $rc; fi
}
mkdir -p "$OUT"
for frame in "soup16k soup 16384 16384" "teatime8k teatime 7680 4320"; do
    set -- $frame
    name=$1; kind=$2; x=$3; y=$4
    CMD="python3 $ROOT/tools/time_flags.py $kind $x $y 0 0x20"
    step ${name}_trace timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/$name/trace" -o trace -- $CMD
    step ${name}_pmc_valu timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d "$OUT/$name/pmc_valu" -o pmc -- $CMD
    step ${name}_pmc_lds timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -f csv -d "$OUT/$name/pmc_lds" -o pmc -- $CMD
done
echo "all passes done"
