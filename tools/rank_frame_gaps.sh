#!/bin/bash
# GPU box: kernel-trace of one rank's eighth of the 7680x4320 teatime frame rendered alone, and
# of the whole frame (tools/time_rank_alone.py), to set the kernels' own durations against the
# time per frame: what is left is what the boundaries between launches cost.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_rank_frame
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python3 $ROOT/tools/time_rank_alone.py teatime8k --band balanced --reps 20 > "$OUT/run.log" 2>&1
echo "exit $?"
tail -4 "$OUT/run.log"
