#!/bin/bash
# trace_kernel with other lane blocks than 4x4 (variants/<name>/libmrt_hip.so from tools/build_variant.sh
# with -DMRT_TRACE_LANE_X/-DMRT_TRACE_LANE_Y): kernel time on four frames, then one counter pass
# per variant on the 4K teatime frame.  GPU box only; stops at the first failure.
# Usage: bash tools/lane_shape_experiment.sh <out dir under gpurun_out> <variant>...
set -u
OUT=$PWD/gpurun_out/$1; shift
mkdir -p "$OUT"
ROOT=$PWD
for v in "$@"; do
  export MRT_HIP_LIB=$ROOT/variants/$v/libmrt_hip.so
  echo "== $v" | tee -a "$OUT/times.log"
  for frame in "teatime 3840 2160" "teatime 7680 4320" "teatime 400 400" "soup 16384 16384"; do
    timeout -k 10 120 python3 -u tools/time_flags.py $frame 0 2>/dev/null | sed "s/^/$frame: /" | tee -a "$OUT/times.log" || exit 1
  done
done
export MRT_HIP_LIB=$ROOT/variants/$1/libmrt_hip.so
echo "== $1 with MRT_FORCE_SLICES=4 (as many waves as an 8x8 lane block gives)" | tee -a "$OUT/times.log"
MRT_FORCE_SLICES=4 timeout -k 10 120 python3 -u tools/time_flags.py teatime 3840 2160 0 2>/dev/null | tee -a "$OUT/times.log" || exit 1
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export MRT_HIP_LIB=$ROOT/variants/$v/libmrt_hip.so
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      -f csv -d "$OUT/pmc_$v" -o pmc -- python3 $ROOT/tools/time_flags.py teatime 3840 2160 0 > "$OUT/pmc_$v.log" 2>&1 || { tail -5 "$OUT/pmc_$v.log"; exit 1; }
  echo "pmc $v done"
done
