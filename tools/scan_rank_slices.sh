#!/bin/bash
# GPU box: a rank's eighth of the 7680x4320 teatime frame, alone and with two frames in flight, against
# the slice count (MRT_FORCE_SLICES; "auto" = the library's choice).  usage: bash tools/scan_rank_slices.sh 9 13 17 ...
for s in auto "$@"; do
  if [ $s = auto ]; then unset MRT_FORCE_SLICES; else export MRT_FORCE_SLICES=$s; fi
  echo -n "slices=$s: "
  timeout -k 10 200 python -u tools/time_rank_alone.py teatime8k --band balanced --reps 20 2>&1 | grep "world 8 band" || exit 1
done
