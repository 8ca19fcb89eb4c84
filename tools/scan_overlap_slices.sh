#!/bin/bash
# Slice counts (MRT_FORCE_SLICES) against frames in flight, tools/time_overlap.py per setting.  GPU box only.
run() { # kind xres yres steps slices...
  local x=$1 y=$2 n=$3; shift 3
  for s in "$@"; do
    echo "${x}x${y} slices $s"
    MRT_FORCE_SLICES=$s timeout -k 10 100 python -u tools/time_overlap.py teatime $x $y $n 2>&1 | grep frames_in_flight | head -2 || exit 1
  done
}
run 1920 1080 100 12 24 32 64 && run 7680 4320 20 2 3 4 6 9 && run 400 400 300 32 64 126
