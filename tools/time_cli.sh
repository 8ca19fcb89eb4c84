#!/bin/bash
# Wall time of the drop-in CLI, scene file in, PNG out (GPU box).  usage: bash tools/time_cli.sh
set -u
ROOT=$(pwd)
python3 - <<'P'
import sys; sys.path[:0] = ['.', 'tools']
import mkscene
open('/tmp/t4k.json', 'w').write(mkscene.teatime(3840, 2160))
open('/tmp/soup16k.json', 'w').write(mkscene.soup(16384, 16384))
P
for scene in t4k soup16k; do
  for i in 1 2 3; do
    s=$(date +%s.%N)
    MRT_STATS=1 timeout 120 $ROOT/opencl-raytracing_amd/mrraytracer /tmp/$scene.json /tmp/$scene.png 2>/tmp/stats.txt >/dev/null || exit 1
    e=$(date +%s.%N)
    echo "$scene run $i: wall $(python3 -c "print(f'{$e - $s:.3f}')") s  $(tail -1 /tmp/stats.txt)"
  done
  MRT_GPUS=1 MRT_GPU_LIST=0 MRT_STATS=1 timeout 120 $ROOT/opencl-raytracing_amd/mrraytracer /tmp/$scene.json /tmp/${scene}_m.png 2>/tmp/stats.txt >/dev/null || exit 1
  echo "$scene through mrt_multi, one rank: $(tail -1 /tmp/stats.txt)"
  cmp /tmp/$scene.png /tmp/${scene}_m.png && echo "$scene: same PNG through mrt_multi"
done
