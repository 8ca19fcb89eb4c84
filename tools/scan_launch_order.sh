#!/bin/bash
# GPU box: frame time against the slice count for both launch orders of trace_kernel
# (variants/tile_major and variants/slice_major: -DMRT_TILE_MAJOR=1 / 0).  Odd counts: in tile-major
# order block b is slice b mod S of tile b / S and runs on XCD b mod 8, so a count that shares a
# factor with 8 gives every XCD the same few slices of every tile.
mkdir -p gpurun_out/order
for v in ${VARIANTS:-tile_major slice_major}; do
  export MRT_HIP_LIB=$PWD/variants/$v/libmrt_hip.so
  timeout -k 10 200 python -u tools/scan_slices.py teatime 3840 2160 --shape wide 7 9 11 13 15 17 19 21 25 33 > gpurun_out/order/$v.4k.log 2>&1 || exit 1
  timeout -k 10 200 python -u tools/scan_slices.py teatime 7680 4320 --shape wide 3 5 7 9 11 13 > gpurun_out/order/$v.8k.log 2>&1 || exit 1
  timeout -k 10 200 python -u tools/scan_slices.py teatime 7680 4320 --world 8 --shape wide 11 17 23 31 37 43 > gpurun_out/order/$v.8k_w8.log 2>&1 || exit 1
  timeout -k 10 200 python -u tools/scan_slices.py teatime 1920 1080 --shape narrow 17 23 27 33 37 43 51 > gpurun_out/order/$v.1080.log 2>&1 || exit 1
  timeout -k 10 200 python -u tools/scan_slices.py teatime 400 400 --shape narrow 33 43 63 95 126 > gpurun_out/order/$v.400.log 2>&1 || exit 1
done
tail -n 20 gpurun_out/order/*.log
