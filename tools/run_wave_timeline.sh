#!/bin/bash
# GPU box: the wave timeline of the 4K frame, of one rank of eight of the 8K frame, and of the 8K frame.
export MRT_HIP_LIB=$PWD/variants/timeline/libmrt_hip.so
mkdir -p gpurun_out
timeout -k 10 120 python -u tools/wave_timeline.py teatime4k --json gpurun_out/timeline_4k.json --dump gpurun_out/timeline_4k.npy > gpurun_out/timeline_4k.log 2>&1 &&
timeout -k 10 120 python -u tools/wave_timeline.py teatime8k --world 8 --rank 0 --json gpurun_out/timeline_8k_w8.json > gpurun_out/timeline_8k_w8.log 2>&1 &&
timeout -k 10 120 python -u tools/wave_timeline.py teatime8k --json gpurun_out/timeline_8k.json > gpurun_out/timeline_8k.log 2>&1
rc=$?
tail -n 1 gpurun_out/timeline_4k.log gpurun_out/timeline_8k_w8.log gpurun_out/timeline_8k.log
exit $rc
