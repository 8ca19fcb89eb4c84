#!/bin/bash
# Device time of mrt_stage_intersect's kernels (teatime, 160 000 rays x 4034 objects) for a list
# of slice counts.  GPU box only.  usage: bash tools/scan_stage_slices.sh 16 32 51 64 ...
for s in "$@"; do
    echo -n "slices=$s: "
    MRT_STAGE_SLICES=$s timeout -k 10 120 python -u tools/time_stages.py teatime 2>&1 | grep "stage_intersect kernels" || exit 1
done
