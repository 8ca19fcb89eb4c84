#!/bin/bash
# What the GPU's shader clock and power are while each kind of work runs (amd-smi beside
# tools/busy_loop.py, a sample every half second), and at rest.  GPU box only.
# usage: tools/sample_clocks.sh > gpurun_out/clocks.log
# (the last two sections need tools/filter_loop_bench, built here beforehand:
#  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/filter_loop_bench.hip -o tools/filter_loop_bench)
cd "$(dirname "$0")/.."
sample() {
    # socket power and the shader clock of each of the eight XCDs
    amd-smi metric -g 0 --power --clock 2>/dev/null | awk '
        /SOCKET_POWER/ { printf "power %s W; sclk", $2 }
        /GFX_[0-7]:/ { gfx = 1; next }
        gfx && /CLK:/ && !/MIN_CLK|MAX_CLK/ { printf " %s", $2; gfx = 0 }
        END { print " MHz" }'
}
echo "== at rest"; sample
for kind in full default stage; do
    echo "== $kind"
    marker=$(mktemp -u /tmp/busy_loop.XXXXXX)
    timeout -k 10 300 python tools/busy_loop.py "$kind" 8 "$marker" &
    pid=$!
    for w in $(seq 1 500); do [ -e "$marker" ] && break; sleep 0.5; done   # (the first import of torch on a fresh box takes minutes)
    sleep 1
    for k in 1 2 3 4 5 6 7 8; do sample; sleep 0.5; done
    rm -f "$marker"
    wait $pid || exit 1
done
# the bare filter loop of the general-ray kernel on plain and on random operands (tools/filter_loop_bench, built beforehand)
if [ -x tools/filter_loop_bench ]; then
    for mode in 4 5; do
        echo "== filter_loop_bench mode $mode"
        timeout -k 10 60 tools/filter_loop_bench $mode 6 > /tmp/filter_loop_$mode.log 2>&1 &
        pid=$!
        sleep 1.5
        for k in 1 2 3 4 5 6; do sample; sleep 0.3; done
        wait $pid || exit 1
        tail -1 /tmp/filter_loop_$mode.log
    done
fi
echo "== at rest again"; sleep 1; sample
