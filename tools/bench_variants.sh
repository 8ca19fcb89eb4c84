#!/bin/bash
# Runs bench.py (headline only) against each variants/*/libmrt_hip.so, three interleaved rounds.
# What a variant writes to stderr is kept (gpurun_out/variants/<name>.round<k>.err) and a variant
# that prints no JSON line is reported as such, with the last line of its stderr, not dropped.
mkdir -p gpurun_out/variants
for round in 1 2 3; do
  for v in variants/*/; do
    n=$(basename $v)
    err=gpurun_out/variants/$n.round$round.err
    MRT_HIP_LIB=$PWD/$v/libmrt_hip.so timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-secondary --steps 30 2>$err \
      | python3 -c "
import json, sys
lines = [l for l in sys.stdin.read().splitlines() if l.startswith('{')]
if lines:
    d = json.loads(lines[-1]); print('$n', round(d['value']), round(d['roofline']['kernel_ms'], 4))
else:
    print('$n', 'NO RESULT:', (open('$err').read().strip().splitlines() or ['(stderr empty)'])[-1][:200])
"
  done
done
