#!/bin/bash
# Runs bench.py (headline only) against each variants/*/libmrt_hip.so, three interleaved rounds.
for round in 1 2 3; do
  for v in variants/*/; do
    n=$(basename $v)
    MRT_HIP_LIB=$PWD/$v/libmrt_hip.so timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-secondary --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['value']), round(d['roofline']['kernel_ms'],4))"
  done
done
