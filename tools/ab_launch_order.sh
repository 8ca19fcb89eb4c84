#!/bin/bash
# GPU box: trace_kernel's launch order, tile-major (the product) against slice-major
# (variants/slice_major, built with -DMRT_TILE_MAJOR=0), three interleaved rounds.
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for v in main slice_major; do
    if [ $v = main ]; then unset MRT_HIP_LIB; else export MRT_HIP_LIB=$PWD/variants/$v/libmrt_hip.so; fi
    timeout -k 10 200 python -u tools/time_rank_alone.py teatime4k --band balanced --reps 20 > gpurun_out/ab/$v.4k.$round.log 2>&1 || exit 1
    timeout -k 10 200 python -u tools/time_rank_alone.py teatime8k --band balanced --reps 20 > gpurun_out/ab/$v.8k.$round.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import glob, re, ast
for v in ("main", "slice_major"):
    for f in ("4k", "8k"):
        rows = []
        for path in sorted(glob.glob(f"gpurun_out/ab/{v}.{f}.?.log")):
            t = open(path).read()
            whole = ast.literal_eval(re.search(r"whole frame: (\{.*\})", t).group(1))
            ranks = [float(x) for x in re.findall(r"rank alone ([0-9.]+) ms", t)]
            rows.append((round(whole["kernel_ms"], 3), round(whole["back_to_back_ms"], 3), round(whole["two_in_flight_ms"], 3), ranks))
        print(v, f, rows)
PY
