#!/bin/bash
# GPU box: the library as built (main) against variants/<name>/libmrt_hip.so on the frames the
# strong-scaling bound is taken on (tools/time_rank_alone.py: the 4K and 8K teatime frames whole and
# as ranks 0 and w - 1 of 2, 4 and 8), three interleaved rounds.  usage: bash tools/ab_variant.sh <name>
V=${1:?variant name}
mkdir -p gpurun_out/ab
for round in 1 2 3; do
  for v in main $V; do
    if [ $v = main ]; then unset MRT_HIP_LIB; else export MRT_HIP_LIB=$PWD/variants/$v/libmrt_hip.so; fi
    timeout -k 10 200 python -u tools/time_rank_alone.py teatime4k --band balanced --reps 20 > gpurun_out/ab/$v.4k.$round.log 2>&1 || exit 1
    timeout -k 10 200 python -u tools/time_rank_alone.py teatime8k --band balanced --reps 20 > gpurun_out/ab/$v.8k.$round.log 2>&1 || exit 1
  done
done
python3 - "$V" <<'PY'
import glob, re, ast, sys
for v in ("main", sys.argv[1]):
    for f in ("4k", "8k"):
        rows = []
        for path in sorted(glob.glob(f"gpurun_out/ab/{v}.{f}.?.log")):
            t = open(path).read()
            whole = ast.literal_eval(re.search(r"whole frame: (\{.*\})", t).group(1))
            ranks = [float(x) for x in re.findall(r"rank alone ([0-9.]+) ms", t)]
            rows.append((round(whole["kernel_ms"], 3), round(whole["back_to_back_ms"], 3), round(whole["two_in_flight_ms"], 3), ranks))
        print(v, f, rows)
PY
