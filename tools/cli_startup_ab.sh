#!/bin/bash
# GPU box: what the 64 MiB fixed array of mrt_render_bands_device_noalloc costs a process at start-up.
# The CLI as built against a build whose array holds 1 Ki keys (variants/smallscratch/: a copy of
# include/ and opencl-raytracing_amd/ with MRT_NOALLOC_MAX_KEYS set to 1024u, `make` in the copy,
# libmrt_hip.so and mrraytracer copied here), three interleaved rounds on the reference's teatime
# scene; MRT_STATS=1 prints the split (mrt_create's own line, then the CLI's).
for r in 1 2 3; do
  for v in main small; do
    if [ $v = main ]; then bin=opencl-raytracing_amd/mrraytracer; else bin=variants/smallscratch/mrraytracer; fi
    echo "== $v round $r"
    MRT_STATS=1 timeout 60 $bin scenes/teatime.json /tmp/o_$v.png 2>&1 >/dev/null | tail -2
  done
done
cmp /tmp/o_main.png /tmp/o_small.png && echo same png
