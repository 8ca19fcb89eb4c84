python3 - <<'P'
import sys; sys.path[:0] = ['.', 'tools']
import mkscene
open('/tmp/soup16k.json', 'w').write(mkscene.soup(16384, 16384))
P
for l in 0 0,0 0,0,0,0 0,0,0,0,0,0,0,0; do
  for g in peer host; do
    echo -n "ranks on GPU 0: $l gather $g: "
    MRT_GPU_LIST=$l MRT_GATHER=$g MRT_STATS=1 MRT_REPEAT=3 timeout 120 opencl-raytracing_amd/mrraytracer /tmp/soup16k.json /tmp/o.png 2>&1 >/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('render_ms_max','render_ms_sum','frame_ms','best_frame_ms','png_ms')})" || exit 1
  done
done
