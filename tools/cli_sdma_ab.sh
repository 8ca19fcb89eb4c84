python3 - <<'P'
import sys; sys.path[:0] = ['.', 'tools']
import mkscene
open('/tmp/t4k.json', 'w').write(mkscene.teatime(3840, 2160))
open('/tmp/soup16k.json', 'w').write(mkscene.soup(16384, 16384))
P
for scene in scenes/teatime.json /tmp/t4k.json /tmp/soup16k.json; do
  for r in 1 2 3; do
    for sdma in 1 0; do
      echo -n "$scene sdma=$sdma: "
      HSA_ENABLE_SDMA=$sdma MRT_STATS=1 timeout 120 opencl-raytracing_amd/mrraytracer $scene /tmp/o_$sdma.png 2>&1 >/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('total_ms','create_wait_ms','upload_ms','first_frame_ms','png_ms')})"
    done
  done
  cmp /tmp/o_0.png /tmp/o_1.png && echo same
done
