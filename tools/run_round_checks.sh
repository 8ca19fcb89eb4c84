set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/r2_gpu11.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_gpu11.log
python tools/mkscene.py --help > /dev/null 2>&1
python - <<'P'
import sys; sys.path[:0]=['.','tools']
import mkscene
open('/tmp/t4k.json','w').write(mkscene.teatime(3840,2160))
P
for i in 1 2; do MRT_STATS=1 timeout 60 opencl-raytracing_amd/mrraytracer /tmp/t4k.json /tmp/o.png 2>&1 >/dev/null | tail -1; done | tee gpurun_out/r2_cli_stats.log
MRT_STATS=1 MRT_ACCEL=grid timeout 60 opencl-raytracing_amd/mrraytracer /tmp/t4k.json /tmp/o2.png 2>&1 >/dev/null | tail -1 | tee -a gpurun_out/r2_cli_stats.log
cmp /tmp/o.png /tmp/o2.png && echo "png identical" | tee -a gpurun_out/r2_cli_stats.log
