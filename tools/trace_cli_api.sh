set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_cli_hip
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --hip-trace --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- $ROOT/opencl-raytracing_amd/mrraytracer $ROOT/scenes/teatime.json /tmp/o.png > "$OUT/run.log" 2>&1
echo "exit $?"
ls "$OUT/trace"
