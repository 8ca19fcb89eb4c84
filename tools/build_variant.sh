#!/bin/bash
# Builds a tuning variant of libmrt_hip.so with extra -D flags into variants/<name>/.
# Usage: [MRT_SRC=<dir with the csrc files>] tools/build_variant.sh <name> [-DMRT_...=...]...
# then   MRT_HIP_LIB=variants/<name>/libmrt_hip.so python bench.py
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${MRT_SRC:-$ROOT/opencl-raytracing_amd/csrc}
OUT=$ROOT/variants/$NAME
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -I$SRC $*"
pids=()
for f in mrt_kernels mrt_records mrt_accel mrt_host mrt_multi mrt_sequence mrt_stages; do /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o & pids+=($!); done
for f in scene_loader json_min png_writer; do g++ -O2 -std=c++17 -fPIC -I$ROOT/include -I$SRC -c $SRC/$f.cpp -o $OUT/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libmrt_hip.so $OUT/*.o -lz -lpthread -ldl
rm -f $OUT/*.o
echo "built $OUT/libmrt_hip.so"
