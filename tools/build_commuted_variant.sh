#!/bin/bash
# Builds variants/<name>/libmrt_hip.so with tools/commute_fma.py applied to the device assembly
# of the hot kernels (mrt_kernels.hip, mrt_stages.hip): hipcc to device assembly, the pass, then
# the same steps hipcc takes itself (assemble, link the code object, bundle, host compile with the
# bundle embedded).  Usage: tools/build_commuted_variant.sh <name> [-D...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/opencl-raytracing_amd/csrc
OUT=$ROOT/variants/$NAME
LLVM=/opt/rocm/lib/llvm/bin
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$ROOT/include -I$SRC $*"
for f in mrt_kernels mrt_stages; do
  (
    /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S $SRC/$f.hip -o $OUT/$f.dev.s 2>/dev/null
    python3 $ROOT/tools/commute_fma.py $OUT/$f.dev.s $OUT/$f.dev2.s
    $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $OUT/$f.dev2.s -o $OUT/$f.dev.o
    $LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $OUT/$f.dev.out $OUT/$f.dev.o
    $LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$OUT/$f.dev.out -output=$OUT/$f.hipfb
    /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -c $SRC/$f.hip -Xclang -fcuda-include-gpubinary -Xclang $OUT/$f.hipfb -o $OUT/$f.o 2>/dev/null
  ) &
done
for f in mrt_records mrt_accel mrt_host mrt_multi mrt_sequence; do /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o 2>/dev/null & done
for f in scene_loader json_min png_writer; do g++ -O2 -std=c++17 -fPIC -I$ROOT/include -I$SRC -c $SRC/$f.cpp -o $OUT/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libmrt_hip.so $OUT/mrt_kernels.o $OUT/mrt_stages.o $OUT/mrt_records.o $OUT/mrt_accel.o $OUT/mrt_host.o $OUT/mrt_multi.o $OUT/mrt_sequence.o $OUT/scene_loader.o $OUT/json_min.o $OUT/png_writer.o -lz -lpthread -ldl
rm -f $OUT/*.o $OUT/*.out $OUT/*.hipfb
echo "built $OUT/libmrt_hip.so"
