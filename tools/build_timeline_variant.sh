#!/bin/bash
# Builds variants/timeline/libmrt_hip.so: the library with trace_kernel's per-wave timeline compiled
# in (-DMRT_TIMELINE), for tools/wave_timeline.py.  Run `make -C opencl-raytracing_amd` first: every
# other object is the product's.
set -e
cd "$(dirname "$0")/../opencl-raytracing_amd"
mkdir -p ../variants/timeline
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -I../include -Icsrc -Wall -DMRT_TIMELINE -c csrc/mrt_kernels.hip -o ../variants/timeline/mrt_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/timeline/libmrt_hip.so ../variants/timeline/mrt_kernels.o \
  csrc/mrt_records.o csrc/mrt_accel.o csrc/mrt_host.o csrc/mrt_multi.o csrc/mrt_sequence.o csrc/mrt_stages.o \
  csrc/scene_loader.o csrc/json_min.o csrc/png_writer.o -lz -lpthread -ldl
