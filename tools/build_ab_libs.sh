#!/bin/bash
# The A/B builds the tools/gpu_r4_*.sh scripts alternate with raweditor_amd/librawdev.so (same sources, same ABI, one switch each;
# load one with RAWDEV_LIB=path).  Run on the build host (hipcc cross-compiles gfx950); the .so files travel with gpurun.
#   tools/librawdev_r4nolut.so     -DRD_Q8_LUT=0    RGBA8 / RGB8 codes by the transcendental shortcut instead of the LDS threshold table
#   tools/librawdev_r4nof16lut.so  -DRD_F16_LUT=0   RGBA-f16 halves + codes by the shortcut instead of the two-level tables
set -eu
cd "$(dirname "$0")/.."
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wall -Wextra -pthread -ldl"
/opt/rocm/bin/hipcc $FLAGS -DRD_Q8_LUT=0 -o tools/librawdev_r4nolut.so raweditor_amd/csrc/rawdev.hip
/opt/rocm/bin/hipcc $FLAGS -DRD_F16_LUT=0 -o tools/librawdev_r4nof16lut.so raweditor_amd/csrc/rawdev.hip
ls -la tools/librawdev_r4nolut.so tools/librawdev_r4nof16lut.so
