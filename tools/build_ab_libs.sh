#!/bin/bash
# The A/B builds the tools/gpu_r*_*.sh scripts alternate with raweditor_amd/librawdev.so (same sources, same ABI, one switch each;
# load one with RAWDEV_LIB=path).  Run on the build host (hipcc cross-compiles gfx950); the .so files travel with gpurun.
#   tools/librawdev_r4nolut.so     -DRD_Q8_LUT=0    RGBA8 / RGB8 codes by the transcendental shortcut instead of the LDS threshold table
#   tools/librawdev_r4nof16lut.so  -DRD_F16_LUT=0   RGBA-f16 halves + codes by the shortcut instead of the two-level tables
#   tools/librawdev_r5base.so      -DRD_F32_PARK=0 -DRD_WAVES_PER_EU=   round 4's register arrangement: the f32 kernel's slider
#                                                   uniforms stay in SGPRs, no waves-per-SIMD target for the compiler
#   tools/librawdev_r5nopark.so    -DRD_F32_PARK=0  the waves-per-SIMD attribute alone (= the product since the A/B)
#   tools/librawdev_r5park.so      -DRD_F32_PARK=1  six slider uniforms of the f32 multi-frame kernel parked in VGPRs (dropped)
#   tools/librawdev_r6rgbplain.so  -DRD_RGB8_ST_PLAIN  RGB8 surface stored with write-back instead of non-temporal stores (dropped:
#                                                   profiles/r06_rgb8_ragged_ab.txt)
# `bash tools/build_ab_libs.sh r5` builds only the round-5 pair, `... r6` only the round-6 build.
set -eu
cd "$(dirname "$0")/.."
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -Wall -Wextra -pthread -ldl"
if [ "${1:-all}" = r6 ]; then
  /opt/rocm/bin/hipcc $FLAGS -DRD_RGB8_ST_PLAIN -o tools/librawdev_r6rgbplain.so raweditor_amd/csrc/rawdev.hip
  ls -la tools/librawdev_r6*.so
  exit 0
fi
if [ "${1:-all}" != r5 ]; then
  /opt/rocm/bin/hipcc $FLAGS -DRD_Q8_LUT=0 -o tools/librawdev_r4nolut.so raweditor_amd/csrc/rawdev.hip
  /opt/rocm/bin/hipcc $FLAGS -DRD_F16_LUT=0 -o tools/librawdev_r4nof16lut.so raweditor_amd/csrc/rawdev.hip
fi
/opt/rocm/bin/hipcc $FLAGS -DRD_F32_PARK=0 "-DRD_WAVES_PER_EU=" -o tools/librawdev_r5base.so raweditor_amd/csrc/rawdev.hip &
/opt/rocm/bin/hipcc $FLAGS -DRD_F32_PARK=0 -o tools/librawdev_r5nopark.so raweditor_amd/csrc/rawdev.hip &
/opt/rocm/bin/hipcc $FLAGS -DRD_F32_PARK=1 -o tools/librawdev_r5park.so raweditor_amd/csrc/rawdev.hip &
wait
ls -la tools/librawdev_r*.so
