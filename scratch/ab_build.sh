#!/bin/bash
# Builds deephumor_amd/lib/libdeephumor_hip_base.so from a git revision of ONE source file (default HEAD's
# gemm_bf16.hip) + the current objects of all others, for same-box A/B timing via DEEPHUMOR_HIP_LIB.
set -e
cd "$(dirname "$0")/.."
f=${1:-gemm_bf16.hip}; rev=${2:-HEAD}
mkdir -p /tmp/ab/a/b /tmp/ab/include && cp include/*.h /tmp/ab/include/ && cp deephumor_amd/csrc/*.h /tmp/ab/a/b/ && git show $rev:deephumor_amd/csrc/$f > /tmp/ab/a/b/$f
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-comment -Iinclude -c /tmp/ab/a/b/$f -o /tmp/ab/base.o
objs=$(ls deephumor_amd/lib/*.o | grep -v "/${f%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o deephumor_amd/lib/libdeephumor_hip_base.so $objs /tmp/ab/base.o
echo built deephumor_amd/lib/libdeephumor_hip_base.so
