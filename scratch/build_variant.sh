#!/bin/bash
# Diagnostic / experimental build of the device library: scratch/build_variant.sh <name> [extra hipcc flags for dib_blur.hip ...]
# -> scratch/libdib_hip_<name>.so (select it with DIB_HIP_LIB=...); the other objects are the product build's.
set -e
name=$1; shift
cd "$(dirname "$0")/../detectinblur_amd/csrc"
make -j4 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c dib_blur.hip -o /tmp/dib_blur_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $(ls *.o | grep -v -e '^dib_blur.o$' -e '^dib_blur_portable.o$') /tmp/dib_blur_$name.o -o ../../scratch/libdib_hip_$name.so
ls -la ../../scratch/libdib_hip_$name.so
