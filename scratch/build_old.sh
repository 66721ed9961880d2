#!/bin/bash
# Builds detectinblur_amd/libdib_hip_old.so from the dib_blur.hip of a git revision (default HEAD), for scratch/t_ab.py A/Bs
# against the working tree:   bash scratch/build_old.sh [rev]
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/k/old
git -C $root show $rev:detectinblur_amd/csrc/dib_blur.hip > /tmp/k/old/dib_blur.hip
git -C $root show $rev:detectinblur_amd/csrc/dib_common.h | sed "s#\"../../include/dib.h\"#\"$root/include/dib.h\"#" > /tmp/k/old/dib_common.h
cd /tmp/k/old && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c dib_blur.hip -o blur_old.o
cd $root/detectinblur_amd/csrc && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 dib_compact.o /tmp/k/old/blur_old.o dib_boxes.o dib_raster.o dib_roi.o dib_eltwise.o dib_epilogue.o dib_error.o -o ../libdib_hip_old.so
echo built detectinblur_amd/libdib_hip_old.so from $rev
