#!/bin/bash
# Round 5, native-size ragged batch + the blur's inner-loop budget: one box, outputs under gpurun_out/native_r5/
O=$GRAFT_REPO_ROOT/gpurun_out/native_r5; mkdir -p $O
cd $GRAFT_REPO_ROOT
python scratch/t_native_ab.py default default@DIB_FLAT_GRID=0 > $O/native_ab.txt 2>&1
DIB_HIP_LIB=scratch/libdib_hip_tl.so python scratch/timeline_native.py > $O/timeline_flat.txt 2>&1
DIB_FLAT_GRID=0 DIB_HIP_LIB=scratch/libdib_hip_tl.so python scratch/timeline_native.py > $O/timeline_2d.txt 2>&1
scratch/ubench/ub_tap > $O/ub_tap.txt 2>&1
scratch/ubench/ub_mix > $O/ub_mix.txt 2>&1
scratch/ubench/ub_place > $O/ub_place.txt 2>&1
scratch/ubench/ub_fill > $O/ub_fill.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export T_NATIVE_CHILD=1
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $GRAFT_REPO_ROOT/scratch/t_native_ab.py > $O/trace.log 2>&1
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/native_kernel_stats.csv
rm -rf $O/trace
