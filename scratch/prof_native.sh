#!/bin/bash
# rocprofv3 kernel durations of the native-size ragged batch's blur: flat 1-D grid (default) vs the 2-D grid
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for flat in 1 0; do
  export DIB_FLAT_GRID=$flat T_NATIVE_CHILD=1
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_native_$flat --output-format csv -- python3 $R/scratch/t_native_ab.py > $R/gpurun_out/prof_native_$flat.log 2>&1
  f=$(find $R/gpurun_out/prof_native_$flat -name '*kernel_stats.csv' | head -1)
  echo "== DIB_FLAT_GRID=$flat"; grep -E "Name|blur_quad" "$f" | cut -c1-260
done
