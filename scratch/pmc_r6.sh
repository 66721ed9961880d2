#!/bin/bash
# Round-6 evidence, ONE gpurun call on one box: the unprofiled default bench, kernel stats of the bench command, separate --pmc passes for
# the step's single launch (FETCH_SIZE and WRITE_SIZE do not fit one pass; warm and cold) and for the tolerance mode's kernel, SQ counters
# of the three accumulation modes, the tolerance mode's per-workgroup timeline, the native-size grid orders, the vertical-run micro-benchmark.
#   gpurun --timeout 1500 -- bash scratch/pmc_r6.sh ; python scratch/collect_r6.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 bench.py --steps 50 --warmup 5 --repeats 20 --no-cpu-baseline --no-train-step --no-eval-sweep --no-polling-side-run > $O/bench_under_rocprof.json 2> $O/trace.log
for mode in warm cold; do
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$mode --output-format csv -- python3 scratch/prof_step_r5.py 24 $mode > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$mode --output-format csv -- python3 scratch/prof_step_r5.py 24 $mode > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE -d $O/fetch_fast16 --output-format csv -- python3 scratch/prof_modes_r6.py fast16 24 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write_fast16 --output-format csv -- python3 scratch/prof_modes_r6.py fast16 24 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_fast16 --output-format csv -- python3 scratch/prof_modes_r6.py fast16 400 > /dev/null 2>&1
bash scratch/pmc_modes_r6.sh > $O/pmc_modes.txt 2>&1
DIB_HIP_LIB=$PWD/scratch/libdib_hip_tl.so DIB_TL_MODE=fast16 python3 scratch/timeline_native.py > $O/tl_fast16.txt 2>&1
python3 scratch/t_native_masks.py > $O/native_masks.txt 2>&1
python3 scratch/t_native_masks.py fma16 >> $O/native_masks.txt 2>&1
for m in fast16 fma16 bitexact; do DIB_AB_MODE=$m python3 scratch/t_native_ab.py default 2>&1 | tail -3 | sed "s/^default /$m /" >> $O/modes_graph.txt; done
scratch/ubench/ub_vrun > $O/ub_vrun.txt 2>&1
find $O -name "*.csv" | wc -l
