#!/bin/bash
# Round-5 evidence: the unprofiled default bench, kernel stats of the bench command, separate --pmc passes for the step's single
# launch (FETCH_SIZE and WRITE_SIZE do not fit one pass), warm and cold, SQ counters, and the in-launch timeline of the hand-off.
#   gpurun -- bash scratch/pmc_r5.sh ; python scratch/collect_r5.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r5; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 bench.py --steps 50 --warmup 5 --repeats 20 --no-cpu-baseline --no-train-step --no-eval-sweep > $O/bench_under_rocprof.json 2> $O/trace.log
for mode in warm cold; do
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$mode --output-format csv -- python3 scratch/prof_step_r5.py 24 $mode > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$mode --output-format csv -- python3 scratch/prof_step_r5.py 24 $mode > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq --output-format csv -- python3 scratch/prof_step_r5.py 12 warm > /dev/null 2>&1
DIB_HIP_LIB=$PWD/scratch/libdib_hip_stamps.so python3 scratch/t_step_stamps.py > $O/step_timeline.json 2>&1
python3 scratch/t_step_fused.py > $O/step_ab.json 2>&1
python3 scratch/t_modes.py > $O/modes.json 2>&1
find $O -name "*.csv" | wc -l
