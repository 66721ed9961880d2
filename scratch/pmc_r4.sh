#!/bin/bash
# Round-4 evidence for the blur kernel: kernel stats of the bench command, then separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), warm and cold.   gpurun -- bash scratch/pmc_r4.sh ; python scratch/collect_r4.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r4; rm -rf $O; mkdir -p $O
# the unprofiled default bench on the SAME box first: headline and profile from one machine
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 bench.py --steps 50 --warmup 5 --repeats 20 --no-cpu-baseline --no-train-step --no-eval-sweep > $O/bench_under_rocprof.json 2> $O/trace.log
for mode in warm cold; do
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$mode --output-format csv -- python3 scratch/prof_blur_r2.py 24 $mode > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$mode --output-format csv -- python3 scratch/prof_blur_r2.py 24 $mode > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq --output-format csv -- python3 scratch/prof_blur_r2.py 12 warm > /dev/null 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $O/sq2 --output-format csv -- python3 scratch/prof_blur_r2.py 12 warm > /dev/null 2>&1
find $O -name "*.csv" | wc -l
