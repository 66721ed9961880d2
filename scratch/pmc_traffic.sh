#!/bin/bash
# HBM traffic of the blur kernel: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1/trace --output-format csv -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train-step > gpurun_out/prof_r1/bench_under_rocprof.json 2> gpurun_out/prof_r1/trace.log
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_r1/fetch --output-format csv -- python scratch/prof_blur.py 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_r1/write --output-format csv -- python scratch/prof_blur.py 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/prof_r1/sq --output-format csv -- python scratch/prof_blur.py 5 > /dev/null 2>&1
find gpurun_out/prof_r1 -name "*.csv" | head -20
