#!/bin/bash
# usage: scratch/prof_train.sh <variant> ; writes gpurun_out/train_tail_<variant>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_t
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python3 scratch/t_train.py $1 > gpurun_out/prof_t.log 2>&1
tail -2 gpurun_out/prof_t.log
python3 scratch/trace_tail.py /tmp/prof_t/t_kernel_trace.csv 320 > gpurun_out/train_tail_$1.txt
head -40 gpurun_out/train_tail_$1.txt
