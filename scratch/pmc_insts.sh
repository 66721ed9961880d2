#!/bin/bash
# Instruction counts of the blur kernel by kind (one --pmc pass each; scratch/prof_blur_r2.py warm).  DIB_BLUR_SHAPE=0|1 selects the shape.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/insts_${DIB_BLUR_SHAPE:-0}; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $O/a --output-format csv -- python3 scratch/prof_blur_r2.py 8 warm > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/b --output-format csv -- python3 scratch/prof_blur_r2.py 8 warm > /dev/null 2>&1
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "blur_" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k][2:]
    print("%-22s %14.0f" % (k, sum(v) / len(v)))
PY
