#!/bin/bash
# SQ counters of the blur kernel by accumulation mode (separate --pmc passes):  gpurun -- bash scratch/pmc_modes_r6.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/pmc_modes; rm -rf $O; mkdir -p $O
for mode in fast16 fma16 bitexact; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $O/a_$mode --output-format csv -- python3 scratch/prof_modes_r6.py $mode > /dev/null 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $O/b_$mode --output-format csv -- python3 scratch/prof_modes_r6.py $mode > /dev/null 2>&1
  rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH -d $O/c_$mode --output-format csv -- python3 scratch/prof_modes_r6.py $mode > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r6/pmc_modes"
for mode in ("fast16", "fma16", "bitexact"):
    tot = collections.defaultdict(list)
    for f in glob.glob("%s/?_%s/**/*counter_collection.csv" % (O, mode), recursive=True):
        for row in csv.DictReader(open(f)):
            if "blur_quad_f16_kernel" in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(mode, {k: round(sum(v) / len(v)) for k, v in sorted(tot.items())}, "launches", {k: len(v) for k, v in tot.items()}.get("SQ_WAVES"))
PY
