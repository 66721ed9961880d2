#!/bin/bash
# The train step's kernels of >= $1 us (default 80) outside MIOpen / hipBLASLt, in execution order, for the last traced step.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pbk
rocprofv3 --kernel-trace -d /tmp/pbk --output-format csv -- python3 scratch/train_only.py 2 > /dev/null 2>&1
python3 - "${1:-80}" <<'PY'
import csv, glob, sys
thr = float(sys.argv[1])
f = glob.glob("/tmp/pbk/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last SGD update but one: find optimizer kernels (multi_tensor / _foreach) -- simpler: last 1/6 of the trace
n = len(rows)
tail = rows[int(n * 5 / 6):]
conv = ("igemm", "Cijk", "miopen", "MIOpen", "conv", "Conv", "gemm", "Gemm")
tot = 0.0
for r in tail:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"]
    if us >= thr and not any(c in name for c in conv):
        tot += us
        print("%8.1f us  grid %10s  %s" % (us, r.get("Grid_Size_X", r.get("Grid_Size", "?")), name[:150]))
print("total %.2f ms in this window (%d kernels in window)" % (tot / 1e3, len(tail)))
PY
