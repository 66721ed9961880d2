#!/bin/bash
# Duration histogram + grid sizes of the train step's kernels whose name contains $1 (kernel trace of 4 + 4 steps).
#   gpurun -- bash scratch/prof_kernel_hist.sh direct_copy CUDAFunctor_add
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pkh
rocprofv3 --kernel-trace -d /tmp/pkh --output-format csv -- python3 scratch/train_only.py 4 > /dev/null 2>&1
python3 - "$@" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pkh/**/*kernel_trace.csv", recursive=True)[0]
allrows = list(csv.DictReader(open(f)))
steps = 8
for pat in sys.argv[1:]:
    rows = [r for r in allrows if pat in r["Kernel_Name"]]
    d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", 0))) for r in rows)
    print("== %s: calls per step %.0f, total %.2f ms per step" % (pat, len(d) / steps, sum(x[0] for x in d) / steps / 1e3))
    by = collections.defaultdict(lambda: [0, 0.0])
    for us, g, w in d:
        by[g][0] += 1; by[g][1] += us
    for g, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   grid %10d: %5.1f calls/step, %7.1f us each, %.3f ms/step" % (g, n / steps, t / n, t / steps / 1e3))
PY
