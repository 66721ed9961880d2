#!/bin/bash
# What MIOpen's SubTensorOpWithScalar1d launches of the train step are: duration histogram + grid sizes from a kernel trace.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pts
rocprofv3 --kernel-trace -d /tmp/pts --output-format csv -- python3 scratch/train_only.py 4 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pts/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "SubTensorOpWithScalar" in r["Kernel_Name"]]
steps = 8
d = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))) for r in rows)
print("calls per step %.0f, total %.2f ms per step" % (len(d) / steps, sum(x[0] for x in d) / steps / 1e3))
buckets = collections.Counter()
tot = collections.Counter()
for us, g in d:
    b = "<5us" if us < 5 else "<20us" if us < 20 else "<100us" if us < 100 else ">=100us"
    buckets[b] += 1; tot[b] += us
for b in ("<5us", "<20us", "<100us", ">=100us"):
    print("%8s: %5.1f calls/step, %.3f ms/step" % (b, buckets[b] / steps, tot[b] / steps / 1e3))
print("largest:", [(round(us, 1), g) for us, g in d[-12:]])
PY
