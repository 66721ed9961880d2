#!/bin/bash
# GPU time per image of the ensemble evaluation loop (scratch/t_eval_prof.py: 6 warm-up + 24 images) under rocprofv3.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pev
rocprofv3 --kernel-trace --stats -d /tmp/pev --output-format csv -- python3 scratch/t_eval_prof.py > /dev/null 2> /tmp/pev.err
grep "per image" /tmp/pev.err
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pev/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("GPU kernel time per image (30 images incl. warm-up and graph captures): %.2f ms, %d launches per image" % (tot / n / 1e6, sum(int(r["Calls"]) for r in rows) / n))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-100s %6s %8.3f ms/img %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / n / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
