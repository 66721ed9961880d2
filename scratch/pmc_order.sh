#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for o in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$o_$c
    rocprofv3 --pmc $c -d /tmp/pmc_${o}_$c --output-format csv -- python3 scratch/t_order.py $o > /dev/null 2>&1
    f=$(find /tmp/pmc_${o}_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $o $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'blur_tiled' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[3]]
v = [float(r['Counter_Value']) for r in rows]
print("order %s %s: %d launches, mean %.1f KiB-units" % (sys.argv[2], sys.argv[3], len(v), sum(v) / len(v)))
PY
  done
done
