#!/bin/bash
# TA / TCP / SQ counters of the blur kernel, one small group per pass (evidence for the window-fill question).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# (TA_* counter groups never finish on this pool: a pass with TA_BUSY_avr / TA_BUFFER_* / TA_*_STALLED_* ran
#  into its timeout twice, so they are not collected.)
for grp in "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"; do
  rm -rf /tmp/pmcf
  timeout 75 rocprofv3 --pmc $grp -d /tmp/pmcf --output-format csv -- python3 scratch/prof_blur.py 5 > /dev/null 2>&1 || echo "  pass timed out / failed: $grp"
  f=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'blur_tiled' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print("  (no data)", e)
for k, v in acc.items():
    print("%-40s %16.1f per launch" % (k, sum(v) / len(v)))
PY
done
