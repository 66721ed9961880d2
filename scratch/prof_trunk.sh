#!/bin/bash
# Kernel timeline of one graphed forward pass at b = 1 (scratch/t_trunk_trace.py) -> gpurun_out/r4_trunk_b1_trace.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/ptk
rocprofv3 --kernel-trace -d /tmp/ptk --output-format csv -- python3 scratch/t_trunk_trace.py > /dev/null 2> /tmp/ptk.err
grep forward /tmp/ptk.err | tail -3
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/ptk/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_pool_fwd" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
out = ["one graphed forward pass at b = 1, 3 x 800 x 1333 (stem to the next image's stem): %d kernels, span %.2f ms, busy %.2f ms" % (
    len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e6, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e6)]
out.append("%8s %8s %7s  %s" % ("start us", "dur us", "WGs", "kernel"))
for r in step:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:120]
    out.append("%8.1f %8.1f %7d  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid // max(wg, 1), name))
open("gpurun_out/r4_trunk_b1_trace.txt", "w").write("\n".join(out) + "\n")
print(out[0])
PY
