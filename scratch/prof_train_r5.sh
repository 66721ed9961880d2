#!/bin/bash
# Detector train step under rocprofv3: per-kernel totals -> achieved fp32 TFLOP/s of the MIOpen convolutions.
#   gpurun -- bash scratch/prof_train_r5.sh      (writes gpurun_out/r5_train_step_conv.txt; copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 scratch/train_flops.py 2>/dev/null | tail -1 > gpurun_out/train_flops.json
rm -rf /tmp/ptc
rocprofv3 --kernel-trace --stats -d /tmp/ptc --output-format csv -- python3 scratch/train_only.py 12 > gpurun_out/train_prof_bench.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json
fl = json.load(open("gpurun_out/train_flops.json"))
f = glob.glob("/tmp/ptc/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
steps = 12 + 4          # train steps the profile covers (warm-up included: same kernels)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
is_conv = lambda n: any(t in n for t in ("igemm", "Igemm", "conv", "Conv", "Cijk", "gemm", "Gemm", "miopen", "MIOpen", "naive_", "Winograd", "winograd"))
conv_ns = sum(float(r["TotalDurationNs"]) for r in rows if is_conv(r["Name"]))
out = []
out.append("Detector train step (bench.py train_step_bench via scratch/train_only.py: b=8, 3x800x1333, fp32, channels-last), rocprofv3 --kernel-trace --stats, %d steps (12 timed + 4 warm-up)" % steps)
out.append("FLOPs per step (torch.utils.flop_counter on the real step, forward + backward): total %.3f T, convolutions %.3f T, matmuls %.3f T"
           % (fl["total_flops_per_step"] / 1e12, fl["conv_flops_per_step"] / 1e12, fl["matmul_flops_per_step"] / 1e12))
out.append("GPU time per step, all kernels: %.2f ms; convolution / GEMM kernels: %.2f ms (%.0f %%)" % (tot / steps / 1e6, conv_ns / steps / 1e6, 100 * conv_ns / tot))
ach = (fl["conv_flops_per_step"] + fl["matmul_flops_per_step"]) / (conv_ns / steps * 1e-9) / 1e12
out.append("achieved on those kernels: %.1f TFLOP/s fp32 = %.0f %% of the 157.3 TFLOP/s fp32 MFMA peak" % (ach, 100 * ach / 157.3))
out.append("")
out.append("%-110s %8s %10s %7s" % ("kernel (top 90 by time)", "calls", "ms/step", "share"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:90]:
    out.append("%-110s %8s %10.3f %6.1f%%" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / steps / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
out.append("")
out.append("kernel launches per step: %.0f; non-convolution kernels: %.2f ms per step over %.0f launches" % (
    sum(int(r["Calls"]) for r in rows) / steps, (tot - conv_ns) / steps / 1e6, sum(int(r["Calls"]) for r in rows if not is_conv(r["Name"])) / steps))
import shutil
tr = glob.glob("/tmp/ptc/**/*kernel_trace.csv", recursive=True)
if tr:
    shutil.copy(tr[0], "gpurun_out/r5_train_kernel_trace.csv")
# ---- is the side-stream tap compaction on the step's critical path?  For every blur launch: when did the compaction of ITS batch
# end, when did the main stream's previous kernel end, when did the blur start.  The blur waits for both; the compaction delays
# the step only where it ends AFTER the main stream's previous kernel.
if tr:
    rowsT = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id") or r.get("Stream_Id") or "") for r in rowsT]
    blur_i = [i for i, e in enumerate(ev) if "blur_quad_f16_kernel" in e[2] or "blur_step_f16_kernel" in e[2]]
    out.append("")
    out.append("side-stream tap compaction vs the blur that consumes its tables (per train step; us):")
    out.append("%6s %14s %14s %14s %12s" % ("step", "compact dur", "compact end->blur", "prev main end->blur", "delays step?"))
    late = 0
    for n, bi in enumerate(blur_i):
        b0 = ev[bi][0]
        comp = [e for e in ev[:bi] if "psf_compact" in e[2]]
        if not comp:
            continue
        c = comp[-1]
        prev_main = [e for e in ev[:bi] if e[3] == ev[bi][3] and e[1] <= b0]
        pm_end = prev_main[-1][1] if prev_main else 0
        delays = c[1] > pm_end
        late += delays
        out.append("%6d %14.1f %14.1f %14.1f %12s" % (n, (c[1] - c[0]) / 1e3, (b0 - c[1]) / 1e3, (b0 - pm_end) / 1e3, "YES" if delays else "no"))
    out.append("compactions that ended after the main stream's previous kernel (i.e. could have delayed the blur): %d of %d" % (late, len(blur_i)))
open("gpurun_out/r5_train_step_conv.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
