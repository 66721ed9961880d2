"""Builds profiles/r5_blur_pmc.json, r5_bench_kernel_stats.csv, r5_bench_under_rocprof.json, r5_bench_default.json,
r5_step_timeline.txt from gpurun_out/prof_r5 (written by scratch/pmc_r5.sh on the GPU box)."""
import csv, glob, json, os, shutil
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_r5")
def counters(sub, skip=6):
    out = {}
    for f in sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        acc = {}
        for r in csv.DictReader(open(f)):
            if "blur_step_f16_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[skip:] if len(v) > skip else v
            out[k] = sum(v) / len(v)
    return out
warm, cold = {}, {}
for sub in ("fetch_warm", "write_warm", "sq"):
    warm.update(counters(sub, 2))
for sub in ("fetch_cold", "write_cold"):
    cold.update(counters(sub, 12))
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(stats)))
step = [r for r in rows if "blur_step_f16_kernel<0>" in r["Name"]][0]
blur = [r for r in rows if "blur_quad_f16_kernel<0, 128, false>" in r["Name"]][0]
trace = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
durs = {"step": [], "blur": []}
tr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(trace)))      # EVERY kernel: a run ends at any other launch
# maximal runs of back-to-back launches of ONE kernel: the roofline loops of bench.py are the longest runs of each
def longest_run(name):
    best, cur = [], []
    for _, d, n in tr:
        if name in n:
            cur.append(d)
        else:
            best, cur = (cur if len(cur) > len(best) else best), []
    return cur if len(cur) > len(best) else best
loop_step, loop_blur = longest_run("blur_step_f16_kernel<0>"), longest_run("blur_quad_f16_kernel<0, 128, false>")
loop_blur = loop_blur[:1040] if len(loop_blur) > 1040 else loop_blur        # warm roofline loop; the cold rotation follows it
shutil.copy(stats, os.path.join(root, "profiles", "r5_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines() if l.startswith("{")][-1]
open(os.path.join(root, "profiles", "r5_bench_under_rocprof.json"), "w").write(line + "\n")
default_line = [l for l in open(os.path.join(src, "bench_default.json")).read().strip().splitlines() if l.startswith("{")][-1]
open(os.path.join(root, "profiles", "r5_bench_default.json"), "w").write(default_line + "\n")
default = json.loads(default_line)
algo = 102374400 + 8 * 32768
tw = warm["FETCH_SIZE"] * 1024 * 2.0 + warm["WRITE_SIZE"] * 1024
tc = cold["FETCH_SIZE"] * 1024 * 2.0 + cold["WRITE_SIZE"] * 1024
doc = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --repeats 20 --no-cpu-baseline --no-train-step --no-eval-sweep (kernel stats in r5_bench_kernel_stats.csv); PMC: separate rocprofv3 --pmc passes over scratch/prof_step_r5.py (same workload through blur_image_list; warm = one resident batch, cold = 6 input batches + 6 live output blocks round-robin, 614 MB)",
    "workload": "configs[1]: batch 8 x 3x800x1333 fp16, 8 PSFs (expl 0.005, low exposure), taps per PSF [43,53,28,26,56,51,19,26]",
    "kernel": "dib::blur_step_f16_kernel<0> (the step's single launch: 8 compacting workgroups + 6,600 blur workgroups of the default 128 x 32 tiles, bit-exact mode)",
    "per_launch_warm": {k: warm[k] for k in sorted(warm)},
    "per_launch_cold": {k: cold[k] for k in sorted(cold)},
    "kernel_avg_ns": float(step["AverageNs"]), "kernel_calls": int(step["Calls"]),
    "kernel_avg_ns_roofline_loop": sum(loop_step) / len(loop_step), "roofline_loop_calls": len(loop_step),
    "blur_only": {"kernel": "dib::blur_quad_f16_kernel<0, 128, false>", "kernel_avg_ns": float(blur["AverageNs"]), "kernel_calls": int(blur["Calls"]),
                  "kernel_avg_ns_roofline_loop": sum(loop_blur) / len(loop_blur), "roofline_loop_calls": len(loop_blur)},
    "unprofiled_same_box": {"kernel_ms": default["roofline"]["kernel_ms"], "blur_only_kernel_ms": default["roofline"]["blur_only"]["kernel_ms"],
                            "ms_per_step": default["ms_per_step"], "value": default["value"],
                            "note": "python bench.py (no profiler) run by the same gpurun call on the same box right before the profiled passes: profiles/r5_bench_default.json"},
    "calibration": {"note": "scratch/ubench/ub_fetch.hip (round 1): 1 GiB read with 2-byte per-lane loads reports FETCH_SIZE = 524,293 KiB (exactly 1/2, as MI355X_MICROARCH.md states); 1 GiB of 2-byte stores reports WRITE_SIZE = 1,048,576 KiB (exact)",
                    "fetch_correction": 2.0, "write_correction": 1.0},
    "hbm_traffic_bytes_per_launch": tw, "hbm_traffic_bytes_per_launch_cold": tc,
    "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic": tw / algo, "traffic_over_algorithmic_cold": tc / algo,
}
json.dump(doc, open(os.path.join(root, "profiles", "r5_blur_pmc.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", "r5_step_timeline.txt"), "w") as f:
    f.write("In-launch timeline of the blur step's single launch (blur_step_f16_kernel<0>, BASELINE batch), 100 MHz wall-clock stamps of a\n"
            "diagnostic build (-DDIB_STEP_STAMPS: scratch/t_step_stamps.py), microseconds from the launch's first stamp.\n"
            "compact_us_rel_t0: per compacting workgroup (one per PSF): start | scans done (PSF loaded, non-zeros counted) | raw taps staged |\n"
            "  segments cut + sum formed + weights divided (the first segment's record goes out here) | offsets + taps + header stored | - |\n"
            "  stores drained | counter signalled.   blur_row0_*: min / median / max over the blur workgroups of grid row 0.\n\n")
    f.write(open(os.path.join(src, "step_timeline.json")).read())
    f.write("\n\nSame box, eager steps (median of 30 blocks of 100 steps; scratch/t_step_fused.py): one launch (fused) against compaction + blur as two launches\n")
    f.write(open(os.path.join(src, "step_ab.json")).read().strip().splitlines()[-1] + "\n")
    f.write("\nAccumulation modes, blur alone on compacted tables (scratch/t_modes.py)\n")
    f.write(open(os.path.join(src, "modes.json")).read().strip().splitlines()[-1] + "\n")
print(json.dumps({k: doc[k] for k in ("kernel_avg_ns", "kernel_avg_ns_roofline_loop", "roofline_loop_calls", "hbm_traffic_bytes_per_launch", "traffic_over_algorithmic", "hbm_traffic_bytes_per_launch_cold", "traffic_over_algorithmic_cold")}), doc["blur_only"])
