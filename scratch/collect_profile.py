"""Builds profiles/r1_blur_pmc.json + copies the kernel-stats CSV from gpurun_out/prof_r1 (written by
scratch/pmc_traffic.sh on the GPU box)."""
import csv, glob, json, os, shutil, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_r1")
def counters(sub):
    out = {}
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "blur_tiled" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    return out
per = {}
for sub in ("fetch", "write", "sq"):
    per.update(counters(sub))
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
blur = [r for r in rows if "blur_tiled" in r["Name"]][0]
comp = [r for r in rows if "psf_compact" in r["Name"]][0]
shutil.copy(stats, os.path.join(root, "profiles", "r1_bench_kernel_stats.csv"))
bench_line = open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1]
open(os.path.join(root, "profiles", "r1_bench_under_rocprof.json"), "w").write(bench_line + "\n")
algo = 102374400
traffic = per["FETCH_SIZE"] * 1024 * 2.0 + per["WRITE_SIZE"] * 1024 * 1.0
old = json.load(open(os.path.join(root, "profiles", "r1_blur_pmc.json")))
doc = {
    "command": old["command"],
    "workload": old["workload"],
    "kernel": "dib::blur_tiled_f16_kernel<4,1> (XCD-band tile order)",
    "per_launch": {k: per[k] for k in sorted(per)},
    "kernel_avg_ns": float(blur["AverageNs"]), "kernel_calls": int(blur["Calls"]),
    "compact_avg_ns": float(comp["AverageNs"]),
    "calibration": old["calibration"],
    "hbm_traffic_bytes_per_launch": traffic,
    "algorithmic_bytes_per_launch": algo,
    "traffic_over_algorithmic": traffic / algo,
    "history": {"flat tile order (before XCD bands)": {"FETCH_SIZE": 65552.6, "WRITE_SIZE": 53362.5,
                "hbm_traffic_bytes_per_launch": 188895001.6, "traffic_over_algorithmic": 1.845, "kernel_avg_ns": 57794.3}},
}
json.dump(doc, open(os.path.join(root, "profiles", "r1_blur_pmc.json"), "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("kernel_avg_ns", "compact_avg_ns", "hbm_traffic_bytes_per_launch", "traffic_over_algorithmic")}))
