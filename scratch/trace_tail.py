"""Aggregate the last `--ms` milliseconds of a rocprofv3 kernel_trace.csv by kernel name."""
import csv, sys, collections
path, win_ms = sys.argv[1], float(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
end = rows[-1][1]
lo = end - int(win_ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows:
    if s >= lo:
        agg[n][0] += 1; agg[n][1] += e - s; busy += e - s
print("window %.1f ms, kernel-busy %.1f ms, %d launches" % (win_ms, busy / 1e6, sum(v[0] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%6.2f%% %6d calls %9.1f us avg %9.2f ms  %s" % (100.0 * t / busy, c, t / c / 1e3, t / 1e6, n[:120] if "elementwise" not in n else n[:420]))

# ---- idle gaps: where the GPU waits for the host ---------------------------------------------------
win = [(s, e, n) for s, e, n in rows if s >= lo]
gaps = collections.defaultdict(lambda: [0, 0])
idle = 0
cur_end = win[0][1]
prev = win[0][2]
for s, e, n in win[1:]:
    if s > cur_end:
        g = s - cur_end
        idle += g
        if g > 20000:
            key = (prev[:60], n[:60])
            gaps[key][0] += 1; gaps[key][1] += g
    if e > cur_end:
        cur_end = e; prev = n
print("\nidle %.1f ms in window; gaps > 20 us by (kernel before -> kernel after):" % (idle / 1e6))
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.2f ms %5d x  %s  ->  %s" % (t / 1e6, c, a, b))

# ---- per-step accounting: one blur_tiled launch marks the start of each step -----------------------------
marks = [s for s, e, n in rows if "blur_tiled" in n]
if len(marks) >= 4:
    print("\nper-step (delimited by blur launches): span / kernel-busy / idle / launches")
    for a, b in zip(marks[-4:-1], marks[-3:]):
        ks = [(s, e) for s, e, n in rows if a <= s < b]
        busy, cur = 0, a
        for s, e in ks:
            s2 = max(s, cur)
            if e > s2: busy += e - s2; cur = e
        print("  %.1f ms  busy %.1f ms  idle %.1f ms  %d launches" % ((b - a) / 1e6, busy / 1e6, (b - a - busy) / 1e6, len(ks)))
