"""A/B of builds / settings of the library on the native-size ragged batch and on the BASELINE batch (blur kernel alone): eager
loop of 200 calls (event-timed: HOST-bound for the ragged batch, ~20 us of interpreter per call) and the same launch replayed
from a HIP graph of 20 (device time, inter-kernel gaps included):
    python scratch/t_native_ab.py default default@DIB_FLAT_GRID=0 scratch/libdib_hip_x.so ...   (own child process each, alternating, 3 rounds)"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("T_NATIVE_CHILD"):
    import numpy as np, torch, bench
    from detectinblur_amd import blur_ops, _lib
    MODE = {"bitexact": _lib.DIB_ACC_BITEXACT, "fma16": _lib.DIB_ACC_FMA16, "fast16": 3}[os.environ.get("DIB_AB_MODE", "bitexact")]
    dev = torch.device("cuda", 0)
    images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
    tables = blur_ops.compact_psfs(psfs, normalize=True, vruns=True)
    idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
    native = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
    out = {}
    ref = {}
    for name, imgs in (("native", native), ("baseline", images)):
        ordered = [imgs[k] for k in idx]
        for _ in range(30): r = blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
        out[name] = round(1e3 * sorted(bench.kernel_time_ms(lambda k: blur_ops.sparse_blur(list(ordered), idx, tables, MODE), 200) for _ in range(7))[3], 2)
        out[name + "_sum"] = float(sum(x.float().sum().item() for x in r))
        # device-side time: 20 launches captured in a HIP graph, replayed (no host in between)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(20): keep = blur_ops.sparse_blur(list(ordered), idx, tables, MODE)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): g.replay()
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 200 * 1e3)
        out[name + "_graph"] = round(sorted(ts)[3], 2)
    print(json.dumps(out))
else:
    libs = sys.argv[1:]
    for rnd in range(3):
        for lib in libs:
            env = dict(os.environ, T_NATIVE_CHILD="1")
            tag = lib
            if "@" in lib:
                lib, *kvs = lib.split("@")
                for kv in kvs:
                    env[kv.split("=")[0]] = kv.split("=")[1]
            if lib != "default": env["DIB_HIP_LIB"] = os.path.abspath(lib)
            r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
            print("%-40s %s" % (tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
