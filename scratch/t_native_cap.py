"""Native-size ragged batch: blur kernel time against the dynamic LDS size (= the cap on workgroups per CU)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np, torch, bench
    from detectinblur_amd import blur_ops
    dev = torch.device("cuda", 0)
    images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
    tables = blur_ops.compact_psfs(psfs, normalize=True)
    idx = sorted(range(8), key=lambda k: -dicts[k]["psf_taps"])
    native = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
    out = {}
    for name, imgs in (("native", native), ("baseline", images)):
        ordered = [imgs[k] for k in idx]
        for _ in range(30): blur_ops.sparse_blur(list(ordered), idx, tables)
        out[name] = round(1e3 * sorted(bench.kernel_time_ms(lambda k: blur_ops.sparse_blur(list(ordered), idx, tables), 200) for _ in range(7))[3], 2)
    print(json.dumps(out))
else:
    for pad in (0, 768, 1024, 3072, 3584, 7000, 7500, 13000):
        env = dict(os.environ, DIB_EXP_LDS_PAD=str(pad))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print("LDS %d B (%d per CU by LDS): %s" % (19712 + pad, 163840 // (19712 + pad), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]))
