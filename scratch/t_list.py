import sys, time
sys.path.insert(0, '.')
import torch, bench
from detectinblur_amd import blur_ops
from detectinblur_amd.models import blur_functions as BF
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, torch.device("cuda", 0))
for name, fn in (("compact list", lambda: blur_ops.compact_psfs(psfs, True)), ("compact stack", lambda: blur_ops.compact_psfs(torch.stack(psfs), True)),
                 ("blur_image_list", lambda: (blur_ops.invalidate_cache(), BF.blur_image_list(list(images), dicts, psfs)))):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-16s host %.1f us/call, with sync %.1f us/call" % (name, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
