"""The blur step's single launch (in-launch tap compaction + hand-off) on a BUSY GPU -- where the reference calls the blur:
engine.py:101, inside a DDP rank whose detector kernels, RCCL and loader uploads share the device.  N steps each
  idle        nothing else on the device (control)
  gemm        back-to-back 4096^3 fp16 GEMMs on a second stream
  conv        a MIOpen 3x3 convolution loop (8 x 256 x 200 x 336 fp32) on a second stream
  rccl        a one-rank RCCL all-reduce loop of 166 MB (the detector's gradient size, reference train.py:238-241)
  process     a second PROCESS running the same blur steps on the same device
with a PSF set that changes every step (a stale table gives other pixels), the BASELINE batch and a ragged one alternating,
every 1,000th step compared bit for bit with compaction + blur as two launches.  Prints per scenario: steps, seconds, the
device status (0 = no hand-off ever gave up), whether the single launch is still in service, and -- on a -DDIB_STEP_POLLSTATS
build (scratch/build_variant.sh polls -DDIB_STEP_POLLSTATS; DIB_HIP_LIB=scratch/libdib_hip_polls.so) -- the most polls any
blur workgroup needed for its first segment / for the counter (waits of more than 32 polls only; the budget is 2^20).
    python scratch/t_step_contention.py [steps per scenario]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF

dev = torch.device("cuda", 0)
child = len(sys.argv) > 2 and sys.argv[1] == "--child"
n_steps = int(sys.argv[2] if child else (sys.argv[1] if len(sys.argv) > 1 else 200000))
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev)
ragged = [torch.rand(3, h, w, generator=torch.Generator().manual_seed(31 + i)).half().to(dev) for i, (h, w) in enumerate(bench.COCO_NATIVE_SIZES)]
rs = np.random.RandomState(5)
sets = []
for s in range(32):
    ps = []
    for k in range(8):
        a = np.zeros((128, 128), np.float64)
        n = int(rs.randint(3, 200)); sp = int(rs.choice([2, 6, 14, 30, 62]))
        a[np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127), np.clip(rs.randint(-sp, sp + 1, n) + 63, 0, 127)] = rs.random_sample(n) + 0.01
        ps.append(torch.from_numpy((a / a.sum()).astype(np.float16)).to(dev))
    sets.append(ps)
sets.append(list(psfs))
l = _lib.lib()
l.dib_debug_set_step_fused.argtypes = [ctypes.c_int]; l.dib_debug_set_step_fused.restype = None
l.dib_debug_step_single_launch.argtypes = [ctypes.c_int]; l.dib_debug_step_single_launch.restype = ctypes.c_int
have_stats = hasattr(l, "dib_debug_poll_stats")
if have_stats:
    l.dib_debug_poll_stats.argtypes = [ctypes.POINTER(ctypes.c_uint), ctypes.c_int]; l.dib_debug_poll_stats.restype = ctypes.c_int


def soak(name, top_up):
    """top_up(): called every few steps; keeps the contender's queue filled."""
    torch.cuda.synchronize()
    if have_stats:
        z = (ctypes.c_uint * 4)(); l.dib_debug_poll_stats(z, 1)
    t0 = time.time(); checked = 0
    for it in range(n_steps):
        if it % 8 == 0:
            top_up()
        ps = sets[it % len(sets)]
        batch = list(images if it % 2 == 0 else ragged)
        BF.blur_image_list(batch, dicts, ps, psfs_complete=True)
        if it % 1000 == 0:
            l.dib_debug_set_step_fused(0)
            ref = list(images if it % 2 == 0 else ragged)
            BF.blur_image_list(ref, dicts, ps, psfs_complete=True)
            l.dib_debug_set_step_fused(1)
            assert all(torch.equal(a, b) for a, b in zip(batch, ref)), (name, it)
            checked += 1
    torch.cuda.synchronize()
    st = (ctypes.c_uint * 4)()
    if have_stats:
        l.dib_debug_poll_stats(st, 1)
    print("%-8s %d steps in %.0f s, %d compared bit for bit with the two-launch path: no difference; device status %d, single launch in service: %d%s" % (
        name, n_steps, time.time() - t0, checked, l.dib_device_status(0), l.dib_debug_step_single_launch(-1),
        "; most polls for a first segment %d, for the counter %d (budget %d)" % (st[0], st[1], 1 << 20) if have_stats else ""), flush=True)


class Contender:
    """Keeps `depth` launches of fn() queued on a stream of its own."""
    def __init__(self, fn, depth=24):
        self.fn, self.depth, self.stream, self.ev, self.n = fn, depth, torch.cuda.Stream(), None, 0
    def __call__(self):
        if self.ev is None or self.ev.query():
            with torch.cuda.stream(self.stream):
                for _ in range(self.depth):
                    self.fn()
                self.ev = torch.cuda.Event(); self.ev.record(self.stream)
            self.n += self.depth


if child:
    soak("child", lambda: None)
    sys.exit(0)

soak("idle", lambda: None)
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
c = Contender(lambda: a @ a)
soak("gemm", c); print("         (%d GEMMs of 4096^3 ran beside them)" % c.n, flush=True)
x = torch.randn(8, 256, 200, 336, device=dev); w = torch.randn(256, 256, 3, 3, device=dev)
c = Contender(lambda: torch.nn.functional.conv2d(x, w, padding=1), depth=8)
soak("conv", c); print("         (%d convolutions ran beside them)" % c.n, flush=True)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
g = torch.randn(166 * 1024 * 1024 // 4, device=dev)
c = Contender(lambda: dist.all_reduce(g), depth=4)
soak("rccl", c); print("         (%d all-reduces of 166 MB ran beside them)" % c.n, flush=True)
torch.cuda.synchronize()
p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(n_steps)])
time.sleep(20)          # the child's import + workload set-up
soak("process", lambda: None)
print("         (second process exit code %d)" % p.wait(), flush=True)
dist.destroy_process_group()
