"""Timeline of the blur step's single launch (diagnostic build: make HIPFLAGS_EXTRA=-DDIB_STEP_STAMPS): when the compacting
workgroups pass their phases and when the blur workgroups of grid row 0 start, see the counter and end (100 MHz stamps)."""
import ctypes, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from detectinblur_amd import _lib, blur_ops
from detectinblur_amd.models import blur_functions as BF
dev = torch.device("cuda", 0)
host = bench.make_psfs_host(0)
images, dicts, psfs, psfs_host, _ = bench.make_workload(0, dev, host)
l = _lib.lib()
l.dib_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
def step():
    batch = list(images); BF.blur_image_list(batch, dicts, psfs, psfs_complete=True); return batch
for _ in range(500): step()
buf = torch.zeros(8 * 32 + 4 * 1024, dtype=torch.int64, device=dev)
l.dib_debug_set_stamp_buffer(buf.data_ptr())
res = []
for rep in range(5):
    buf.zero_(); torch.cuda.synchronize()
    for _ in range(3): step()      # the last launch's stamps survive
    torch.cuda.synchronize()
    a = buf.cpu().numpy()
    comp = a[:8 * 8].reshape(8, 8); blur = a[256:256 + 4 * 832].reshape(832, 4)
    t0 = min(comp[:, 0].min(), blur[blur[:, 0] > 0, 0].min())
    c = (comp - t0) * 0.01     # us
    b = blur[blur[:, 0] > 0]
    res.append({"compact_us_rel_t0 (start, scans, staged, segments+sum+divide [record out], tables, -, drained, signalled)": np.round(c, 2).tolist(),
                "blur_row0_start_us": [round(float(x), 2) for x in np.percentile((b[:, 0] - t0) * 0.01, [0, 50, 100])],
                "blur_row0_ready_us": [round(float(x), 2) for x in np.percentile((b[:, 1] - t0) * 0.01, [0, 50, 100])],
                "blur_row0_end_us": [round(float(x), 2) for x in np.percentile((b[:, 2] - t0) * 0.01, [0, 50, 100])]})
l.dib_debug_set_stamp_buffer(None)
print(json.dumps(res[-1], indent=1))
