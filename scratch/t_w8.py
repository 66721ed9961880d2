import sys, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import dib_oracle as O
from detectinblur_amd import blur_ops, _lib
l = _lib.lib(); l.dib_debug_set_waves.argtypes = [ctypes.c_int]
l.dib_debug_set_waves(int(sys.argv[1]))
rs = np.random.RandomState(5)
for shape in [(1, 70, 100), (3, 97, 301), (2, 130, 257), (1, 64 + 1, 600), (3, 200, 640)]:
    img = rs.random_sample(shape).astype(np.float16)
    a = np.zeros((128, 128)); n = 20
    a[rs.randint(55, 72, n), rs.randint(55, 72, n)] = rs.random_sample(n) + 0.01
    psf = O.to_half_like_torch(a * 0.2)
    want = [img.copy()]; O.blur_image_list(want, [{"blurring": True}], [psf])
    tables = blur_ops.compact_psfs(torch.from_numpy(psf[None]).cuda(), normalize=True)
    out = blur_ops.sparse_blur([torch.from_numpy(img).cuda()], [0], tables)[0]
    torch.cuda.synchronize()
    ok = np.array_equal(out.cpu().numpy().view(np.uint16).squeeze(), want[0].view(np.uint16).squeeze())
    print(shape, "ok" if ok else "MISMATCH", flush=True)
