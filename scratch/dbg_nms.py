import sys
sys.path.insert(0, '.')
import numpy as np, torch
from detectinblur_amd.models import detector_ops as ops
for n in (65, 128, 129, 200, 700, 1025, 3000, 4097, 9000):
    rs = np.random.RandomState(n)
    c = rs.uniform(0, 400, (n, 2)); s = rs.uniform(5, 80, (n, 2))
    boxes = torch.tensor(np.concatenate([c - s / 2, c + s / 2], 1), dtype=torch.float32)
    scores = torch.tensor(rs.permutation(n) / float(n), dtype=torch.float32)
    want = ops.nms(boxes, scores, 0.5).tolist()
    got = ops.nms(boxes.cuda(), scores.cuda(), 0.5).cpu().tolist()
    if got == want:
        print(n, "ok", len(want)); continue
    order = scores.argsort(descending=True).tolist()
    rank = {b: i for i, b in enumerate(order)}
    wr, gr = sorted(rank[b] for b in want), sorted(rank[b] for b in got)
    print(n, "MISMATCH len want %d got %d" % (len(wr), len(gr)), "extra ranks", sorted(set(gr) - set(wr))[:10], "missing ranks", sorted(set(wr) - set(gr))[:10])
    print("  got order sorted by rank?", gr == [rank[b] for b in got])
