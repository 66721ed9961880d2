"""MIOpen find-db entries for the batch-1 inference shapes (the shipped user db covered the b = 8 training shapes only):
time the detector's static trunk and the estimator at b = 1 with the immediate-mode choice, then with
torch.backends.cudnn.benchmark = True (MIOpen's Find: measured choice, written to MIOPEN_USER_DB_PATH = detectinblur_amd/miopen_db),
and copy the updated database to gpurun_out/."""
import glob, os, shutil, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch import nn
import detectinblur_amd  # noqa: F401  (sets MIOPEN_USER_DB_PATH)
from detectinblur_amd.models.blur_estimator import resnet18
from detectinblur_amd.models.faster_rcnn import fasterrcnn_resnet50_fpn
print("MIOPEN_USER_DB_PATH =", os.environ.get("MIOPEN_USER_DB_PATH"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = fasterrcnn_resnet50_fpn(num_classes=91, pretrained=False, pretrained_backbone=False).to(dev).eval()
est = resnet18(); est.fc = nn.Linear(512, 4); est = est.to(dev).eval()
x = torch.rand(1, 3, 800, 1344, device=dev).contiguous(memory_format=torch.channels_last)
xe = torch.rand(1, 3, 800, 1312, device=dev)
m._sizes = {1: torch.tensor([[1333.0, 800.0]], device=dev)}


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    print("immediate mode: trunk %.2f ms, estimator %.2f ms" % (timeit(lambda: m._trunk(x)), timeit(lambda: est(xe))))
    torch.backends.cudnn.benchmark = True
    print("after Find    : trunk %.2f ms, estimator %.2f ms" % (timeit(lambda: m._trunk(x)), timeit(lambda: est(xe))))
os.makedirs("gpurun_out/miopen_db", exist_ok=True)
for f in glob.glob(os.path.join(os.environ["MIOPEN_USER_DB_PATH"], "*")):
    shutil.copy(f, "gpurun_out/miopen_db/")
    print("copied", f, os.path.getsize(f))
