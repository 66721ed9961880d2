"""Cost of pinning host memory while the GPU is busy: time of tensor.pin_memory() (fresh block / cached block) and of the
H2D copy out of it, with an idle GPU and with a stream kept busy by long kernels."""
import sys, threading, time
import torch
dev = torch.device("cuda", 0)
x = torch.rand(8192, 8192, device=dev)
src = [torch.rand(3, 800, 1333) for _ in range(8)]


def busy(stop):
    while not stop.is_set():
        for _ in range(20):
            torch.mm(x, x)            # ~ms-scale kernels back to back
        torch.cuda.current_stream().synchronize()


def probe(tag):
    ts = []
    for s in src:
        t0 = time.perf_counter(); p = s.pin_memory(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, "pin 8 x 12.8 MB (fresh): " + " ".join("%.1f" % t for t in ts), flush=True)
    keep = [s.pin_memory() for s in src]
    del keep
    ts = []
    for s in src:
        t0 = time.perf_counter(); p = s.pin_memory(); ts.append((time.perf_counter() - t0) * 1e3)
    print(tag, "pin 8 x 12.8 MB (cached blocks): " + " ".join("%.1f" % t for t in ts), flush=True)
    side = torch.cuda.Stream()
    p = [s.pin_memory() for s in src]
    with torch.cuda.stream(side):
        t0 = time.perf_counter()
        d = [q.to(dev, non_blocking=True) for q in p]
        side.synchronize()
        print(tag, "H2D of 8 pinned tensors on a side stream: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


probe("idle GPU |")
torch.cuda.empty_cache()
torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
stop = threading.Event()
th = threading.Thread(target=busy, args=(stop,)); th.start()
time.sleep(0.5)
src = [torch.rand(3, 801, 1333) for _ in range(8)]     # another size class? (same rounded bucket) -- fresh tensors anyway
probe("busy GPU |")
src = [torch.rand(3, 1200, 1333) for _ in range(8)]    # a size that needs new blocks
probe("busy GPU, new size |")
stop.set(); th.join()
