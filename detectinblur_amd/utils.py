"""Drop-in for the hot-path parts of the reference's utils.py.

  * expand_targets / fix_bounding_box_squeeze   reference utils.py:360-434  (HIP, one launch/image)
  * get_norm_params                              reference utils.py:219-273  (host tables)
  * collate_fn, distributed helpers, meters      reference utils.py:474-785  (below)
"""
import numpy as np
import os

import torch

from . import blur_ops

# ---------------------------------------------------------------------------------------------
# boxes
# ---------------------------------------------------------------------------------------------


def expand_targets(targets_GPU, blur_dicts, psfs_GPU, images_GPU, tables=None):
    """Grows every ground-truth box by the PSF's non-zero extent, then clamps (in place on
    target["boxes"]; returns the same list).  Reference utils.py:360-392.
    `tables` (beyond the reference's signature): the tap tables of exactly the blurring PSFs, in order, as handed to
    `blur_image_list(..., tables=)`; without it the extents are recomputed from `psfs_GPU`, as the reference does."""
    idx = [i for i, bd in enumerate(blur_dicts) if bd["blurring"]]
    if not idx:
        return targets_GPU
    for i in idx:
        # "This function is not flexible on purpose" (utils.py:366-370)
        if psfs_GPU[i].shape[0] != 128:
            raise Exception("Trying to expand with filters that are not 128 wide!")
    if tables is not None:
        if tables.count != len(idx) or tables.K != 128:
            raise ValueError("tables do not belong to the blurring PSFs of this batch")
    else:
        psfs = []
        for i in idx:
            p = psfs_GPU[i]
            psfs.append(p if p.dtype in (torch.float16, torch.float32) else p.float())
        if len({p.dtype for p in psfs}) != 1:
            psfs = [p.float() for p in psfs]
        tables = blur_ops.compact_psfs(psfs, normalize=True)
    for k, i in enumerate(idx):
        boxes = targets_GPU[i]["boxes"]
        shape = images_GPU[i].shape
        work = boxes if (boxes.dtype == torch.float32 and boxes.is_contiguous()) else boxes.float().contiguous()
        blur_ops.expand_boxes(work, tables, k, int(shape[1]), int(shape[2]))
        if work is not boxes:
            boxes.copy_(work)
    return targets_GPU


def fix_bounding_box_squeeze(target, image_shape):
    """Clamp to the image, open up degenerate boxes by one pixel each way, clamp again.
    Reference utils.py:395-434.  image_shape is (C, H, W)."""
    boxes = target["boxes"]
    work = boxes if (boxes.dtype == torch.float32 and boxes.is_contiguous()) else boxes.float().contiguous()
    blur_ops.clamp_boxes(work, int(image_shape[1]), int(image_shape[2]))
    if work is not boxes:
        boxes.copy_(work)
    return target


def convert_to_xywh(boxes):
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin, ymin, xmax - xmin, ymax - ymin), dim=1)


# ---------------------------------------------------------------------------------------------
# per-image normalisation statistics
# ---------------------------------------------------------------------------------------------

_CANONICAL_MEAN = (0.485, 0.456, 0.406)
_CANONICAL_STD = (0.229, 0.224, 0.225)
# channel std of COCO train images after blurring: rows = (clean, E0..E4), per blur type P1..P3
# (data published in the reference, utils.py:228-230)
_BLUR_STD = np.asarray([
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2288, 0.2325], [0.2270, 0.2221, 0.2261],
     [0.2209, 0.2161, 0.2203], [0.2127, 0.2082, 0.2126], [0.2087, 0.2043, 0.2088]],
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2267, 0.2218, 0.2258],
     [0.2184, 0.2137, 0.2180], [0.2048, 0.2006, 0.2051], [0.1950, 0.1911, 0.1957]],
    [[0.2384, 0.2334, 0.2370], [0.2337, 0.2287, 0.2325], [0.2266, 0.2217, 0.2258],
     [0.2182, 0.2136, 0.2178], [0.2012, 0.1972, 0.2017], [0.1824, 0.1790, 0.1838]],
])
_BLUR_STD_SCALED = (_BLUR_STD * 0.229) / 0.2384     # utils.py:232-234


def get_norm_params(blur_dicts, use_custom_image_norm):
    """Returns (means, stds), each float64 [B,3] (or [1,3] when blur_dicts is None).
    Reference utils.py:219-273, including its quirk that a blurred image whose param_index is
    outside 0..2 (the stored-PSF off-by-one, transforms.py:427-428) keeps all-zero rows."""
    if blur_dicts is None:
        return np.array([_CANONICAL_MEAN]), np.array([_CANONICAL_STD])
    n = len(blur_dicts)
    means, stds = np.zeros((n, 3)), np.zeros((n, 3))
    for i, bd in enumerate(blur_dicts):
        custom = use_custom_image_norm and bd["blurring"] and bd["param_index"] is not None
        if not custom or bd["fraction_index"] == -1:
            means[i], stds[i] = _CANONICAL_MEAN, _CANONICAL_STD
        elif bd["param_index"] in (0, 1, 2):
            means[i] = _CANONICAL_MEAN
            stds[i] = _BLUR_STD_SCALED[bd["param_index"], bd["fraction_index"] + 1]
    return means, stds


def collate_fn(batch):
    return tuple(zip(*batch))


# ---------------------------------------------------------------------------------------------
# distributed / logging helpers (reference utils.py:474-785)
# ---------------------------------------------------------------------------------------------
import datetime  # noqa: E402
import errno  # noqa: E402
import os  # noqa: E402
import pickle  # noqa: E402
import time  # noqa: E402
from collections import defaultdict, deque  # noqa: E402

import torch.distributed as dist  # noqa: E402


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def mkdir(path):
    try:
        os.makedirs(path)
    except OSError as e:
        if e.errno != errno.EEXIST:
            raise


def setup_for_distributed(is_master):
    """Only rank 0 prints (pass force=True to print anyway).  Reference utils.py:719-731."""
    import builtins as __builtin__
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


_FORKSERVER = {"ctx": None}


def loader_context():
    """multiprocessing context for DataLoader workers: a fork SERVER started while this process has not touched the GPU yet,
    None (= fork from this process) otherwise.  Why: forking a process that holds a GPU context write-protects its pinned
    (GPU-registered) host memory for copy-on-write; every such fork makes the driver invalidate those mappings and park
    the process's queues until they are restored -- measured as 20-45 s in which the GPU does nothing at the start of every
    new loader (8 workers, `bench.py` sweep cells: 25 s per cell for 0.3 s of work).  Workers forked from a clean server
    leave this process alone.  Must be called (first) BEFORE the first CUDA call: the server is started by fork + exec."""
    if _FORKSERVER["ctx"] is None:
        import multiprocessing as mp
        if torch.cuda.is_initialized() or os.environ.get("DIB_LOADER_FORK"):
            _FORKSERVER["ctx"] = False
        else:
            try:
                ctx = mp.get_context("forkserver")
                ctx.set_forkserver_preload(["torch", "numpy", "detectinblur_amd.transforms", "detectinblur_amd.coco_utils", "detectinblur_amd.utils"])
                from multiprocessing import forkserver
                forkserver.ensure_running()
                _FORKSERVER["ctx"] = ctx
            except Exception as e:      # noqa: BLE001 -- no fork server on this platform: plain fork, with its stall
                print("detectinblur_amd: no fork server for the DataLoader workers (%s: %s); forking from this process" % (type(e).__name__, e))
                _FORKSERVER["ctx"] = False
    return _FORKSERVER["ctx"] or None


def init_distributed_mode(args):
    """torchrun-style rendezvous: RANK / WORLD_SIZE / LOCAL_RANK from the environment, one process
    per GPU.  Backend "nccl" IS RCCL on ROCm (xGMI inside a node); CPU-only hosts (tests) get gloo.
    Reference utils.py:763-785 hard-codes 'nccl'."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    print("| distributed init (rank {}): {}".format(args.rank, getattr(args, "dist_url", "env://")), flush=True)
    kw = {}
    if use_gpu:
        kw["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank, **kw)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def all_gather(data):
    """Gathers arbitrary picklable objects from every rank (reference utils.py:536-576)."""
    world_size = get_world_size()
    if world_size == 1:
        return [data]
    out = [None] * world_size
    dist.all_gather_object(out, data)
    return out


def reduce_dict(input_dict, average=True):
    """All-reduces the values of a dict of scalar tensors (one stacked collective).  Reference
    utils.py:579-603; used for logging only -- the gradients go through DDP's bucketed all-reduce."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}


class SmoothedValue(object):
    """Windowed median / average plus the global average of a series (reference utils.py:474-533)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / max(self.count, 1)

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        if not self.deque:          # nothing logged yet (the training loop logs a step once the next one is enqueued)
            return "-"
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        raise AttributeError(attr)

    def __str__(self):
        return self.delimiter.join("{}: {}".format(n, str(m)) for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                msg = [header, "[{}/{}]".format(i, n), "eta: " + eta, str(self), "time: " + str(iter_time), "data: " + str(data_time)]
                if torch.cuda.is_available():
                    msg.append("max mem: {:.0f}".format(torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)))
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print("{} Total time: {} ({:.4f} s / it)".format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


def warmup_lr_scheduler(optimizer, warmup_iters, warmup_factor):
    """Linear warm-up from warmup_factor to 1 over warmup_iters steps (reference utils.py:700-708)."""
    def f(x):
        if x >= warmup_iters:
            return 1
        alpha = float(x) / warmup_iters
        return warmup_factor * (1 - alpha) + alpha
    return torch.optim.lr_scheduler.LambdaLR(optimizer, f)


def make_sgd(params, lr, momentum, weight_decay, foreach=False):
    """The reference's optimizer (train.py:245-246: torch.optim.SGD with momentum and weight decay).  For parameters on the GPU
    torch's fused implementation is asked for -- the same update in 3 launches instead of 15 (0.44 -> 0.21 ms per step at the
    detector's 83 trainable tensors, scratch/t_fused_sgd.py); its multiply-adds round once where the default (foreach)
    implementation rounds twice, a difference of <= 2.2e-7 of a tensor's largest weight per step, three orders of magnitude below
    what two runs of the same step differ by (atomic accumulation in MIOpen's weight-gradient kernels).  `foreach=True` (train.py
    --foreach_sgd) selects torch's default."""
    import torch
    params = list(params)
    fused = (not foreach) and bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    kw = {"fused": True} if fused else {}
    return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, **kw)


def restore_sgd_implementation(optimizer, foreach=False):
    """After `optimizer.load_state_dict(...)`: a checkpoint's param_groups arrive wholesale, `fused` / `foreach` included -- a
    reference-written checkpoint (no `fused` key) would silently switch the run to the foreach kernels, one written on the GPU
    and resumed with the parameters on the CPU would raise in step().  The implementation is chosen again, from the parameters
    as they are now, exactly as `make_sgd` chooses it."""
    import torch
    for group in optimizer.param_groups:
        ps = group["params"]
        fused = (not foreach) and bool(ps) and all(p.is_cuda and p.dtype == torch.float32 for p in ps)
        group["fused"] = True if fused else None
        group["foreach"] = None
    return optimizer
