"""HIP graphs for launch-bound inference loops.

At batch size 1 (the reference evaluates one image at a time, evaluate.py:335-339) the static part of the detector --
ResNet-50 + FPN + RPN head + proposal decoding, top-k, sort and NMS: ~300 launches of a few microseconds each -- costs the
Python interpreter more than it costs the GPU.  `StaticGraph` captures a function of ONE input tensor into a HIP graph
after two warm-up calls (MIOpen's kernel selection and every allocation happen there) and replays it on later calls: the
input is copied into the captured buffer, the outputs are the captured tensors (valid until the next replay, consumed in
stream order by whatever follows).  No tracing compiler: the graph is the recorded launches of the eager code.
"""
import torch


class StaticGraph(object):
    """The graph holds RAW POINTERS to everything its launches read.  What lives outside the capture's memory pool and is not
    kept alive by anybody else (cached anchors, lookup tables) must therefore be among `fn`'s outputs: `self.out` is the
    reference that keeps it alive as long as the graph (generalized_rcnn._trunk returns its anchors for that reason)."""

    def __init__(self, fn, example, warmup=2, pool=None):
        self.static_in = example.clone(memory_format=torch.preserve_format)
        cur = torch.cuda.current_stream(example.device)
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a DataLoader's pin-memory thread allocating host memory must not invalidate the capture
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
            self.out = fn(self.static_in)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.out


class GraphCache(object):
    """Shape-keyed StaticGraphs of one function, least recently used first out; a shape whose capture failed runs eagerly from
    then on.  A shape is captured when it is seen for the `capture_after`-th time (rare shapes of a varied dataset stay eager
    instead of evicting a common shape's graph: the reference evaluates COCO at batch 1, dozens of padded shapes).  All graphs
    of one cache share ONE memory pool: they are replayed one at a time and their outputs are consumed in stream order before
    the next replay, so the pool holds the largest shape's activations, not the sum over the shapes (what a graph keeps for itself
    are its input and output buffers: ~0.1 GB for the detector's trunk at batch 1).  `limit`: real COCO images reach ~50 padded
    input shapes after the transform (the grid `python -m detectinblur_amd.kernel_choices --fill --shapes coco-eval` fills the kernel-choice data for); a cache smaller
    than that would evict and recapture (three forward passes each) all through an evaluation."""

    def __init__(self, fn, limit=64, capture_after=2):
        self.fn, self.limit, self.capture_after = fn, limit, capture_after
        self.graphs, self.seen, self.pool = {}, {}, None

    def clear(self):
        self.graphs.clear()
        self.seen.clear()

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device, x.is_contiguous(memory_format=torch.channels_last))
        if key in self.graphs:
            g = self.graphs.pop(key)
            self.graphs[key] = g                    # most recently used last
            return g(x) if g is not None else self.fn(x)
        n = self.seen[key] = self.seen.pop(key, 0) + 1                 # re-inserted: most recently seen last
        while len(self.seen) > 64 * self.limit:                         # a varied dataset's rare shapes: forget the oldest sightings
            self.seen.pop(next(iter(self.seen)))
        if n < self.capture_after:
            return self.fn(x)
        while len(self.graphs) >= self.limit:
            self.graphs.pop(next(iter(self.graphs)))        # least recently used
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        try:
            g = StaticGraph(self.fn, x, pool=self.pool)
        except Exception as e:      # noqa: BLE001 -- e.g. an op that synchronises; stay correct, stay eager
            import sys
            sys.stderr.write("detectinblur_amd.graphs: capture failed for %s (%s: %s); running eagerly\n" % (key[0], type(e).__name__, e))
            torch.cuda.synchronize()
            g = None
        self.graphs[key] = g
        return g(x) if g is not None else self.fn(x)
