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
    def __init__(self, fn, example, warmup=2):
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
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = fn(self.static_in)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.out


class GraphCache(object):
    """shape-keyed StaticGraphs of one function; a shape whose capture failed runs eagerly from then on."""

    def __init__(self, fn, limit=8):
        self.fn, self.limit, self.graphs = fn, limit, {}

    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device, x.is_contiguous(memory_format=torch.channels_last))
        g = self.graphs.get(key)
        if g is None and key not in self.graphs:
            if len(self.graphs) >= self.limit:
                self.graphs.pop(next(iter(self.graphs)))
            try:
                g = StaticGraph(self.fn, x)
            except Exception as e:      # noqa: BLE001 -- e.g. an op that synchronises; stay correct, stay eager
                import sys
                sys.stderr.write("detectinblur_amd.graphs: capture failed for %s (%s: %s); running eagerly\n" % (key[0], type(e).__name__, e))
                torch.cuda.synchronize()
                g = None
            self.graphs[key] = g
        return g(x) if g is not None else self.fn(x)
