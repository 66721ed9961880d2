"""GraphCache's policy (detectinblur_amd/graphs.py) without a GPU: a shape is captured on its n-th sighting, a hit refreshes
its position, the least recently used graph goes first, clear() forgets everything."""
import torch


def test_lru_capture_after_and_clear(monkeypatch):
    from detectinblur_amd import graphs
    made = []

    class Fake(object):
        def __init__(self, fn, example, warmup=2, pool=None):
            self.shape = tuple(example.shape)
            made.append(self.shape)

        def __call__(self, x):
            return ("graph", self.shape)

    monkeypatch.setattr(graphs, "StaticGraph", Fake)
    monkeypatch.setattr(torch.cuda, "graph_pool_handle", lambda: (0, 0))
    cache = graphs.GraphCache(lambda x: ("eager", tuple(x.shape)), limit=2, capture_after=2)
    a, b, c = torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 8), torch.zeros(1, 3, 8, 8)
    assert cache(a)[0] == "eager" and not made                 # first sighting: eager
    assert cache(a)[0] == "graph" and made == [(1, 3, 4, 4)]   # second: captured
    assert cache(b)[0] == "eager" and cache(b)[0] == "graph"
    assert cache(a)[0] == "graph"                              # a hit moves `a` behind `b`
    assert cache(c)[0] == "eager" and cache(c)[0] == "graph"   # third shape: the least recently used (`b`) goes
    assert list(k[0] for k in cache.graphs) == [(1, 3, 4, 4), (1, 3, 8, 8)]
    assert cache(a)[0] == "graph" and len(made) == 3
    assert cache(b)[0] == "graph" and len(made) == 4           # seen often enough before: recaptured at once, evicting `c`
    assert list(k[0] for k in cache.graphs) == [(1, 3, 4, 4), (1, 3, 4, 8)]
    cache.clear()
    assert not cache.graphs and cache(a)[0] == "eager"


def test_failed_capture_stays_eager(monkeypatch):
    from detectinblur_amd import graphs

    class Boom(object):
        def __init__(self, *a, **k):
            raise RuntimeError("an op synchronised during capture")

    monkeypatch.setattr(graphs, "StaticGraph", Boom)
    monkeypatch.setattr(torch.cuda, "graph_pool_handle", lambda: (0, 0))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    calls = []
    cache = graphs.GraphCache(lambda x: calls.append(1) or "eager", limit=2, capture_after=1)
    x = torch.zeros(1, 3, 4, 4)
    assert cache(x) == "eager" and cache(x) == "eager" and len(calls) == 2
    assert list(cache.graphs.values()) == [None]
