"""View-parallel data parallelism for the per-view hot path (new capability; the reference is single-GPU).

One process per GPU, identical curve parameters on every rank, rank r renders views r, r+N, ... of the step's view
batch, and ONE all-reduce (RCCL over xGMI on the GPU box, gloo in the CPU tests) sums the curve-level gradients.
The gradients of all learnable tensors live in a single flat buffer so the exchange needs no packing kernels:

    [ _curve_points 12 | _width 1 | _opacity 1 | _mask m | _features_dc m | _features_rest m*((D+1)^2-1) ]  floats/curve

Device-agnostic host logic (plain torch tensors + torch.distributed)."""
from typing import Dict, List, Sequence

import torch


def shard_views(n_views: int, rank: int, world: int) -> List[int]:
    """Indices of the views rank `rank` renders: r, r+N, r+2N, ... (every view exactly once across ranks)."""
    return list(range(rank, n_views, world))


def _dist_world(group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, 1
    return dist, dist.get_world_size(group)


def global_densification_stats(g, group=None):
    """The densification statistics of ALL ranks' views, for a topology decision of a view-parallel run: every rank
    accumulates ``xyz_gradient_accum`` / ``denom`` (gaussian_model.py:618-620) over ITS views only, so the decision
    thresholds the quotient of the all-reduced sums (gaussian_curve_model.py:352: the mean over the views of all ranks).
    Returns ``(accum, denom)`` as NEW tensors: the rank-local buffers are left untouched, because they keep accumulating
    until a split resets them (densification_postfix) -- reducing them in place would count the views seen so far once per
    rank again at the next decision.  ``max_radii2D`` is maximised in place (idempotent).  With identical parameters (same
    all-reduced gradients, same Adam step) every rank then takes the same split / prune decisions and the flat buffers
    keep the same size.  8 B/splat + 4 B/splat, only on densification iterations.  Without a process group:
    the local buffers themselves."""
    dist, world = _dist_world(group)
    if world == 1:
        return g.xyz_gradient_accum, g.denom
    accum, denom = g.xyz_gradient_accum.clone(), g.denom.clone()
    if accum.numel():
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    t = getattr(g, "max_radii2D", None)
    if t is not None and t.numel():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return accum, denom


def assert_same_topology(n_curves: int, device=None, group=None):
    """Raises on every rank if the ranks disagree on the number of curves (the next gradient all-reduce would mix
    buffers of different sizes)."""
    dist, world = _dist_world(group)
    if world == 1:
        return
    t = torch.tensor([n_curves, -n_curves], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError(f"view-parallel ranks diverged: curve counts between {-int(t[1])} and {int(t[0])} after a "
                           "topology edit (were the densification statistics synchronised?)")


class ViewStreams:
    """Keeps several independent views in flight on one GPU by rotating them over `n` HIP streams.

    The per-view pipeline alternates latency-bound kernels (curve sampling, preprocess, scatter, tile sort: dependent
    memory round trips and atomics) with VALU-bound ones (forward/backward compositing); views are independent
    (BASELINE north_star), so running view i+1's binning under view i's compositing fills both.  Gradients of
    all views of a step accumulate into the same ``.grad`` buffers (autograd serialises the accumulation), exactly
    like the views of the other ranks do through the all-reduce.  `fork()` / `join()` order the streams against
    the caller's current stream with events only -- the host never blocks."""

    def __init__(self, n: int, device=None):
        self.n = max(1, int(n))
        self.streams = [torch.cuda.Stream(device) for _ in range(self.n)] if self.n > 1 else []

    def fork(self):
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)

    def run(self, i: int, fn, *args, **kw):
        if not self.streams:
            return fn(*args, **kw)
        with torch.cuda.stream(self.streams[i % self.n]):
            return fn(*args, **kw)

    def join(self):
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)


class StaticCamera:
    """Camera whose pose tensors are fixed device buffers (one packed [35] float tensor: view 16 | proj 16 | centre 3),
    so that a captured hipGraph can be replayed for any view after one small stream-ordered copy."""

    def __init__(self, proto, device):
        self.image_height, self.image_width = int(proto.image_height), int(proto.image_width)
        self.FoVx, self.FoVy = proto.FoVx, proto.FoVy
        self.pack = torch.zeros(35, dtype=torch.float32, device=device)
        self.world_view_transform = self.pack[0:16].view(4, 4)
        self.full_proj_transform = self.pack[16:32].view(4, 4)
        self.camera_center = self.pack[32:35]

    @staticmethod
    def packed(cam):
        return torch.cat([cam.world_view_transform.reshape(-1), cam.full_proj_transform.reshape(-1),
                          cam.camera_center.reshape(-1)]).float().contiguous()

    def load(self, packed):
        self.pack.copy_(packed, non_blocking=True)


def capture_graph(fn, stream, warmup=2):
    """Warm `fn` up on `stream`, then capture it there as a hipGraph; returns (graph, fn's captured return value).
    Warm-up and capture share the stream so the autograd gradient accumulators created by the warm-up live on the
    capture stream (a captured backward that has to synchronise with the default stream cannot be captured)."""
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(stream)
    graph = torch.cuda.CUDAGraph()
    with no_gc(), torch.cuda.graph(graph, stream=stream):
        out = fn()
    return graph, out


class no_gc:
    """Keeps Python's cyclic collector from running inside a stream capture.  A collection that happens to start while the
    stream is capturing (observed inside the autograd thread of a captured backward) finalises whatever garbage is around --
    retired graphs, events, tensors -- and their HIP destroy / free calls are illegal during capture: the process aborts.
    torch.cuda.graph collects once on entry; this closes the window until exit."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *a):
        import gc
        if self._was:
            gc.enable()
        return False


class FlatGrads:
    """Owns the flat gradient buffer and installs views of it as the ``.grad`` of the given parameters."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.names = list(params)
        self.params = params
        total = sum(p.numel() for p in params.values())
        any_p = next(iter(params.values()))
        self.flat = torch.zeros(total, dtype=torch.float32, device=any_p.device)
        self.slices = {}
        o = 0
        for n, p in params.items():
            self.slices[n] = (o, o + p.numel())
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero_(self):
        self.flat.zero_()

    def view(self, name: str) -> torch.Tensor:
        a, b = self.slices[name]
        return self.flat[a:b].view_as(self.params[name])

    def all_reduce(self, group=None, average: bool = False):
        """Sum (or mean) over ranks, in place.  No-op without an initialised process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            self.flat.div_(dist.get_world_size(group))

    @staticmethod
    def floats_per_curve(m: int = 12, sh_degree: int = 0) -> int:
        return 12 + 1 + 1 + m + m + m * ((sh_degree + 1) ** 2 - 1)
