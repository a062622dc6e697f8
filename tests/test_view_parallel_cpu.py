"""CPU (gloo, world_size 2): the view-parallel host logic -- view sharding and the single flat-buffer all-reduce --
reproduces the single-process sum of per-view curve gradients."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(B=37, m=12):
    g = torch.Generator().manual_seed(0)
    return {"curve_points": torch.randn(B, 4, 3, generator=g).requires_grad_(True),
            "width": torch.randn(B, 1, generator=g).requires_grad_(True),
            "opacity": torch.randn(B, 1, generator=g).requires_grad_(True),
            "mask": torch.randn(B, m, 1, generator=g).requires_grad_(True)}


def _view_loss(p, v):
    """A deterministic stand-in for 'render view v and take the loss' (no GPU in this test)."""
    s = 0.1 * (v + 1)
    return ((p["curve_points"] * s).sin().sum() + (p["width"] * (s + 1)).pow(2).sum() +
            (p["opacity"] * s).exp().sum() + (torch.sigmoid(p["mask"]) * s).sum())


def _worker(rank, world, port, n_views, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from curve_gaussian_amd.view_parallel import FlatGrads, shard_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = _params()
    flat = FlatGrads(p)
    flat.zero_()
    mine = shard_views(n_views, rank, world)
    for v in mine:
        _view_loss(p, v).backward()   # accumulates into the flat buffer through the .grad views
    flat.all_reduce()
    if rank == 0:
        torch.save({"flat": flat.flat.clone(), "mine": mine, "cp": flat.view("curve_points").clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_all_reduce_equals_single_process_sum(tmp_path):
    from curve_gaussian_amd.view_parallel import FlatGrads, shard_views
    n_views, world = 7, 2
    assert sorted(shard_views(n_views, 0, world) + shard_views(n_views, 1, world)) == list(range(n_views))
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, _free_port(), n_views, out), nprocs=world, join=True)
    got = torch.load(out)
    p = _params()
    ref = FlatGrads(p)
    for v in range(n_views):
        _view_loss(p, v).backward()
    assert got["mine"] == [0, 2, 4, 6]
    torch.testing.assert_close(got["flat"], ref.flat, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(got["cp"], p["curve_points"].grad, rtol=1e-6, atol=1e-6)


def test_flat_layout_matches_survey_message_size():
    from curve_gaussian_amd.view_parallel import FlatGrads
    assert FlatGrads.floats_per_curve(12, 0) == 38          # 152 B / curve
    p = _params(B=5)
    f = FlatGrads(p)
    assert f.flat.numel() == 5 * (12 + 1 + 1 + 12)
    p["width"].grad.add_(1.0)
    a, b = f.slices["width"]
    assert float(f.flat[a:b].sum()) == 5.0 and float(f.flat.sum()) == 5.0
    f.all_reduce()  # no process group: no-op


# ---------------------------------------------------------------- topology edits stay rank-consistent
def _cpu_model(B=40, seed=3):
    """The product's curve model and topology code on CPU tensors: only prepare_scaling_rot (a HIP op in the product) is
    swapped for the torch restatement of the oracle -- this test is about the host logic around it."""
    sys.path.insert(0, ROOT)
    from curve_gaussian_amd import synthetic as S
    from curve_gaussian_amd.scene import GaussianCurveModel
    from oracle import torch_ref as TR

    class CpuModel(GaussianCurveModel):
        def prepare_scaling_rot(self, eps=1e-8):
            self._xyz, self._rotation, self._scaling = TR.prepare_scaling_rot(self._curve_points, self._width,
                                                                              self.is_bezier, self.n_gaussians, eps)
    g = torch.Generator().manual_seed(seed)
    c = S.make_curves(B, seed)
    c["opacity"] = torch.randn(B, 1, generator=g) * 2
    gm = CpuModel(0, 12, device="cpu").create_from_curves(c["curve_points"], c["width"], c["opacity"], c["mask"], c["is_bezier"])
    gm.training_setup()
    return gm


def _local_stats(gm, rank):
    """What one rank accumulates over ITS views: different on every rank."""
    g = torch.Generator().manual_seed(100 + rank)
    P = gm._xyz.shape[0]
    for _ in range(3):
        class VS:
            grad = torch.randn(P, 3, generator=g) * 3e-4
        gm.add_densification_stats(VS, torch.rand(P, generator=g) > 0.4)
    gm.max_radii2D = torch.rand(P, generator=g) * 10


def _topology_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from curve_gaussian_amd.view_parallel import FlatGrads
    gm = _cpu_model()
    _local_stats(gm, rank)
    thr = 6.2e-4
    gm.densify_and_prune(thr, 0.1, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))
    named = {"curve_points": gm._curve_points, "width": gm._width, "opacity": gm._opacity, "mask": gm._mask}
    flat = FlatGrads(named)          # the next step's exchange buffer: same size on every rank, or all_reduce would fail
    flat.flat.fill_(float(rank + 1))
    flat.all_reduce()
    torch.save({"cp": gm._curve_points.detach().clone(), "op": gm._opacity.detach().clone(), "isb": gm.is_bezier.clone(),
                "n": flat.flat.numel(), "sum": float(flat.flat[0]), "denom": gm.denom.clone()}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_take_the_same_densify_and_prune_decisions(tmp_path):
    """Each rank accumulates densification statistics over its own views; densify_and_prune all-reduces them first
    (sum, sum, max), so both ranks split / prune the same curves and end with identical shapes and parameters -- equal to
    a single process that saw all the views."""
    out = str(tmp_path / "topo.pt")
    mp.spawn(_topology_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert a["n"] == b["n"] and a["sum"] == 3.0
    assert torch.equal(a["cp"], b["cp"]) and torch.equal(a["op"], b["op"]) and torch.equal(a["isb"], b["isb"])
    # single process with the statistics of both ranks
    gm = _cpu_model()
    other = _cpu_model()
    _local_stats(gm, 0)
    _local_stats(other, 1)
    gm.xyz_gradient_accum += other.xyz_gradient_accum
    gm.denom += other.denom
    n_before = gm._curve_points.shape[0]
    gm.densify_and_prune(6.2e-4, 0.1, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))
    assert gm._curve_points.shape[0] != n_before          # the edit really changed the topology
    assert torch.equal(a["cp"], gm._curve_points.detach()) and torch.equal(a["op"], gm._opacity.detach())
    # without the synchronisation the two ranks WOULD diverge: their local statistics select different curves
    g0, g1 = _cpu_model(), _cpu_model()
    _local_stats(g0, 0)
    _local_stats(g1, 1)
    sel = lambda g: (((g.xyz_gradient_accum / g.denom).nan_to_num(0.0)).reshape(-1, 12).max(1).values >= 6.2e-4)
    assert not torch.equal(sel(g0), sel(g1))


def _two_edits_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gm = _cpu_model()
    n0 = gm._curve_points.shape[0]
    _local_stats(gm, rank)
    gm.densify_and_prune(1.0, 0.0, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))   # threshold out of reach: no split, no prune
    assert gm._curve_points.shape[0] == n0
    _local_stats(gm, rank + 10)                                                               # the interval's next views
    gm.densify_and_prune(6.2e-4, 0.1, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))
    torch.save({"cp": gm._curve_points.detach().clone(), "op": gm._opacity.detach().clone()}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_statistics_survive_a_decision_without_a_split(tmp_path):
    """The statistics buffers are only reset when a split happens (densification_postfix); a densify_and_prune that splits
    nothing leaves them accumulating.  The all-reduce of the view-parallel decision therefore must not write the global sums
    back into the rank-local buffers -- the second decision would count the first interval once per rank again.  Two ranks,
    two consecutive decisions (the first out of reach), against ONE process that saw the views of both ranks."""
    out = str(tmp_path / "topo2.pt")
    mp.spawn(_two_edits_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(a["cp"], b["cp"]) and torch.equal(a["op"], b["op"])
    gm = _cpu_model()
    n0 = gm._curve_points.shape[0]
    for seed in (0, 1):
        o = _cpu_model()
        _local_stats(o, seed)
        gm.xyz_gradient_accum += o.xyz_gradient_accum
        gm.denom += o.denom
    gm.densify_and_prune(1.0, 0.0, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))
    for seed in (10, 11):
        o = _cpu_model()
        _local_stats(o, seed)
        gm.xyz_gradient_accum += o.xyz_gradient_accum
        gm.denom += o.denom
    gm.densify_and_prune(6.2e-4, 0.1, 1.0, 20, torch.zeros(gm._xyz.shape[0], dtype=torch.int32))
    assert gm._curve_points.shape[0] != n0
    assert torch.equal(a["cp"], gm._curve_points.detach()) and torch.equal(a["op"], gm._opacity.detach())


def test_ranks_draw_disjoint_views_from_one_stream():
    """TrainStep._next_view: every rank runs the same random stream and takes its own element of each group of `world`
    draws -- no view is rendered twice in one step, every view once per epoch (train.py:85-90 across ranks)."""
    import random
    from curve_gaussian_amd.train_step import TrainStep

    def draws(rank, world, n_cams, steps):
        t = TrainStep.__new__(TrainStep)
        t.rng, t.stack, t.rank, t.world, t.cams = random.Random(5), [], rank, world, list(range(n_cams))
        return [t._next_view() for _ in range(steps)]
    a, b = draws(0, 2, 8, 8), draws(1, 2, 8, 8)
    for i in range(8):
        assert a[i] != b[i]
    assert sorted(a[:4] + b[:4]) == list(range(8)) and sorted(a[4:] + b[4:]) == list(range(8))
    assert sorted(draws(0, 1, 5, 5)) == list(range(5))


def test_no_gc_suspends_the_cyclic_collector_and_restores_it():
    """view_parallel.no_gc wraps stream captures: a cyclic collection inside a capture finalises retired graphs / events
    whose HIP destroy calls are illegal there (observed as an abort in a captured autograd backward)."""
    import gc
    from curve_gaussian_amd.view_parallel import no_gc
    assert gc.isenabled()
    with no_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    gc.disable()
    try:
        with no_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()      # a caller that runs without the collector keeps running without it
    finally:
        gc.enable()
