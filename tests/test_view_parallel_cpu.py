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
