"""Host-side cost of the pieces of one eager training iteration (run on the GPU box):
    python profiles/probes/host_breakdown.py [cfg1] [n]
Each piece is called n times back to back WITHOUT synchronising in between (the GPU runs behind), so the figure is what the
host spends enqueueing it; pieces that block on the device (finish, nonzero) are marked."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curve_gaussian_amd import _lib as L  # noqa: E402
from curve_gaussian_amd import synthetic as S  # noqa: E402
from curve_gaussian_amd.gaussian_renderer import PipelineParams, render  # noqa: E402
from curve_gaussian_amd.ops import view_render as VR  # noqa: E402
from curve_gaussian_amd.ops.losses import photometric_loss  # noqa: E402
from curve_gaussian_amd.scene import GaussianCurveModel  # noqa: E402
from curve_gaussian_amd.train_step import TrainStep, unit_grad  # noqa: E402


def timeit(name, fn, n, sync_before=True):
    for _ in range(5):
        fn()
    if sync_before:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:58s} {dt:8.1f} us")
    return dt


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda:0")
    curves, cams = S.make_config(cfg, n_views=4)
    cams = [c.to(dev) for c in cams]
    cam = cams[0]
    H, W = cam.image_height, cam.image_width
    gm = GaussianCurveModel(0, 12, device=dev).create_from_curves(curves["curve_points"], curves["width"], curves["opacity"],
                                                                  curves["mask"], curves["is_bezier"])
    g = torch.Generator().manual_seed(1)
    gts = [((torch.rand(1, H, W, generator=g) > 0.97).float() * torch.rand(1, H, W, generator=g)).to(dev) for _ in cams]
    ts = TrainStep(gm, cams, gts)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    dL = (torch.randn(1, H, W, generator=g) * 1e-3).to(dev)
    import math
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    render(cam, gm, pipe, bg)
    xyz = gm.get_xyz
    print(f"--- {cfg}: P = {xyz.shape[0]}, {W}x{H}, shim = {L.use_shim()}")
    timeit("torch.zeros_like(xyz, requires_grad=True)", lambda: torch.zeros_like(xyz, requires_grad=True), n)
    timeit("torch.empty(1,H,W) x1", lambda: torch.empty(1, H, W, device=dev), n)
    z = torch.zeros_like(xyz, requires_grad=True)

    def fwd_only():
        pend = []
        with torch.no_grad():
            out = VR.view_render(gm._curve_points, gm._width, gm._opacity, None, z, gm.is_bezier, 12, 0.01, bg, cam, tanx, tany, 0, None,
                                 True, True, pend)
        VR.finish(pend[0])
        return out
    timeit("view_render forward (no grad) + finish [blocks on scatter]", fwd_only, n)

    def fwd_grad():
        pend = []
        out = VR.view_render(gm._curve_points, gm._width, gm._opacity, None, z, gm.is_bezier, 12, 0.01, bg, cam, tanx, tany, 0, None,
                             True, True, pend)
        VR.finish(pend[0])
        return out
    timeit("view_render forward (autograd node) + finish", fwd_grad, n)

    def fwd_bwd():
        out = fwd_grad()
        torch.autograd.backward(out[0], dL)
    timeit("view_render forward + backward", fwd_bwd, n)
    timeit("render(...) defaults (visibility + rend_dir)", lambda: render(cam, gm, pipe, bg), n)
    timeit("render(...) no visibility / rend_dir / clamp", lambda: render(cam, gm, pipe, bg, compute_visibility=False, clamp=False,
                                                                          compute_rend_dir=False), n)
    radii = render(cam, gm, pipe, bg)["radii"]
    timeit("(radii > 0).nonzero()  [device sync]", lambda: (radii > 0).nonzero(), n)
    nv = int((radii > 0).sum())
    timeit("nonzero_static(radii > 0, n)", lambda: torch.nonzero_static(radii > 0, size=nv), n)
    img = render(cam, gm, pipe, bg, clamp=False)["render"].detach().requires_grad_(True)
    timeit("photometric_loss forward", lambda: photometric_loss(img, gts[0][:1], 10.0, 0.1, clamp=True), n)

    def loss_fb():
        photometric_loss(img, gts[0][:1], 10.0, 0.1, clamp=True).backward(gradient=unit_grad(dev))
    timeit("photometric_loss forward + backward", loss_fb, n)
    timeit("update_learning_rate", lambda: gm.update_learning_rate(10), n)
    for p in (gm._curve_points, gm._width, gm._opacity):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    timeit("optimizer.step(zero_grad=True)", lambda: gm.optimizer.step(zero_grad=True), n)
    timeit("prepare_scaling_rot()", lambda: gm.prepare_scaling_rot(), n)
    timeit("TrainStep.step()", lambda: ts.step(), n)
    st = L.raw_stream(dev)
    timeit("raw_stream(dev)", lambda: L.raw_stream(dev), n)
    lib = L.load()
    timeit("one trivial ctypes call (cgs_version)", lambda: lib.cgs_version(), n)
    e = torch.cuda.Event()
    timeit("torch.cuda.Event.record()", lambda: e.record(), n)
    a = torch.zeros(16, device=dev)
    timeit("a.add_(1) (one tiny torch kernel launch)", lambda: a.add_(1), n)


if __name__ == "__main__":
    main()
