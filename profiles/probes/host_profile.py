"""Where the eager drop-in routes spend their HOST time (run on the GPU box):
    python profiles/probes/host_profile.py [cfg1|cfg2|cfg3] [n_iter]
For each of: render() fused route + backward, render(fused=False) + backward, TrainStep.step() -- wall ms per iteration
(GPU synchronised only at the end: host-bound when the GPU idles) and the top cProfile entries by cumulative host time."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curve_gaussian_amd import synthetic as S  # noqa: E402
from curve_gaussian_amd.gaussian_renderer import PipelineParams, render  # noqa: E402
from curve_gaussian_amd.scene import GaussianCurveModel  # noqa: E402
from curve_gaussian_amd.train_step import TrainStep  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    curves, cams = S.make_config(cfg, n_views=8)
    cams = [c.to(dev) for c in cams]
    H, W = cams[0].image_height, cams[0].image_width
    gm = GaussianCurveModel(0, 12, device=dev).create_from_curves(curves["curve_points"], curves["width"], curves["opacity"],
                                                                  curves["mask"], curves["is_bezier"])
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    g = torch.Generator().manual_seed(1)
    dL = (torch.randn(1, H, W, generator=g) * 1e-3).to(dev)
    gts = [((torch.rand(1, H, W, generator=g) > 0.97).float() * torch.rand(1, H, W, generator=g)).to(dev) for _ in cams]

    def dropin(**kw):
        def f(i):
            pkg = render(cams[i % len(cams)], gm, pipe, bg, **kw)
            torch.autograd.backward(pkg["render"], dL, retain_graph=True)
        return f

    ts = TrainStep(gm, cams, gts)
    cases = [("render() fused + backward", dropin()), ("render(fused=False) + backward", dropin(fused=False)),
             ("TrainStep.step()", lambda i: ts.step())]
    for name, fn in cases:
        for i in range(8):
            fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_it):
            fn(i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"=== {cfg} {name}: {t_all / n_it * 1e3:.4f} ms/iter wall, {t_host / n_it * 1e3:.4f} ms/iter host enqueue")
        pr = cProfile.Profile()
        pr.enable()
        for i in range(n_it):
            fn(i)
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
        lines = s.getvalue().splitlines()
        print("\n".join(l[:170] for l in lines[4:44]))


if __name__ == "__main__":
    main()
