"""A/B of TrainStep(fused=True) with and without gradient sinks (render(grad_sinks=...)), same process, alternating chunks:
    python profiles/probes/grad_sinks_ab.py [cfg1] [chunks]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curve_gaussian_amd import synthetic as S  # noqa: E402
from curve_gaussian_amd import train_step as TS  # noqa: E402
from curve_gaussian_amd.scene import GaussianCurveModel  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    dev = torch.device("cuda:0")
    curves, cams = S.make_config(cfg, n_views=8)
    cams = [c.to(dev) for c in cams]
    H, W = cams[0].image_height, cams[0].image_width
    g = torch.Generator().manual_seed(1)
    gts = [((torch.rand(1, H, W, generator=g) > 0.97).float() * torch.rand(1, H, W, generator=g)).to(dev) for _ in cams]
    orig = TS.render
    steps = {}
    for sinks in (False, True):
        gm = GaussianCurveModel(0, 12, device=dev).create_from_curves(curves["curve_points"], curves["width"], curves["opacity"],
                                                                      curves["mask"], curves["is_bezier"])
        steps[sinks] = TS.TrainStep(gm, cams, gts)
    res = {False: [], True: []}
    for c in range(chunks + 2):
        for sinks in (False, True):
            TS.render = (lambda *a, **k: orig(*a, **{**k, "grad_sinks": False})) if not sinks else orig
            ts = steps[sinks]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(16):
                ts.step()
            torch.cuda.synchronize()
            if c >= 2:
                res[sinks].append((time.perf_counter() - t0) / 16 * 1e3)
    for sinks in (False, True):
        v = sorted(res[sinks])
        print(f"{cfg} grad_sinks={sinks}: median {v[len(v) // 2]:.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}")


if __name__ == "__main__":
    main()
