"""One training-iteration form, n steps, for a kernel trace or a wall-clock figure (run on the GPU box):
    python profiles/probes/train_step_modes.py cfg3 graph|direct|autograd [n] [warm-up steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curve_gaussian_amd import synthetic as S  # noqa: E402
from curve_gaussian_amd.scene import GaussianCurveModel  # noqa: E402
from curve_gaussian_amd.train_step import GraphedTrainStep, TrainStep  # noqa: E402


def main():
    cfg, mode = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    warm = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    dev = torch.device("cuda:0")
    curves, cams = S.make_config(cfg, n_views=8)
    cams = [c.to(dev) for c in cams]
    H, W = cams[0].image_height, cams[0].image_width
    g = torch.Generator().manual_seed(1)
    gts = [((torch.rand(1, H, W, generator=g) > 0.97).float() * torch.rand(1, H, W, generator=g)).to(dev) for _ in cams]
    gm = GaussianCurveModel(0, 12, device=dev).create_from_curves(curves["curve_points"], curves["width"], curves["opacity"],
                                                                  curves["mask"], curves["is_bezier"])
    ts = GraphedTrainStep(gm, cams, gts) if mode == "graph" else TrainStep(gm, cams, gts, direct=(mode == "direct"))
    for _ in range(warm):
        ts.step()
    if mode == "graph":
        ts.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step()
    if mode == "graph":
        ts.finish()
    torch.cuda.synchronize()
    print(f"{cfg} {mode}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per iteration ({n} after {warm} warm-up steps)")
    for rep in range(3):   # the same again: steady state
        t0 = time.perf_counter()
        for _ in range(n):
            ts.step()
        if mode == "graph":
            ts.finish()
        torch.cuda.synchronize()
        print(f"    again: {(time.perf_counter() - t0) / n * 1e3:.4f} ms")


if __name__ == "__main__":
    main()
