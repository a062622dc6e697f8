"""Per-kernel times of the per-view path (cgs_view_forward + cgs_view_backward, serial, eager) through the ctypes bindings only,
so that `CGS_LIB=<experiment build>` selects the library (the compiled shim always binds curve_gaussian_amd/libcurvegs.so).

    python profiles/probes/kernel_times.py [cfg3] [views]     ->  one JSON line: {"lib": ..., "kernel_us": {...}, "view_us": ...}

Used for A/B pairs and for the what-if builds of profiles/r06_experiments.md (kernels with one cost removed: their images
are wrong on purpose, only the times mean something)."""
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from curve_gaussian_amd import _lib as L  # noqa: E402
from curve_gaussian_amd import synthetic as S  # noqa: E402
from curve_gaussian_amd.ops import curve_sampling  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n_views = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device("cuda", 0)
    lib = L.load()
    curves, cams = S.make_config(cfg, n_views=n_views)
    cams = [c.to(dev) for c in cams[:n_views]]
    B, m = curves["curve_points"].shape[0], S.N_GAUSSIANS
    P = B * m
    H, W = cams[0].image_height, cams[0].image_width
    tanx, tany = math.tan(cams[0].FoVx * 0.5), math.tan(cams[0].FoVy * 0.5)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    cp0, w0, op0 = (curves[k].to(dev).contiguous() for k in ("curve_points", "width", "opacity"))
    isb = curve_sampling._bezier_mask(curves["is_bezier"].to(dev), dev)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device="cpu").manual_seed(103)
    dL = (torch.randn(1, H, W, generator=g) * 1e-3).to(dev)
    u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=dev)
    f32 = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
    cap = 1024
    d = dict(coef=curve_sampling.sample_coefficients(m, dev), norms=torch.zeros(384, dtype=torch.float64, device=dev),
             geom=u8(lib.cgs_geometry_bytes(P)), nbin=int(lib.cgs_binning_bytes(cap * tiles)), img=u8(lib.cgs_image_bytes(W, H)),
             color=f32(1, H, W), invd=f32(1, H, W), omap=f32(4, H, W), radii=torch.zeros(P, dtype=torch.int32, device=dev),
             g_m2d=f32(P, 3), scratch=f32(int(lib.cgs_view_backward_scratch_floats(B, m))))
    d["bin"] = u8(d["nbin"])
    flat = f32(B * 14)
    g_cp, g_w, g_op = flat[0:12 * B], flat[12 * B:13 * B], flat[13 * B:14 * B]
    pt, cf = L.ptr, C.c_float

    def view(cam):
        st = L.raw_stream(dev)
        L.check(lib.cgs_view_forward(B, m, pt(cp0), pt(w0), pt(isb), pt(d["coef"]), cf(1e-8), pt(d["norms"]), pt(op0), None, cf(0.01),
                                     None, pt(d["geom"]), pt(d["bin"]), d["nbin"], pt(d["img"]), cap, pt(bg), W, H,
                                     pt(cam.world_view_transform), pt(cam.full_proj_transform), pt(cam.camera_center), tanx, tany,
                                     pt(d["color"]), pt(d["invd"]), pt(d["omap"]), pt(d["radii"]), None, None, None, st),
                "cgs_view_forward")
        L.check(lib.cgs_view_backward(B, m, pt(cp0), pt(w0), pt(isb), pt(d["coef"]), cf(1e-8), pt(d["norms"]), pt(op0), None, cf(0.01),
                                      None, pt(d["geom"]), pt(d["bin"]), pt(d["img"]), pt(bg), W, H, pt(cam.world_view_transform),
                                      pt(cam.full_proj_transform), pt(cam.camera_center), tanx, tany, pt(d["radii"]), pt(dL), None,
                                      pt(d["g_m2d"]), pt(g_cp), pt(g_w), pt(g_op), None, pt(d["scratch"]), 1, st),
                "cgs_view_backward")

    for c in cams:   # warm-up
        view(c)
    torch.cuda.synchronize()
    lib.cgs_prof_reset()
    lib.cgs_prof_enable(1)
    reps = 3
    for _ in range(reps):
        for c in cams:
            view(c)
    torch.cuda.synchronize()
    lib.cgs_prof_enable(0)
    ks = {k: round(ms / max(n, 1) * 1e3, 2) for k, (ms, n) in L.prof_collect().items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c in cams:
            view(c)
    torch.cuda.synchronize()
    view_us = (time.perf_counter() - t0) / (reps * len(cams)) * 1e6
    print(json.dumps({"lib": os.path.basename(L.LIB_PATH), "config": cfg, "views": len(cams),
                      "kernel_us": dict(sorted(ks.items(), key=lambda kv: -kv[1])), "sum_us": round(sum(ks.values()), 1),
                      "view_us_wall": round(view_us, 1), "image_sum": float(d["color"].sum()), "grad_norm": float(flat.norm())}))


if __name__ == "__main__":
    main()
