"""Kernel times of the rasterizer instances the OPERATOR API reaches (GaussianRasterizer, the reference's own call shape,
diff_cur_rasterization/__init__.py:46-151), per BASELINE config, measured with the library's own HIP events (cgs_prof_*):

    python profiles/probes/general_instances.py [cfg3] [n_rep]

Cases (upstream gradients / what requires grad):
    reference_call    colours == 1 without grad, only dL/dcolour flowing in (gaussian_renderer/__init__.py:96-129) -> the gated
                      pair-major unit kernel
    training_general  the same call with the unit route switched off (cgs_set_operator_unit_route(0)) -> k_render_bwd3<0,0,0>
    colour_grad       arbitrary colours that require grad, dL/dcolour only                            -> k_render_bwd3<0,0,1>
    colour_allmap     ... + dL/dall_map                                                               -> k_render_bwd3<1,0,1>
    all_grad          ... + dL/dinvdepth + dL/dall_map                                                -> k_render_bwd3<1,1,1>
Prints one JSON object {case: {kernel: us}}; bench.py's `general_route` block calls time_instances() below."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

CASES = ("reference_call", "training_general", "colour_grad", "colour_allmap", "all_grad")


def config_splats(cfg, dev, view=0):
    """Splat tensors of one view of a BASELINE config, made by the product's own sampling / attribute kernels."""
    from curve_gaussian_amd import synthetic as S
    from curve_gaussian_amd.ops.curve_sampling import sample_curves, splat_attributes
    curves, cams = S.make_config(cfg, n_views=max(view + 1, 1))
    cam = cams[view].to(dev)
    c = {k: v.to(dev) for k, v in curves.items()}
    xyz, rot, scl = sample_curves(c["curve_points"], c["width"], c["is_bezier"], 12)
    rotn, opac, scales, amap = splat_attributes(rot, xyz, c["opacity"], scl, cam.camera_center, cam.world_view_transform, 12,
                                                None, 0.01)
    return dict(means3D=xyz.detach(), rotations=rotn.detach(), opacities=opac.detach(), scales=scales.detach(),
                all_map=amap.detach()), cam


def time_instances(cfg="cfg3", n_rep=6, cases=CASES, dev=None):
    import math

    from curve_gaussian_amd import _lib as L
    from curve_gaussian_amd.diff_cur_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = dev or torch.device("cuda:0")
    lib = L.load()
    sp, cam = config_splats(cfg, dev)
    P = sp["means3D"].shape[0]
    H, W = cam.image_height, cam.image_width
    rs = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center, prefiltered=False, debug=False,
        antialiasing=False, render_geo=True)
    g = torch.Generator().manual_seed(5)
    dcol = (torch.randn(1, H, W, generator=g) * 1e-3).to(dev)
    dinv = (torch.randn(1, H, W, generator=g) * 1e-3).to(dev)
    damap = (torch.randn(4, H, W, generator=g) * 1e-3).to(dev)
    rand_col = torch.rand(P, 1, generator=g).to(dev)
    out = {}
    for case in cases:
        unit = case in ("reference_call", "training_general")
        prev = lib.cgs_set_operator_unit_route(0 if case == "training_general" else 1)
        colors = torch.ones(P, 1, device=dev) if unit else rand_col.clone().requires_grad_(True)
        ins = {k: v.clone().requires_grad_(True) for k, v in sp.items()}
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        rast = GaussianRasterizer(rs)

        def once():
            color, radii, invd, amap = rast(means3D=ins["means3D"], means2D=m2d, opacities=ins["opacities"],
                                            colors_precomp=colors, scales=ins["scales"], rotations=ins["rotations"],
                                            all_map=ins["all_map"])
            loss = (color * dcol).sum()
            if case == "all_grad":
                loss = loss + (invd * dinv).sum()
            if case in ("all_grad", "colour_allmap"):
                loss = loss + (amap * damap).sum()
            loss.backward()

        for _ in range(2):
            once()
        torch.cuda.synchronize()
        lib.cgs_prof_reset()
        lib.cgs_prof_enable(1)
        for _ in range(n_rep):
            once()
        torch.cuda.synchronize()
        lib.cgs_prof_enable(0)
        prof = L.prof_collect()
        lib.cgs_prof_reset()
        lib.cgs_set_operator_unit_route(prev)
        out[case] = {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in sorted(prof.items())}
    stats = lib.cgs_last_forward_stats
    import ctypes as C
    R, longest, path = C.c_int64(), C.c_int64(), C.c_int()
    stats(C.byref(R), C.byref(longest), C.byref(path))
    out["_workload"] = {"config": cfg, "splats": P, "width": W, "height": H, "instances_R": int(R.value),
                        "binning_path": int(path.value)}
    return out


if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    print(json.dumps(time_instances(cfg, n_rep)))
