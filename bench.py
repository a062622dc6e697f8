#!/usr/bin/env python
"""bench.py -- headline benchmark of the curve-Gaussian hot path on MI355X.

Metric (BASELINE.json): Msplats rasterized/s, forward+backward, plus train-step ms and the HBM-roofline fraction.
A "step" is one gradient-exchange batch: `--views-per-step` (default 8) independent views per rank through the per-view
hot path, their curve-parameter gradients summed and (N > 1) all-reduced once -- the unit BASELINE's "train-step" implies.
The timed region is the K-step region repeated until it lasts >= --min-seconds (3 s; `repeats`, `views_timed` in the output; `steps`
and `warmup` are echoed unchanged), so the driver's small K still measures a steady state.  One view is:
    rasterizer forward  (preprocess -> tile binning + per-tile depth sort -> alpha-composite)
  + rasterizer backward (dL/d{mean2D,conic,opacity,colour,all_map} -> dL/d{mean3D,scale,rotation})
with inputs already resident in HBM (SURVEY.md section 8d "raster-only benchmark": dL_dcolor ~ N(0,1)*1e-3,
dL_dinvdepth = dL_dall_map = 0 as in training).  Default workload = BASELINE cfg3 (the configuration the north-star
target is quoted on): 16 667 curves x 12 = 200 004 splats, 1600x1600, Fibonacci-sphere cameras.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python bench.py --gpus N --steps K --warmup W        # launches N ranks itself (torch.distributed.run underneath)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W
`--gpus` must equal the number of ranks (WORLD_SIZE) and must not exceed the GPUs visible on the node: both are checked
and the run exits non-zero with a message instead of silently timing one rank.

Multi-GPU: views shard across ranks (rank r renders views r, r+N, ...), weak scaling (K views per rank); the
per-step exchange is ONE RCCL all-reduce of the flat curve-level gradient buffer.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); exported by the image,
# kept here for launches from a clean environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def algorithmic_bytes(P, R, H, W, passes=6):
    """compulsory traffic of the REFERENCE algorithm per view, fwd+bwd."""
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    return 480 * P + (228 + 24 * passes) * R + 64 * H * W + 32 * tiles


KERNEL_ALG_BYTES = {
    # per-launch algorithmic bytes of each kernel (same table, split per reference kernel)
    "render_bwd": lambda P, R, HW, T: 148 * R + 32 * HW + 8 * T,        # K8: 52 r + 96 rmw per instance, 32 B/px, 8 B/tile
    "render_fwd": lambda P, R, HW, T: 52 * R + 32 * HW + 8 * T,         # K6
    # the view path's forward instance sorts each tile's bucket itself: K4 (histogram + 6 radix passes) + K5 + K6 in one launch
    "render_fwd_sorting": lambda P, R, HW, T: (152 + 8 + 52) * R + 32 * HW + (8 + 16) * T,
    "preprocess_fwd": lambda P, R, HW, T: 104 * P,                      # K1 44 r + 60 w
    "preprocess_bwd": lambda P, R, HW, T: 228 * P,                      # K9 (60+36) + K10 (92+40)
    "tile_sort": lambda P, R, HW, T: (8 + 24 * 6) * R,                  # K4 histogram + 6 radix passes
    "scatter": lambda P, R, HW, T: 20 * P + 12 * R,                     # K3
    "scan_tiles": lambda P, R, HW, T: 8 * P + 16 * T,                   # K2 (+K5 range writes)
}


def committed_profile(config):
    """The newest profiles/rNN_traffic.json + rNN_pmc.csv pair whose `config` is `config` (files without the field are
    the round-1 cfg3 profiles), or None."""
    import glob
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True):
        try:
            tj = json.load(open(tf))
            if tj.get("config", "cfg3") != config:
                continue
            pmc = {}
            pf = tf.replace("_traffic.json", "_pmc.csv")
            if os.path.exists(pf):
                for line in open(pf):
                    f = line.strip().split(",")
                    if len(f) == 3 and not line.startswith("#") and f[0] != "Kernel":
                        pmc.setdefault(f[0], {})[f[1]] = float(f[2])
            return {"traffic": tj.get("kernels", {}), "pmc": pmc, "source": "profiles/" + os.path.basename(tf)}
        except Exception:
            continue
    return None


ISSUE_PEAK_G = 1024 * 2.4 / 2.0   # G wave-instructions/s: 1024 SIMDs, 2.4 GHz, 2 cycles per wave64 FMA-class instruction

# ------------------------------------------------------------------------------------------------ operator-API instances
# Kernel times of the rasterizer instances the OPERATOR API reaches (GaussianRasterizer, the reference's own call shape,
# diff_cur_rasterization/__init__.py:46-151), measured with the library's own HIP events (cgs_prof_*).  Cases (upstream
# gradients / what requires grad):
#   reference_call    colours == 1 without grad, only dL/dcolour flowing in (gaussian_renderer/__init__.py:96-129) -> the gated
#                     pair-major unit kernel
#   training_general  the same call with OPT_GENERAL_BACKWARD in its settings (per-call option)    -> k_render_bwd3<0,0,0>
#   colour_grad       arbitrary colours that require grad, dL/dcolour only                            -> k_render_bwd3<0,0,1>
#   colour_allmap     ... + dL/dall_map                                                               -> k_render_bwd3<1,0,1>
#   all_grad          ... + dL/dinvdepth + dL/dall_map                                                -> k_render_bwd3<1,1,1>
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
        from curve_gaussian_amd.diff_cur_rasterization import OPT_GENERAL_BACKWARD
        colors = torch.ones(P, 1, device=dev) if unit else rand_col.clone().requires_grad_(True)
        ins = {k: v.clone().requires_grad_(True) for k, v in sp.items()}
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        rast = GaussianRasterizer(rs._replace(options=OPT_GENERAL_BACKWARD) if case == "training_general" else rs)

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

        # (the first case also warms the clocks up: an idle GPU runs the first few dozen launches 10 % slower)
        for _ in range(30 if case == cases[0] else 3):
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
        out[case] = {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in sorted(prof.items())}
    stats = lib.cgs_last_forward_stats
    import ctypes as C
    R, longest, path = C.c_int64(), C.c_int64(), C.c_int()
    stats(C.byref(R), C.byref(longest), C.byref(path))
    out["_workload"] = {"config": cfg, "splats": P, "width": W, "height": H, "instances_R": int(R.value),
                        "binning_path": int(path.value)}
    return out



def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32,
                    help="timed steps; one step = one gradient-exchange batch of --views-per-step views per rank "
                         "(raster mode: one view)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="the K-step region is repeated (inside one barrier-bracketed timing) until it lasts this long")
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--mode", default="view", choices=["view", "raster"],
                    help="view: full per-view hot path (curve sampling -> splat attrs -> raster fwd+bwd -> curve grads "
                         "[-> RCCL all-reduce]); raster: rasterizer fwd+bwd only on precomputed splats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-times", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-general-route", action="store_true",
                    help="skip the `general_route` block (kernel times of the operator-API instances, GaussianRasterizer)")
    ap.add_argument("--no-realistic-size", action="store_true",
                    help="skip the `realistic_size` block of the default cfg3 line (one child run of cfg2, the size the reference trains)")
    ap.add_argument("--all-configs", action="store_true",
                    help="after the run, also run cfg1..cfg5 (one child process each, N = 1, short) and attach a compact "
                         "`all_configs` block: value, ms per view, whole-path roofline fraction, train step, drop-in view")
    ap.add_argument("--cpu-views", type=int, default=6)
    ap.add_argument("--torch-cpu-splats", type=int, default=48,
                    help="cfg1 only: splats of the bounded sample the pure-PyTorch CPU rasterizer (oracle/torch_ref.py) is timed on")
    ap.add_argument("--streams", type=int, default=3,
                    help="view mode: independent views kept in flight per GPU (HIP streams); 1 = strictly serial views")
    ap.add_argument("--no-graph", action="store_true",
                    help="view mode: launch every view eagerly (Python autograd + ctypes) instead of replaying one "
                         "captured hipGraph per stream")
    ap.add_argument("--train-step-multi", action="store_true",
                    help="N > 1 only: also time the view-parallel training iteration (GraphedTrainStep in collective mode: "
                         "every rank renders one view, ONE all-reduce of the flat gradients, identical Adam step on all "
                         "ranks).  Off by default: the scaling run measures the raster throughput only")
    ap.add_argument("--autograd-view", action="store_true",
                    help="view mode graphs: capture the drop-in Python API (sample_curves / splat_attributes / "
                         "rasterize_gaussians autograd Functions) instead of the autograd-free fused entry points")
    ap.add_argument("--no-pingpong", action="store_true",
                    help="view mode: one set of per-stream gradient buffers (the streams drain at every step boundary) "
                         "instead of two used by alternate steps")
    ap.add_argument("--views-per-step", type=int, default=8,
                    help="view mode: views per rank whose gradients are summed before the all-reduce (one optimizer "
                         "step's view batch per rank)")
    ap.add_argument("--fused-sort", type=int, default=-1, choices=[-1, 0, 1],
                    help="A/B: tile sort inside the forward compositor (sets CGS_FUSED_TILE_SORT, which the library reads once); "
                         "-1 = library default")
    return ap.parse_args(argv)


def start_rank(args):
    """Who this process is: (rank, world, device).  `python bench.py --gpus N` without a launcher becomes the launcher and
    never returns; inconsistent --gpus / WORLD_SIZE / visible GPUs exit non-zero with a message."""
    # Control-flow rehearsal of the multi-rank path on a box with ONE GPU (tests only): every rank uses device 0 and the
    # collectives go through gloo (RCCL refuses two ranks on one device).  Never set for a measurement.
    rehearsal = os.environ.get("CGS_BENCH_REHEARSAL") == "1"
    if args.gpus < 1:
        sys.exit(f"bench.py: --gpus {args.gpus}: need at least one GPU")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    n_dev = torch.cuda.device_count()
    if args.gpus > n_dev and not rehearsal:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible on this node: one rank per GPU, no "
                 f"oversubscription (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = "
                 f"{os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', 'unset'))})")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher -- one process per GPU over RCCL, exactly the
        # torch.distributed.run command line the docstring gives; the ranks print, this process only relays the exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not os.environ.get("CGS_BENCH_FORCE_DIST"):
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): the two must agree -- "
                 f"`n_gpus` in the output line is the real world size")
    torch.cuda.set_device(local_rank)
    return SimpleNamespace(rank=rank, world=world, dev=torch.device("cuda", local_rank), rehearsal=rehearsal)


class Workload:
    """The synthetic workload of one rank, resident in HBM before anything is timed: the curve tensors of the BASELINE config,
    their splats (raster mode and the instance counts), this rank's cameras with their direction maps, the upstream gradient
    and the flat curve-level gradient buffer the step exchanges.  Read-only after construction."""

    def __init__(self, args, who):
        from curve_gaussian_amd import _lib as L
        from curve_gaussian_amd import synthetic as S
        from curve_gaussian_amd.ops import curve_sampling
        if args.fused_sort >= 0:
            os.environ["CGS_FUSED_TILE_SORT"] = str(args.fused_sort)   # read once by the library, at its first forward
        self.args, self.rank, self.world, self.dev, self.rehearsal = args, who.rank, who.world, who.dev, who.rehearsal
        self.lib = L.load()
        dev = self.dev
        self.K, self.Wm = args.steps, args.warmup
        self.curves, cams = S.make_config(args.config)
        curves = self.curves
        self.B = curves["curve_points"].shape[0]
        self.m = S.N_GAUSSIANS
        self.xyz, rot, self.scl = curve_sampling.sample_curves(curves["curve_points"].to(dev), curves["width"].to(dev),
                                                               curves["is_bezier"].to(dev), self.m)
        self.P = P = self.xyz.shape[0]
        self.rotn = torch.nn.functional.normalize(rot).contiguous()
        self.opac = torch.sigmoid(curves["opacity"].to(dev)).unsqueeze(1).expand(-1, self.m, -1).reshape(-1, 1).contiguous()
        self.colors = torch.ones(P, 1, device=dev)
        self.H, self.W = cams[0].image_height, cams[0].image_width
        self.tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        self.bg = torch.zeros(3, device=dev)
        self.G = max(1, args.views_per_step) if args.mode == "view" else 1   # views per step (per rank)
        n_my = (self.K + self.Wm) * self.G
        dev_cams = {}      # one device copy per distinct view (the list below cycles through them)
        self.my_cams = []
        for i in range(n_my):
            vi = (self.rank + i * self.world) % len(cams)
            if vi not in dev_cams:
                dev_cams[vi] = cams[vi].to(dev)
            self.my_cams.append(dev_cams[vi])
        self.tanx, self.tany = math.tan(cams[0].FoVx * 0.5), math.tan(cams[0].FoVy * 0.5)
        # per-view direction map (gaussian_renderer/__init__.py:98-104) precomputed: raster-only benchmark
        Rm = curve_sampling.quaternion_to_matrix(self.rotn)[..., 0]
        self.amaps = {}
        for c in {id(c): c for c in self.my_cams}.values():
            neg = (Rm * (c.camera_center - self.xyz)).sum(-1) < 0
            d = torch.where(neg[:, None], -Rm, Rm) @ c.world_view_transform[:3, :3]
            self.amaps[id(c)] = torch.cat([d, torch.ones(P, 1, device=dev)], 1).contiguous()
        g = torch.Generator(device="cpu").manual_seed(103)
        self.dL_dcolor = (torch.randn(1, self.H, self.W, generator=g) * 1e-3).to(dev)
        self.empty = torch.empty(0, device=dev)
        # flat curve-level gradient buffer exchanged per step (38 floats / curve)
        self.flat_grads = torch.zeros(self.B * 38, device=dev)
        # "view" mode: the learnable curve tensors; their .grad lives inside ONE flat buffer (38 floats / curve:
        # curve_points 12 | width 1 | opacity 1 | mask 12 | features_dc 12), so the per-step exchange is a single
        # all-reduce with no packing kernels
        self.base = [curves["curve_points"].to(dev), curves["width"].to(dev), curves["opacity"].to(dev)]
        self.isb = curves["is_bezier"].to(dev)

    def raster_forward(self, cam, options=False):
        """The operator-level forward on the precomputed splats of this workload (raster mode, instance counts, parity)."""
        from curve_gaussian_amd.diff_cur_rasterization import _C
        return _C.rasterize_gaussians(
            self.bg, self.xyz, self.colors, self.opac, self.scl, self.rotn, 1.0, self.empty, self.amaps[id(cam)],
            cam.world_view_transform, cam.full_proj_transform, self.tanx, self.tany, self.H, self.W, self.empty, 0,
            cam.camera_center, False, False, True, options)

    def raster_backward(self, cam, radii, gB, R, bB, iB):
        from curve_gaussian_amd.diff_cur_rasterization import _C
        return _C.rasterize_gaussians_backward(
            self.bg, self.empty, self.xyz, radii, self.colors, self.amaps[id(cam)], self.opac, self.scl, self.rotn, 1.0,
            self.empty, cam.world_view_transform, cam.full_proj_transform, self.tanx, self.tany, self.dL_dcolor, self.empty,
            self.empty, self.empty, 0, cam.camera_center, gB, R, bB, iB, False, True, False)


class ViewPipeline:
    """The schedule the headline times, and nothing else: `views_per_step` independent views per step and rank, up to `streams` of
    them in flight (one captured hipGraph per stream, replayed after a 140-byte camera copy), their curve-parameter gradients
    summed into a flat buffer, ONE all-reduce per step over the ranks, double-buffered gradient sets so that the reduction of step
    s overlaps the views of step s + 1.  (--streams 1 --views-per-step 1 is the reference's one-view-per-iteration schedule.)

    The measurements taken AFTER the timed region (serial views, per-kernel times, the gradient check) ask `run_views` for another
    schedule through its keyword arguments; they never rebind the pipeline's own streams, step size or launch mode."""

    def __init__(self, w):
        import ctypes
        from curve_gaussian_amd.view_parallel import StaticCamera, ViewStreams, capture_graph
        self.w = w
        args, dev, lib = w.args, w.dev, w.lib
        self.G = w.G
        self.dist = None        # the process group is created after the hipGraph captures: no RCCL thread runs during capture
        self.exchange = True    # False for the rank-0-only sections after the timed region (per-kernel times): no collectives there
        self.stats = {"R": 0, "visible": 0}
        self.settings = {}
        self.vstreams = ViewStreams(args.streams if args.mode == "view" else 1, dev)
        # every stream accumulates into its own flat buffer through its own leaf aliases (a shared .grad would make
        # autograd funnel all accumulation through one stream and serialise the views); slot 0 is the exchanged buffer
        # Two such sets, used by alternate steps: the streams start step s+1 (other set) while the main stream is still
        # summing / all-reducing step s, so neither the per-step reduction nor the collective drains the view pipeline;
        # a set is reused at step s+2, ordered after its reduction by an event.  (--no-pingpong: one set, join per step.)
        self.n_sets = 2 if (args.mode == "view" and args.streams > 1 and not args.no_pingpong) else 1
        self.flat_sets = [[w.flat_grads if (q == 0 and i == 0) else torch.zeros_like(w.flat_grads)
                           for i in range(max(args.streams, 1))] for q in range(self.n_sets)]
        self.leaf_sets = [[self.make_leaves(f) for f in flats] for flats in self.flat_sets]
        self.reduced = [None] * self.n_sets      # event: this set's gradients have been reduced (it may be zeroed and refilled)
        self.step_no = 0
        self.view_seq = 0     # views dispatched so far (stream choice)

        # ---- hipGraph mode: one captured per-view pipeline per stream (sync-free forward with fixed-capacity buckets),
        # replayed for any view after a 140-byte camera copy; the host cost per view drops from ~0.4 ms to ~0.02 ms
        self.view_graphs = None
        self.direct_bufs = []
        self.overflow_acc = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cap = 0
        self.eager_direct = None   # [set][stream] -> body: the fused direct body launched eagerly (--no-graph without --autograd-view)
        self.packs = {}
        longest = 1
        if args.mode == "view":
            with torch.no_grad():   # longest tile list over this rank's views (eager, exact path) -> bucket capacity
                for cam in {id(c): c for c in w.my_cams}.values():
                    self.step_raster(cam)
                    mlen = ctypes.c_int64()
                    lib.cgs_last_forward_stats(None, ctypes.byref(mlen), None)
                    longest = max(longest, int(mlen.value))
            self.cap = (int(longest * 1.5) + 64 + 63) // 64 * 64
            if self.cap > int(lib.cgs_bucket_capacity_limit()):
                self.cap = 0
        cap = self.cap
        if args.mode == "view" and args.no_graph and not args.autograd_view and cap:
            self.eager_direct = [[self.make_direct_view(self.flat_sets[q][si], cap) for si in range(self.vstreams.n)]
                                 for q in range(self.n_sets)]
            self.direct_bufs.extend(d for row in self.eager_direct for _, d in row)
        if args.mode == "view" and not args.no_graph:
            try:
                if not cap:
                    raise RuntimeError(f'tile lists of {longest} entries exceed the bucket limit')
                self.packs = {id(c): StaticCamera.packed(c) for c in w.my_cams}
                view_graphs = []      # [set][stream] -> (graph, its static camera, its stream)
                for q in range(self.n_sets):
                    view_graphs.append([])
                    for si in range(self.vstreams.n):
                        scam = StaticCamera(w.my_cams[0], dev)
                        scam.load(self.packs[id(w.my_cams[0])])
                        if args.autograd_view:      # the drop-in Python API inside the graph (3 autograd Functions per view)
                            sink = []

                            def body(scam=scam, leaves=self.leaf_sets[q][si], sink=sink):
                                del sink[:]
                                self.step_view(scam, leaves, False, cap, sink)
                                self.overflow_acc.add_(sink[0][2:3])   # sticky bucket-overflow flag, checked after the timed region
                        else:
                            direct, dbufs = self.make_direct_view(self.flat_sets[q][si], cap)
                            self.direct_bufs.append(dbufs)

                            def body(scam=scam, direct=direct):
                                direct(scam)

                        st = self.vstreams.streams[si] if self.vstreams.streams else torch.cuda.Stream()
                        graph, _ = capture_graph(body, st)
                        view_graphs[q].append((graph, scam, st))
                self.zero_gradients()
                self.overflow_acc.zero_()
                for dbufs in self.direct_bufs:
                    dbufs["sticky"].zero_()
                torch.cuda.synchronize()
                self.view_graphs = view_graphs
            except Exception as e:   # capture is an optimisation: fall back to eager launches
                print(f"bench: hipGraph capture unavailable ({e}); using eager launches", file=sys.stderr)
                self.view_graphs = None
        # per-kernel timing of the headline (fused direct) body: its own buffers, serial eager launches
        self.prof_direct = [None, None]
        if cap and not args.autograd_view and args.mode == "view":
            self.prof_direct = list(self.make_direct_view(torch.zeros_like(w.flat_grads), cap))
        self.use_graphs = self.view_graphs is not None

    # ------------------------------------------------------------------------------------------------ per-view bodies
    def step_raster(self, cam, collect=False):
        """raster mode: the operator-level forward + backward on the precomputed splats."""
        w = self.w
        (R, color, radii, gB, bB, iB, invd, om) = w.raster_forward(cam)
        grads = w.raster_backward(cam, radii, gB, R, bB, iB)
        if self.dist is not None and self.exchange:
            self.dist.all_reduce(w.flat_grads)
        if collect:
            self.stats["R"] += R
            self.stats["visible"] += int((radii > 0).sum())
        return grads

    def make_leaves(self, flat):
        # aliases of the same parameter storage with their own autograd identity and their own flat .grad buffer
        B = self.w.B
        leaves = [t.detach().requires_grad_(True) for t in self.w.base]
        leaves[0].grad = flat[0:12 * B].view(B, 4, 3)
        leaves[1].grad = flat[12 * B:13 * B].view(B, 1)
        leaves[2].grad = flat[13 * B:14 * B].view(B, 1)
        return leaves

    def step_view(self, cam, leaves, collect=False, static_cap=0, sink=None):
        """One view through the drop-in Python API (three autograd Functions)."""
        from curve_gaussian_amd.diff_cur_rasterization import GaussianRasterizationSettings, rasterize_gaussians
        from curve_gaussian_amd.ops import curve_sampling
        w = self.w
        p_cp, p_w, p_op = leaves
        s_xyz, s_rot, s_scl = curve_sampling.sample_curves(p_cp, p_w, w.isb, w.m)            # prepare_scaling_rot
        rot_n, opacity, scales, amap = curve_sampling.splat_attributes(
            s_rot, s_xyz, p_op, s_scl, cam.camera_center, cam.world_view_transform, w.m)   # render() glue, fused
        rs = self.settings.get((id(cam), static_cap))
        if rs is None:
            rs = self.settings[(id(cam), static_cap)] = GaussianRasterizationSettings(
                image_height=w.H, image_width=w.W, tanfovx=w.tanx, tanfovy=w.tany, bg=w.bg, scale_modifier=1.0,
                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0,
                campos=cam.camera_center, prefiltered=False, debug=False, antialiasing=False, render_geo=True,
                static_bucket_cap=static_cap, status_sink=sink)
        color, radii, invd, om = rasterize_gaussians(s_xyz, None, w.empty, w.colors, opacity, scales, rot_n, w.empty, amap, rs)
        color.backward(w.dL_dcolor)   # synthetic upstream gradient; grads accumulate into the flat buffer's views
        if collect:
            self.stats["visible"] += int((radii > 0).sum())

    def make_direct_view(self, flat, cap):
        """The autograd-free per-view body on the fused entry points (cgs_view_forward / cgs_view_backward): what the hipGraph
        of a stream replays -- 8 library kernels per view, no torch kernels; the curve-parameter gradients are ADDED into the
        stream's flat buffer by the backward itself (accumulate = 1), bucket overflows are counted in a sticky status word.
        -> (body(cam, shared=False), its buffers)"""
        import ctypes as C
        from curve_gaussian_amd import _lib as L
        from curve_gaussian_amd.ops import curve_sampling
        w = self.w
        lib, dev, B, m, P, W, H, bg, tanx, tany, dL_dcolor = w.lib, w.dev, w.B, w.m, w.P, w.W, w.H, w.bg, w.tanx, w.tany, w.dL_dcolor
        u8 = lambda n: torch.zeros(int(n), dtype=torch.uint8, device=dev)
        f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
        d = dict(coef=curve_sampling.sample_coefficients(m, dev), norms=torch.empty(384, dtype=torch.float64, device=dev),
                 geom=u8(lib.cgs_geometry_bytes(P)), nbin=int(lib.cgs_binning_bytes(cap * w.tiles)), img=u8(lib.cgs_image_bytes(W, H)),
                 color=f32(1, H, W), invd=f32(1, H, W), omap=f32(4, H, W), radii=torch.empty(P, dtype=torch.int32, device=dev),
                 g_m2d=f32(P, 3), scratch=f32(int(lib.cgs_view_backward_scratch_floats(B, m))))
        d["bin"] = u8(d["nbin"])
        off = int(lib.cgs_image_status_offset(W, H)) + 4 * int(lib.cgs_status_words())
        d["sticky"] = d["img"][off:off + 4].view(torch.int32)
        g_cp, g_w, g_op = flat[0:12 * B], flat[12 * B:13 * B], flat[13 * B:14 * B]
        cp0, w0, op0 = w.base
        isb_u8 = curve_sampling._bezier_mask(w.isb, dev)
        pt, cf = L.ptr, C.c_float

        def body(cam, shared=False):   # shared: the view-batch mode (cgs_view_forward_shared / CGS_VIEW_SHARED), per call
            st = L.raw_stream(dev)
            fwd = lib.cgs_view_forward_shared if shared else lib.cgs_view_forward
            L.check(fwd(B, m, pt(cp0), pt(w0), pt(isb_u8), pt(d["coef"]), cf(1e-8), pt(d["norms"]), pt(op0), None,
                        cf(0.01), None, pt(d["geom"]), pt(d["bin"]), d["nbin"], pt(d["img"]), cap, pt(bg), W, H,
                        pt(cam.world_view_transform), pt(cam.full_proj_transform), pt(cam.camera_center),
                        tanx, tany, pt(d["color"]), pt(d["invd"]), pt(d["omap"]), pt(d["radii"]), None, None,
                        None, st), "cgs_view_forward")
            L.check(lib.cgs_view_backward(B, m, pt(cp0), pt(w0), pt(isb_u8), pt(d["coef"]), cf(1e-8), pt(d["norms"]), pt(op0), None,
                                          cf(0.01), None, pt(d["geom"]), pt(d["bin"]), pt(d["img"]), pt(bg), W, H,
                                          pt(cam.world_view_transform), pt(cam.full_proj_transform), pt(cam.camera_center),
                                          tanx, tany, pt(d["radii"]), pt(dL_dcolor), None, pt(d["g_m2d"]), pt(g_cp), pt(g_w),
                                          pt(g_op), None, pt(d["scratch"]), 3 if shared else 1, st), "cgs_view_backward")
        return body, d

    # ------------------------------------------------------------------------------------------------ the schedule
    def _replay_view(self, vs, j, cam, q=0):
        graph, scam, st = self.view_graphs[q][j % min(len(self.view_graphs[q]), vs.n)]
        if vs.streams:
            with torch.cuda.stream(st):
                scam.load(self.packs[id(cam)])
                graph.replay()
        else:   # single stream: replay on the caller's stream
            scam.load(self.packs[id(cam)])
            graph.replay()

    def run_views(self, view_list, collect=False, *, serial=False, views_per_step=None, graphs=None):
        """Runs the views of `view_list`, `views_per_step` per step, on the pipeline's schedule.  The keyword arguments select another
        schedule for ONE call without touching the pipeline's own: serial = one view in flight on the caller's stream;
        graphs = False forces eager launches (True: replay if graphs were captured)."""
        from curve_gaussian_amd.view_parallel import ViewStreams
        w, args = self.w, self.w.args
        vs = ViewStreams(1) if serial else self.vstreams
        G = self.G if views_per_step is None else views_per_step
        use_graphs = self.use_graphs if graphs is None else (graphs and self.view_graphs is not None)
        n_sets, flat_sets, leaf_sets, reduced = self.n_sets, self.flat_sets, self.leaf_sets, self.reduced
        for g0 in range(0, len(view_list), G):
            if args.mode == "raster":
                self.step_raster(view_list[g0], collect)
                continue
            q = self.step_no % n_sets if vs.streams else 0
            self.step_no += 1
            flats, leaves = flat_sets[q], leaf_sets[q]
            cur = torch.cuda.current_stream()
            if n_sets == 1 or not vs.streams:
                for f in flats[:vs.n]:
                    f.zero_()
                vs.fork()
            else:   # each stream clears its own buffer of this set, once the set's previous use has been reduced
                for i, st in enumerate(vs.streams):
                    if reduced[q] is not None:
                        st.wait_event(reduced[q])
                    else:
                        st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        flats[i].zero_()
            for cam in view_list[g0:g0 + G]:
                # the view's stream: round-robin over ALL views, not restarted per step -- with 8 views per step on 3 streams
                # a per-step restart hands streams 0 / 1 three views and stream 2 two, every step (the streams run on across
                # step boundaries, so one of them idles a third of the time)
                j = self.view_seq % vs.n
                self.view_seq += 1
                if use_graphs and not collect:
                    self._replay_view(vs, j, cam, q)
                elif collect and self.prof_direct[0] is not None:   # per-kernel timing of the headline (fused direct) body
                    self.prof_direct[0](cam)
                    self.stats["visible"] += int((self.prof_direct[1]["radii"] > 0).sum())
                elif self.eager_direct is not None and not collect:
                    vs.run(j, self.eager_direct[q][j % vs.n][0], cam)
                else:
                    vs.run(j, self.step_view, cam, leaves[j % vs.n], collect)
            vs.join()     # (events only: the main stream waits, the view streams run on into the next step)
            for f in flats[1:vs.n]:
                flats[0].add_(f)
            if self.dist is not None and self.exchange:
                self.dist.all_reduce(flats[0])
            if n_sets > 1 and vs.streams:
                reduced[q] = torch.cuda.Event()
                reduced[q].record(cur)

    def last_step_gradient(self, serial=False):
        """The flat gradient buffer the most recent step reduced into."""
        return self.flat_sets[(self.step_no - 1) % self.n_sets if (self.vstreams.streams and not serial) else 0][0]

    def zero_gradients(self):
        for flats in self.flat_sets:
            for f in flats:
                f.zero_()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def form_process_group(self):
        """One rank per GPU over RCCL (`nccl`); the rehearsal of the tests uses gloo on one device."""
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if self.w.rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=self.w.dev)
        self.dist = dist
        # graph replay only if EVERY rank captured its graphs: the schedules below contain collectives, and ranks on
        # different schedules would wait for each other forever
        flag = torch.tensor([1 if self.use_graphs else 0], device=self.w.dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self.use_graphs = bool(int(flag.item()))


# ------------------------------------------------------------------------------------------------ the timed region
def time_headline(pipe):
    """THE measurement: Wm untimed warm-up steps, then exactly K steps (repeated back to back until the region lasts --min-seconds)
    between barrier + synchronize brackets, max over ranks.  Everything else bench.py reports is measured after this returns.
    -> {"elapsed": seconds of the timed region (max over ranks), "reps": K-step regions inside it, "own_n1", "rank_diag"}"""
    w = pipe.w
    args, dev, K, Wm, G, my_cams = w.args, w.dev, w.K, w.Wm, pipe.G, w.my_cams
    distributed = w.world > 1 or bool(os.environ.get("CGS_BENCH_FORCE_DIST"))   # (the env switch exercises RCCL with a single rank)
    own_n1 = None   # this rank's single-GPU rate on the same workload, measured BEFORE the process group exists
    if distributed and args.mode == "view":
        # Self-diagnosis of a multi-GPU run: the same K-step region with no collective and no other rank
        # in the schedule -- what this process does alone, in this process, on this GPU.  efficiency_vs_own_n1 = per-rank
        # rate of the timed run / this; a low value with a small all_reduce_ms points at the host (N ranks' Python on one
        # node), a large all_reduce_ms at the wire / RCCL channels.
        pipe.run_views(my_cams[:Wm * G])
        torch.cuda.synchronize()
        t_own, n_own = 0.0, 0
        while t_own < min(1.0, args.min_seconds) or n_own == 0:
            t0 = time.perf_counter()
            pipe.run_views(my_cams[Wm * G:(Wm + K) * G])
            torch.cuda.synchronize()
            t_own += time.perf_counter() - t0
            n_own += 1
            if n_own >= 10000:
                break
        own_n1 = {"ms_per_step": t_own / (n_own * K) * 1e3, "msplats_per_s": w.P * G * K * n_own / t_own / 1e6}
        pipe.zero_gradients()
    if distributed:
        pipe.form_process_group()
    dist = pipe.dist

    def timed(reps):
        """Wm warm-up steps, then the K-step region `reps` times back to back inside ONE barrier + synchronize bracket."""
        pipe.run_views(my_cams[:Wm * G])
        pipe.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            pipe.run_views(my_cams[Wm * G:(Wm + K) * G])
        pipe.barrier()
        return time.perf_counter() - t0

    def timed_long():
        """A pilot K-step region sizes the repeat count (max over ranks, so every rank repeats equally), then the
        measurement proper."""
        pilot = timed(1)
        if dist is not None:
            tp = torch.tensor([pilot], device=dev, dtype=torch.float64)
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
            pilot = float(tp.item())
        reps = max(1, min(10000, int(math.ceil(1.1 * args.min_seconds / max(pilot, 1e-6)))))   # (the pilot runs colder, hence slower)
        return (timed(reps) if reps > 1 else pilot), reps

    elapsed, reps = timed_long()
    for dbufs in pipe.direct_bufs:   # sticky overflow counters of the direct bodies
        pipe.overflow_acc.add_(dbufs["sticky"])
    if dist is not None:   # every rank takes the same decision (the re-timing below contains collectives)
        dist.all_reduce(pipe.overflow_acc, op=dist.ReduceOp.MAX)
    if int(pipe.overflow_acc.item()) != 0:   # a tile list outgrew its bucket on some rank: graph results invalid
        print("bench: bucket overflow in graph mode, re-timing with eager launches", file=sys.stderr)
        pipe.use_graphs = False
        elapsed, reps = timed_long()
    rank_diag = None
    if dist is not None:
        # per-rank times of the timed region (the headline takes the MAX), each rank's own N = 1 rate, and the all-reduce of the
        # step's flat gradient buffer timed ALONE with events on the stream it is issued on
        mine = torch.tensor([elapsed, own_n1["ms_per_step"] if own_n1 else 0.0, own_n1["msplats_per_s"] if own_n1 else 0.0],
                            device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank = [float(t[0].item()) for t in every]
        ar_ms = None
        if args.mode == "view":
            buf = pipe.flat_sets[0][0]
            for _ in range(3):
                dist.all_reduce(buf)
            pipe.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_ar = 20
            e0.record()
            for _ in range(n_ar):
                dist.all_reduce(buf)
            e1.record()
            e1.synchronize()
            ar = torch.tensor([e0.elapsed_time(e1) / n_ar], device=dev, dtype=torch.float64)
            dist.all_reduce(ar, op=dist.ReduceOp.MAX)
            ar_ms = float(ar.item())
            buf.zero_()
        rank_diag = {"per_rank_elapsed_s": per_rank, "own_n1": [(float(t[1].item()), float(t[2].item())) for t in every],
                     "all_reduce_ms": ar_ms}
        elapsed = max(per_rank)
    return {"elapsed": elapsed, "reps": reps, "own_n1": own_n1, "rank_diag": rank_diag}


# ------------------------------------------------------------------------------------------------ after the timed region
def check_step_gradient(pipe):
    """The step's summed gradient under the timed schedule (streams, graphs, double-buffered sets) against the same views run one at
    a time, eagerly, on one stream: the overlap machinery must not change what is computed.  -> relative L2 error or None"""
    w, G = pipe.w, pipe.G
    if not (w.args.mode == "view" and (pipe.vstreams.n > 1 or pipe.use_graphs) and (pipe.dist is None or w.world == 1)):
        return None
    pipe.barrier()
    for _ in range(3):     # several consecutive steps: both sets, and a reuse of the first
        pipe.run_views(w.my_cams[:G])
    pipe.barrier()
    got = pipe.last_step_gradient().clone()
    pipe.run_views(w.my_cams[:G], serial=True, graphs=False)
    pipe.barrier()
    ref = pipe.last_step_gradient(serial=True).clone()
    grad_check = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
    if not grad_check < 1e-3 and os.environ.get("CGS_BENCH_WHATIF") != "1":   # (what-if timing builds compute garbage)
        raise RuntimeError(f"bench: overlapped schedule changed the step gradient (relative L2 error {grad_check:.3e})")
    return grad_check


def time_serial_views(pipe):
    """The reference's schedule for comparison: one view at a time, one stream (latency of a single view's hot path), eager and as
    one hipGraph replay per view.  -> (eager ms per view, graph ms per view), None where not measured"""
    w = pipe.w
    if not (w.args.mode == "view" and (pipe.vstreams.n > 1 or pipe.G > 1)):
        return None, None
    serial = {}
    for name, ug in (("eager", False), ("graph", pipe.use_graphs)):
        if name == "graph" and not ug:
            continue
        sv = w.my_cams[:min(len(w.my_cams), 64)]
        pipe.run_views(sv[:4], serial=True, views_per_step=1, graphs=ug)
        pipe.barrier()
        ts0 = time.perf_counter()
        pipe.run_views(sv, serial=True, views_per_step=1, graphs=ug)
        pipe.barrier()
        serial[name] = (time.perf_counter() - ts0) / len(sv) * 1e3
    return serial["eager"], serial.get("graph")


def time_view_parallel_train_step(pipe):
    """--train-step-multi: the view-parallel training iteration (all ranks; world == 1: CGS_BENCH_FORCE_DIST, tests)."""
    w, dist = pipe.w, pipe.dist
    if not (dist is not None and w.args.train_step_multi and w.args.mode == "view"):
        return None
    from curve_gaussian_amd.scene import GaussianCurveModel
    from curve_gaussian_amd.train_step import GraphedTrainStep
    curves, dev, H, W = w.curves, w.dev, w.H, w.W
    gmv = GaussianCurveModel(0, w.m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                    curves["opacity"], curves["mask"],
                                                                    curves["is_bezier"])
    vcams = w.my_cams[:8]
    ggv = torch.Generator(device="cpu").manual_seed(7 + w.rank)
    vgts = [((torch.rand(1, H, W, generator=ggv) > 0.97).float() * torch.rand(1, H, W, generator=ggv)).to(dev) for _ in vcams]
    tsv = GraphedTrainStep(gmv, vcams, vgts, rank=w.rank, world=w.world)
    for _ in range(3):
        tsv.step()
    tsv.finish()
    pipe.barrier()
    tv0 = time.perf_counter()
    n_vp = 32
    for _ in range(n_vp):
        tsv.step()
    tsv.finish()
    pipe.barrier()
    tvt = torch.tensor([(time.perf_counter() - tv0) / n_vp * 1e3], device=dev, dtype=torch.float64)
    dist.all_reduce(tvt, op=dist.ReduceOp.MAX)
    del tsv, gmv
    return float(tvt.item())


def time_kernels(pipe):
    """Per-kernel times (HIP events on the launch stream, serial eager launches; rank 0) and the instance counts of the same views.
    -> (kernel_ms, mean instances per view, mean visible splats per view)"""
    w, lib, stats = pipe.w, pipe.w.lib, pipe.stats
    if w.args.no_kernel_times or w.rank != 0:
        return {}, 0.0, 0.0
    from curve_gaussian_amd import _lib as L
    kernel_ms = {}
    pipe.exchange = False     # the other ranks are past their last collective
    lib.cgs_prof_reset()
    lib.cgs_prof_enable(1)
    # the same views whatever --steps / --warmup say (the first 18 of this rank: what profiles/collect.sh traces), so
    # the live per-kernel times are comparable from run to run and with the committed rocprofv3 summary
    prof_cams = w.my_cams[:min(len(w.my_cams), 18)]
    n_prof = len(prof_cams)
    pipe.run_views(prof_cams, collect=True, serial=True)   # serial views: per-kernel event times must not overlap
    torch.cuda.synchronize()
    lib.cgs_prof_enable(0)
    for name, (ms, n) in L.prof_collect().items():
        kernel_ms[name] = ms / max(n, 1)
    if w.args.mode == "view":  # instance counts come from the raster-only call on the same views
        for i in range(n_prof):
            (R_i, *_rest) = w.raster_forward(prof_cams[i])
            stats["R"] += R_i
        # ... and the REFERENCE algorithm's instance count for the same views: every tile of the 3-sigma rect, no
        # alpha >= 1/255 tile culling (SURVEY 8d's byte formula is the reference's compulsory traffic)
        from curve_gaussian_amd.diff_cur_rasterization import OPT_NO_TILE_CULLING
        for i in range(min(n_prof, 6)):
            (R_i, *_rest) = w.raster_forward(prof_cams[i], OPT_NO_TILE_CULLING)
            stats["R_ref"] = stats.get("R_ref", 0) + R_i
        stats["R_ref"] /= min(n_prof, 6)
    return kernel_ms, stats["R"] / n_prof, stats["visible"] / n_prof


def time_shared_sampling(pipe):
    """The same per-view body when the curve sampling is SHARED by the views of a step (cgs_view_forward_shared): the parameters are
    constant inside a step, so the norm pass of the forward and the last pass of the sampling backward can run once per step.
    Reported separately -- NOT `value`: the reference's iteration is one view per parameter state."""
    import ctypes as C
    from curve_gaussian_amd import _lib as L
    from curve_gaussian_amd.ops import curve_sampling
    w = pipe.w
    lib, dev, B, m = w.lib, w.dev, w.B, w.m
    flat_a = torch.zeros_like(w.flat_grads)
    body_s, d_s = pipe.make_direct_view(flat_a, pipe.cap)
    cp0, w0, _op0 = w.base
    isb_u8 = curve_sampling._bezier_mask(w.isb, dev)
    Gs = max(1, w.args.views_per_step)
    cams_s = w.my_cams[:Gs]

    def step_shared(flat, on):
        st = L.raw_stream(dev)
        if on:
            L.check(lib.cgs_view_shared_begin(B, m, L.ptr(cp0), L.ptr(isb_u8), L.ptr(d_s["coef"]), L.ptr(d_s["norms"]),
                                              L.ptr(d_s["scratch"]), st), "cgs_view_shared_begin")
        for c in cams_s:
            body_s(c, on)
        if on:
            L.check(lib.cgs_view_shared_end(B, m, L.ptr(cp0), L.ptr(w0), L.ptr(isb_u8), L.ptr(d_s["coef"]), C.c_float(1e-8),
                                            L.ptr(d_s["norms"]), L.ptr(d_s["scratch"]), L.ptr(flat[0:12 * B]),
                                            L.ptr(flat[12 * B:13 * B]), 1, st), "cgs_view_shared_end")
    flat_a.zero_(); step_shared(flat_a, False); torch.cuda.synchronize(); ref_s = flat_a.clone()
    flat_a.zero_(); step_shared(flat_a, True); torch.cuda.synchronize()
    rel_s = float((flat_a - ref_s).norm() / ref_s.norm().clamp_min(1e-30))
    lib.cgs_prof_reset(); lib.cgs_prof_enable(1)
    for _ in range(2):
        step_shared(flat_a, True)
    torch.cuda.synchronize(); lib.cgs_prof_enable(0)
    ks = {name: ms / (2 * Gs) for name, (ms, n) in L.prof_collect().items()}   # per VIEW (step kernels amortised)
    shared = {"views_per_step": Gs, "kernel_ms_per_view": {k: round(v, 5) for k, v in sorted(ks.items(), key=lambda kv: -kv[1])},
              "sum_kernel_ms": round(sum(ks.values()), 5),
              "non_compositor_kernel_ms": round(sum(v for k, v in ks.items() if k not in ("render_fwd", "render_bwd")), 5),
              "step_gradient_rel_l2_vs_per_view_sampling": float(f"{rel_s:.3e}"),
              "note": "serial eager launches, HIP events per kernel; k_sample_f12 and k_sample_bwd<3> run once per step of "
                      f"{Gs} views (their time is divided by {Gs}); NOT part of value"}
    if not rel_s < 1e-3:
        raise RuntimeError(f"bench: shared sampling changed the step gradient (relative L2 {rel_s:.3e})")
    return shared


# ------------------------------------------------------------------------------------------------ the output line
def headline_fields(pipe, head, R_mean, vis_mean):
    """metric / value / config of the contract, from the timed region alone."""
    w, args = pipe.w, pipe.w.args
    K, Wm, G, world, P, B, m, W, H = w.K, w.Wm, pipe.G, w.world, w.P, w.B, w.m, w.W, w.H
    elapsed, reps = head["elapsed"], head["reps"]
    steps_timed = K * reps
    views_timed = steps_timed * G            # per rank
    ms_per_step = elapsed / steps_timed * 1e3
    ms_per_view = elapsed / views_timed * 1e3   # per rank
    value = P * world * views_timed / elapsed / 1e6
    out = {
        "metric": "Msplats rasterized/s (fwd+bwd)", "value": round(value, 3), "unit": "Msplats/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms_per_step, 4),
        "repeats": reps, "views_timed": views_timed * world, "timed_seconds": round(elapsed, 4),
        "ms_per_view": round(ms_per_view, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: synthetic curve-Gaussians, {B} curves x {m} = {P} splats, "
                               f"{W}x{H}, " + ("curve sampling + splat attrs + raster fwd+bwd + curve-param grads per view" if args.mode == "view" else "raster fwd+bwd per view"),
                   "mode": args.mode,
                   "splats": P, "curves": B, "width": W, "height": H, "tiles": w.tiles,
                   "instances_per_view_R": round(R_mean, 1), "visible_per_view": round(vis_mean, 1),
                   "step": f"{G} view(s) per rank, gradients summed" + (", one RCCL all-reduce" if world > 1 else ""),
                   "views_per_rank": views_timed, "views_per_step_per_rank": G, "views_in_flight_per_gpu": pipe.vstreams.n,
                   "launch": ("hipGraph replay per view" if pipe.use_graphs else "eager launches") + " (" + ("autograd body: drop-in Python API" if args.autograd_view or (not pipe.use_graphs and pipe.eager_direct is None) else "fused direct body: cgs_view_forward / cgs_view_backward") + ")",
                   "step_boundary": "double-buffered gradient sets (reduction/all-reduce of step s overlaps the views of "
                                    "step s+1)" if pipe.n_sets > 1 else "join per step",
                   "parallelism": f"view-parallel x{world}"},
    }
    # how many ranks the step's collective really spanned (1: no process group, nothing was exchanged) and how the step
    # boundary is scheduled -- top-level, whatever N, so a scaling record can be read without the config block
    out["rccl_ranks"] = pipe.dist.get_world_size() if pipe.dist is not None else 1
    assert out["rccl_ranks"] == world == args.gpus or os.environ.get("CGS_BENCH_FORCE_DIST"), \
        f"--gpus {args.gpus}, WORLD_SIZE {world}, collective spans {out['rccl_ranks']} rank(s)"
    out["step_boundary"] = out["config"]["step_boundary"]
    return out, ms_per_view


def scaling_fields(out, pipe, head):
    """N > 1: what the run should show (prediction) and what it showed (per-rank times, all-reduce alone, own N = 1 rates)."""
    w = pipe.w
    world, B, K = w.world, w.B, w.K
    rank_diag, own_n1 = head["rank_diag"], head["own_n1"]
    if world > 1:
        # what this run should show, machine-readable: per-rank step = G views at the 1-GPU per-view
        # time, ONE all-reduce of 38 floats per curve off the critical path (double-buffered gradient sets); ring wire time
        # 2 (N-1)/N S over one xGMI link direction (~64 GB/s) + per-hop latency; efficiency >= 0.95 at N = 8
        S_bytes = 38 * 4 * B
        wire_ms = 2.0 * (world - 1) / world * S_bytes / 64e9 * 1e3
        out["expected_scaling"] = {
            "all_reduce_bytes": S_bytes, "all_reduce_wire_ms_ring_one_link": round(wire_ms, 4),
            "all_reduce_latency_ms": round(2 * (world - 1) * 0.008, 3),
            "overlapped_with_next_step": pipe.n_sets > 1,
            "min_efficiency_vs_1gpu": {2: 0.97, 4: 0.96, 8: 0.95}.get(world, 0.95),
            "reference_predictions": {"cfg3": {"step_ms": 2.0, "all_reduce_bytes": 2533000},
                                      "cfg5": {"step_ms": 5.6, "all_reduce_bytes": 12667000}},
            "if_below": "check step_boundary == double-buffered (else the all-reduce serialises with the views); then "
                        "NCCL_MAX_NCHANNELS=4 (RCCL channels starving the compositors of CUs)"}
    if rank_diag is not None:
        # measured, beside the prediction above: a bad first multi-GPU run says WHERE it lost
        n_steps = K * head["reps"]
        per = [t / n_steps * 1e3 for t in rank_diag["per_rank_elapsed_s"]]
        slow = max(range(len(per)), key=per.__getitem__)
        out["all_reduce_ms"] = None if rank_diag["all_reduce_ms"] is None else round(rank_diag["all_reduce_ms"], 4)
        out["all_reduce_note"] = ("one all-reduce of the flat gradient buffer (38 floats / curve) timed alone, max over ranks; "
                                  "in the timed schedule it overlaps the next step's views")
        out["per_rank_ms_per_step"] = {"min": round(min(per), 4), "max": round(max(per), 4), "slowest_rank": slow,
                                       "all": [round(x, 4) for x in per]}
        if own_n1 is not None:
            own = rank_diag["own_n1"]
            out["own_n1"] = {"ms_per_step_rank0": round(own[0][0], 4), "msplats_per_s_rank0": round(own[0][1], 1),
                             "msplats_per_s_min_over_ranks": round(min(o[1] for o in own), 1),
                             "msplats_per_s_max_over_ranks": round(max(o[1] for o in own), 1),
                             "note": "the same K-step region on this GPU before the process group formed: no collective, "
                                     "no barrier; every rank measures its own while the others measure theirs"}
            # whole-job rate / (N x the mean of the ranks' own single-GPU rates)
            out["efficiency_vs_own_n1"] = round(out["value"] / max(sum(o[1] for o in own), 1e-9), 4)


def roofline_fields(out, pipe, kernel_ms, shared, R_mean, ms_per_view):
    """`roofline` of the dominant kernel (live time, committed PMC traffic), the issue rooflines, per-kernel times, `whole_path`."""
    w, args, stats = pipe.w, pipe.w.args, pipe.stats
    P, H, W, tiles = w.P, w.H, w.W, w.tiles
    alg_view = algorithmic_bytes(P, R_mean, H, W)
    dom = max(kernel_ms, key=kernel_ms.get)
    # the forward compositor of the view path carries the per-tile sort (K4 / K5) when the bucket capacity allows it
    fused_sort = dom == "render_fwd" and args.mode == "view" and "tile_sort" not in kernel_ms
    alg_key = "render_fwd_sorting" if fused_sort else dom
    alg_dom = KERNEL_ALG_BYTES.get(alg_key, lambda *a: 0)(P, R_mean, H * W, tiles)
    ach = alg_dom / (kernel_ms[dom] * 1e-3) / 1e9
    # HBM bytes per launch and instruction counts come from the rocprofv3 PMC passes committed under profiles/ -- they
    # belong to ONE workload: used only when that profile was taken on this --config (never as constants on another)
    traffic = None
    prof = committed_profile(args.config)
    if prof is not None:
        traffic = prof["traffic"].get(dom, {}).get("hbm_bytes_per_launch")
    out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                       "kernel_ms": round(kernel_ms[dom], 5), "algorithmic_bytes_per_launch": int(alg_dom)}
    if fused_sort:
        # a build that removes passes still reports against the reference algorithm's bytes for the work the
        # launch does (here K4 + K5 + K6: it sorts its tile's bucket, in LDS, before compositing).  The same time on the
        # K6 bytes alone -- what this launch can actually move through HBM -- is reported beside it, by name.
        k6 = KERNEL_ALG_BYTES["render_fwd"](P, R_mean, H * W, tiles)
        out["roofline"]["algorithmic_bytes_note"] = "K4 + K5 + K6 of the reference algorithm (SURVEY 8d); frac_k6_only: K6 alone"
        out["roofline"]["algorithmic_bytes_k6_only"] = int(k6)
        out["roofline"]["frac_k6_only"] = round(k6 / (kernel_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if traffic:
        out["roofline"]["frac_on_measured_traffic"] = round(traffic / (kernel_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if prof is not None:
        out["roofline"]["traffic_source"] = prof["source"]
        # the counters come from ANOTHER run (another box) than kernel_ms: the kernel's duration in that run, for scale
        pk = prof["traffic"].get(dom, {}).get("kernel_us_in_trace")
        if pk is not None:
            out["roofline"]["profile_kernel_ms"] = round(pk * 1e-3, 5)
            # the same fraction with time, traffic and counters from ONE collection (the committed profile run)
            out["roofline"]["frac_in_profile_run"] = round(alg_dom / (pk * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        insts = prof["pmc"].get(dom, {})
        if "SQ_INSTS_VALU" in insts:
            # what binds the dominant kernel: the SIMDs' vector issue.  Instruction counts per
            # launch from the committed PMC pass, rate from the live kernel time; peak = 1024 SIMDs x 2.4 GHz / 2
            # cycles per wave64 VALU instruction (v_fma_f32; most other instructions take 4 or more).
            # ONE peak for this quantity (DESIGN.md section 4 quotes the same): a SIMD issues a wave64 f32 FMA-class
            # instruction every 2 cycles, everything else (compares, selects, moves, integer) every 4 or more; the peak
            # is the FMA rate, and the measured mix's average cost is reported beside it so that "frac" can be read.
            rate = insts["SQ_INSTS_VALU"] / (kernel_ms[dom] * 1e-3) / 1e9
            peak = ISSUE_PEAK_G
            out["issue_roofline"] = {"bound": "valu-issue", "kernel": dom, "source": prof["source"],
                                     "valu_wave_instr_per_launch": int(insts["SQ_INSTS_VALU"]),
                                     "salu_instr_per_launch": int(insts.get("SQ_INSTS_SALU", 0)),
                                     "lds_instr_per_launch": int(insts.get("SQ_INSTS_LDS", 0)),
                                     "achieved": round(rate, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
                                     "frac": round(rate / peak, 4),
                                     "simd_cycles_per_valu_instr": round(1024 * 2.4 / rate, 3),
                                     "peak_note": "1024 SIMDs x 2.4 GHz / 2 cycles (wave64 v_fma_f32); 4-cycle "
                                                  "instructions (v_cmp, v_cndmask, integer) halve it: a mix at "
                                                  "simd_cycles_per_valu_instr ~3 is issue-saturated"}
        bi = prof["pmc"].get("render_bwd", {})
        if "SQ_INSTS_VALU" in bi and "render_bwd" in kernel_ms and dom != "render_bwd":
            rate_b = bi["SQ_INSTS_VALU"] / (kernel_ms["render_bwd"] * 1e-3) / 1e9
            out["issue_roofline_render_bwd"] = {
                "bound": "valu-issue", "kernel": "render_bwd", "source": prof["source"],
                "valu_wave_instr_per_launch": int(bi["SQ_INSTS_VALU"]), "salu_instr_per_launch": int(bi.get("SQ_INSTS_SALU", 0)),
                "lds_instr_per_launch": int(bi.get("SQ_INSTS_LDS", 0)), "achieved": round(rate_b, 1),
                "peak": round(ISSUE_PEAK_G, 1), "unit": "G wave-instr/s", "frac": round(rate_b / ISSUE_PEAK_G, 4),
                "simd_cycles_per_valu_instr": round(1024 * 2.4 / rate_b, 3)}
    out["kernel_ms_per_view"] = {k: round(v, 5) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])}
    if shared is not None:
        out["shared_sampling"] = shared
    out["whole_path"] = {"algorithmic_bytes_per_view": int(alg_view),
                         "achieved_GBps": round(alg_view / (ms_per_view * 1e-3) / 1e9, 2),   # whole job, per view
                         "hbm_roofline_frac": round(alg_view / (ms_per_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "sum_kernel_ms": round(sum(kernel_ms.values()), 5)}
    # A view whose pixels (nearly) all TERMINATE early -- cfg5: 1 M splats, every pixel saturates -- bins instances it never
    # composites (everything behind a pixel's cut): the R-proportional bytes then overstate what the compositors touch and
    # the fraction reads close to 1 for no merit of the kernels.  Such a line carries no whole-path fraction.
    term_frac = None
    if pipe.prof_direct[1] is not None:
        npix = H * W
        off_nc = (4 * npix + 127) // 128 * 128
        ncw = pipe.prof_direct[1]["img"][off_nc:off_nc + 4 * npix].view(torch.int32)
        term_frac = float((ncw < 0).float().mean())          # bit 31 of the saved n_contrib word: the pixel terminated
        out["whole_path"]["terminated_pixel_frac"] = round(term_frac, 4)
        if term_frac > 0.9:
            out["whole_path"]["hbm_roofline_frac_if_every_instance_were_composited"] = out["whole_path"]["hbm_roofline_frac"]
            out["whole_path"]["hbm_roofline_frac"] = None
            out["whole_path"]["hbm_roofline_frac_note"] = (
                f"n/a: {term_frac:.0%} of the pixels of the last profiled view terminate early (T < 1e-4); the instances "
                "behind their cuts are binned and sorted but never composited, so bytes proportional to R overstate the traffic")
    if stats.get("R_ref"):
        # the same fraction on the instance count the REFERENCE algorithm creates for these views (its 3-sigma tile rects:
        # tile culling drops the instances that stay below alpha 1/255 on all 256 pixels, the reference bins and sorts them)
        alg_ref = algorithmic_bytes(P, stats["R_ref"], H, W)
        out["whole_path"]["reference_instances_per_view"] = round(stats["R_ref"], 1)
        out["whole_path"]["algorithmic_bytes_per_view_reference_R"] = int(alg_ref)
        out["whole_path"]["hbm_roofline_frac_reference_R"] = (
            None if (term_frac is not None and term_frac > 0.9) else round(alg_ref / (ms_per_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))


def train_step_fields(out, w):
    """The other half of the metric (N = 1): the training iteration in its forms (graph replay, eager autograd, eager direct), the
    literal drop-in render() + backward per view, and the `general_route` block of the operator-API instances."""
    args, dev, lib, curves, my_cams = w.args, w.dev, w.lib, w.curves, w.my_cams
    H, W, m, P, tiles, bg, dL_dcolor = w.H, w.W, w.m, w.P, w.tiles, w.bg, w.dL_dcolor
    from curve_gaussian_amd.scene import GaussianCurveModel
    from curve_gaussian_amd.train_step import TrainStep
    gm = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                 curves["opacity"], curves["mask"],
                                                                 curves["is_bezier"])
    tcams = my_cams[:8]
    gg = torch.Generator(device="cpu").manual_seed(7)
    gts = [((torch.rand(1, H, W, generator=gg) > 0.97).float() * torch.rand(1, H, W, generator=gg)).to(dev) for _ in tcams]
    # the literal drop-in route: gaussian_renderer.render() with the reference's default arguments (train.py:95-97) and
    # autograd's backward of the image, one view at a time, eager launches.  For a GaussianCurveModel under the default
    # pipeline flags this is ONE autograd node over cgs_view_forward_checked / cgs_view_backward -- the headline kernels.
    from curve_gaussian_amd.gaussian_renderer import PipelineParams, render as dropin_render
    pipe = PipelineParams()
    import ctypes as _ct
    exact_path_views = [0]
    _path = _ct.c_int(0)

    def dropin_pass(cams, **kw):
        for c in cams:
            pkg = dropin_render(c, gm, pipe, bg, **kw)
            if kw.get("fused") is False:   # did this blocking forward fall back to the exact layout (a bucket overflow)?
                lib.cgs_last_forward_stats(None, None, _ct.byref(_path))
                exact_path_views[0] += 1 if _path.value == 0 else 0
            # (retain_graph: the general route differentiates through the prepare_scaling_rot graph, which train.py
            # rebuilds after every optimizer step and this loop keeps)
            torch.autograd.backward(pkg["render"], dL_dcolor.reshape(pkg["render"].shape), retain_graph=True)
    def dropin_time(**kw):
        """ms per view: median over 7 timings of 3 passes of all cameras (every camera rendered twice before: bucket capacities and
        binning hints of each view are settled, so no pass contains an overflow redo; the median drops host hiccups)."""
        for _ in range(2):
            dropin_pass(tcams, **kw)
        torch.cuda.synchronize()
        ts_ = []
        for _ in range(7):   # (one synchronisation per 24 views: the pipeline's fill / drain is not the steady state)
            td0 = time.perf_counter()
            for _ in range(3):
                dropin_pass(tcams, **kw)
            torch.cuda.synchronize()
            ts_.append((time.perf_counter() - td0) / (3 * len(tcams)) * 1e3)
        return round(sorted(ts_)[len(ts_) // 2], 4)
    for name, kw in (("dropin_view_ms", {}), ("dropin_view_no_visibility_ms", {"compute_visibility": False, "compute_rend_dir": False}),
                     ("dropin_view_general_route_ms", {"fused": False})):
        exact_path_views[0] = 0
        out[name] = dropin_time(**kw)
    out["dropin_general_route_exact_path_views"] = exact_path_views[0]   # of 8 x (2 + 21) forwards: bucket overflows redone
    # the same two routes through the ctypes bindings instead of the compiled host shim (CGS_TORCH_SHIM=0), same process, same box
    from curve_gaussian_amd import diff_cur_rasterization as _DCR
    prev_env = os.environ.get("CGS_TORCH_SHIM")
    os.environ["CGS_TORCH_SHIM"] = "0"
    _DCR._ExtProxy._impl = None
    try:
        for name, kw in (("dropin_view_ctypes_ms", {}), ("dropin_view_general_route_ctypes_ms", {"fused": False})):
            out[name] = dropin_time(**kw)
    finally:
        if prev_env is None:
            os.environ.pop("CGS_TORCH_SHIM", None)
        else:
            os.environ["CGS_TORCH_SHIM"] = prev_env
        _DCR._ExtProxy._impl = None
    out["dropin_note"] = ("every figure: median of 7 timings of 24 views, the lower of two such rounds (dropin_rounds); dropin_view_ms: render(cam, gaussians, pipe, bg) with the reference's defaults + backward of the "
                          "image per view, eager (fused view route: cgs_view_forward_checked / cgs_view_backward); "
                          "..._no_visibility: without the nonzero() host sync and the world-space direction map; "
                          "..._general_route: fused=False (GaussianRasterizer, the round-3 drop-in path); "
                          "..._ctypes: the same routes through the ctypes bindings (CGS_TORCH_SHIM=0) instead of the compiled "
                          "host shim curve_gaussian_amd/_cgs_torch.so, measured in the same process")
    if not args.no_general_route:
        # ---- what the reference's own GaussianRasterizer call reaches: kernel times per instance, the
        # backward compositor's HBM-roofline fraction on K8's algorithmic bytes, and the whole operator route
        gi = time_instances(args.config, 6, dev=dev)
        wl = gi.pop("_workload")
        Rg = float(wl["instances_R"])
        k8 = KERNEL_ALG_BYTES["render_bwd"](P, Rg, H * W, tiles)
        inst = {}
        for case, ks in gi.items():
            bwd_us = ks.get("render_bwd_unit_gated", 0.0) + ks.get("render_bwd", 0.0)
            inst[case] = {"render_bwd_us": round(bwd_us, 1), "render_fwd_us": ks.get("render_fwd"),
                          "sum_kernel_us": round(sum(ks.values()), 1),
                          "render_bwd_hbm_frac": round(k8 / (bwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        ref = gi["reference_call"]
        ref_sum_us = sum(ref.values())
        alg_ref_call = algorithmic_bytes(P, Rg, H, W)
        out["general_route"] = {
            "what": "GaussianRasterizer (operator API) on one view of this config: per-kernel times (us, HIP events, serial) "
                    "of the instance each kind of upstream gradient reaches; reference_call = the reference's own call "
                    "(unit colours without grad, only d/dcolour): device-side verdict -> pair-major unit backward",
            "instances_R": int(Rg),
            "instances": inst,
            "kernels_us": gi,
            "roofline": {"bound": "hbm", "kernel": "render_bwd (reference_call: k_render_bwd_unit, gated)",
                         "achieved": round(k8 / (inst["reference_call"]["render_bwd_us"] * 1e-6) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": inst["reference_call"]["render_bwd_hbm_frac"],
                         "traffic": None, "algorithmic_bytes_per_launch": int(k8),
                         "kernel_ms": round(inst["reference_call"]["render_bwd_us"] * 1e-3, 5)},
            "whole_route_reference_call": {"sum_kernel_ms": round(ref_sum_us * 1e-3, 5),
                                           "algorithmic_bytes_per_view": int(alg_ref_call),
                                           "hbm_roofline_frac": round(alg_ref_call / (ref_sum_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                           "eager_ms_per_view": out.get("dropin_view_general_route_ms")},
        }
    for prm in (gm._curve_points, gm._width, gm._opacity, gm._mask):
        prm.grad = None
    # Every form of the iteration is timed over the SAME window of a run from the synthetic initial state -- iterations
    # 4 .. 35 after three warm-up iterations: the optimizer moves the scene (opacities fall against the sparse random
    # targets), so the work per iteration drifts (cfg3: -25 % over the first 130 iterations) and figures from different
    # windows are not comparable.
    n_ts = 32

    def time_eager(ts):
        for _ in range(3):
            ts.step()
        torch.cuda.synchronize()
        tt0 = time.perf_counter()
        for _ in range(n_ts):
            ts.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - tt0) / n_ts * 1e3

    def eager_median(**kw):
        """Median of three runs, each from a fresh model (the window is part of the figure); the eager forms are host-bound
        on the small configs and a noisy host second would otherwise decide the number."""
        runs = []
        for _ in range(3):
            gmx = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                          curves["opacity"], curves["mask"],
                                                                          curves["is_bezier"])
            runs.append(time_eager(TrainStep(gmx, tcams, gts, **kw)))
            del gmx
        return sorted(runs)[1]

    eager_ms = eager_median()
    # the same eager iteration without autograd (TrainStep(direct=True): library calls one after the other, exact binning,
    # nothing captured)
    out["train_step_eager_direct_ms"] = round(eager_median(direct=True), 4)
    # second round of the eager drop-in timings, minutes of wall clock after the first: each figure is the lower of its
    # two medians (both kept in dropin_rounds) -- these routes are host-bound and a busy host inflates a whole round
    second = {}
    for name, kw in (("dropin_view_ms", {}), ("dropin_view_no_visibility_ms", {"compute_visibility": False, "compute_rend_dir": False}),
                     ("dropin_view_general_route_ms", {"fused": False})):
        second[name] = dropin_time(**kw)
    out["dropin_rounds"] = {"first": {k: out[k] for k in second}, "second": second}
    for k, v in second.items():
        out[k] = min(out[k], v)
    if "general_route" in out:
        out["general_route"]["whole_route_reference_call"]["eager_ms_per_view"] = out["dropin_view_general_route_ms"]
    # the same iteration replayed as one hipGraph launch (sync-free forward, device-state Adam)
    from curve_gaussian_amd.train_step import GraphedTrainStep
    gm2 = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                  curves["opacity"], curves["mask"],
                                                                  curves["is_bezier"])
    gs = GraphedTrainStep(gm2, tcams, gts)
    for _ in range(3):
        gs.step()
    gs.finish()
    torch.cuda.synchronize()
    n_gs = 32
    tt0 = time.perf_counter()
    for _ in range(n_gs):
        gs.step()
    gs.finish()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - tt0) / n_gs * 1e3
    # ... and with the regularisers of train.py:113-131 switched on (not part of the BASELINE metric)
    gm3 = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                  curves["opacity"], curves["mask"],
                                                                  curves["is_bezier"])
    gr = GraphedTrainStep(gm3, tcams, gts, regularisers=True)
    for _ in range(3):
        gr.step()
    gr.finish()
    torch.cuda.synchronize()
    tt0 = time.perf_counter()
    for _ in range(n_gs):
        gr.step()
    gr.finish()
    torch.cuda.synchronize()
    out["train_step_with_regularisers_ms"] = round((time.perf_counter() - tt0) / n_gs * 1e3, 4)
    # the late phase of train.py (iteration > 7000): straight-through curve mask + mask loss + every regulariser
    # including the end-point connection loss (O(B^2) memory in the reference, a hashed grid here)
    gm4 = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                  curves["opacity"], curves["mask"],
                                                                  curves["is_bezier"])
    gl = GraphedTrainStep(gm4, tcams, gts, regularisers=True, densify_until_iter=0, conn_from_iter=0)
    for _ in range(3):
        gl.step()
    gl.finish()
    torch.cuda.synchronize()
    tt0 = time.perf_counter()
    for _ in range(n_gs):
        gl.step()
    gl.finish()
    torch.cuda.synchronize()
    out["train_step_late_phase_ms"] = round((time.perf_counter() - tt0) / n_gs * 1e3, 4)
    # the iteration with an image-only forward: train.py:98-107 reads `render` alone, inverse depth and all_map are
    # computed by the reference's kernel but consumed by the TensorBoard report only (train.py:351-364)
    gm5 = GaussianCurveModel(0, m, device=dev).create_from_curves(curves["curve_points"], curves["width"],
                                                                  curves["opacity"], curves["mask"],
                                                                  curves["is_bezier"])
    gi = GraphedTrainStep(gm5, tcams, gts, aux_outputs=False)
    for _ in range(3):
        gi.step()
    gi.finish()
    torch.cuda.synchronize()
    tt0 = time.perf_counter()
    for _ in range(n_gs):
        gi.step()
    gi.finish()
    torch.cuda.synchronize()
    out["train_step_image_only_forward_ms"] = round((time.perf_counter() - tt0) / n_gs * 1e3, 4)
    out["train_step_ms"] = round(graph_ms, 4)
    out["train_step_eager_ms"] = round(eager_ms, 4)
    out["train_step_graph_recaptures"] = gs.recaptures
    out["train_step_note"] = ("every figure = mean over iterations 4..35 of a run from the synthetic initial state (the work per iteration drifts as the optimizer moves the scene); train_step_ms: GraphedTrainStep (whole iteration = one hipGraph replay); "
                              "train_step_eager_ms: TrainStep (Python autograd, ~30 launches); train_step_eager_direct_ms: TrainStep(direct=True), the same eager iteration as plain library calls without autograd (exact binning, nothing captured).  Iteration = lr update + view pick + render (fused attrs + raster) + edge_aware_loss + fused_ssim "
                              "+ backward + Adam (6 groups) + prepare_scaling_rot; regularisers of train.py:110-146 "
                              "excluded (SURVEY 8d).  train_step_image_only_forward_ms: the same iteration when the forward "
                              "writes `render` only (GraphedTrainStep(aux_outputs=False)); NOT the headline: the reference's "
                              "kernel always produces inverse depth and all_map")


def cpu_baseline_fields(out, w):
    """`cpu_baseline`: the oracle port (oracle/raster_ref.c with OpenMP, oracle/torch_ref.py) timed on a bounded sample of the same
    workload on this box's host cores, and GPU vs CPU parity on the first of its views."""
    from curve_gaussian_amd.diff_cur_rasterization import _C
    args, curves, my_cams, amaps, empty = w.args, w.curves, w.my_cams, w.amaps, w.empty
    xyz, colors, opac, scl, rotn, bg, dL_dcolor = w.xyz, w.colors, w.opac, w.scl, w.rotn, w.bg, w.dL_dcolor
    H, W, m, P, tanx, tany = w.H, w.W, w.m, w.P, w.tanx, w.tany
    import numpy as np
    import oracle
    oracle.build()
    from oracle import raster as ORA
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    ORA.set_num_threads(threads)
    n = lambda t: t.detach().cpu().numpy()
    a_xyz, a_col, a_op, a_sc, a_rot = n(xyz), n(colors), n(opac), n(scl), n(rotn)
    dcol = n(dL_dcolor)
    nv = max(1, args.cpu_views)
    t_fwd = t_bwd = 0.0
    err = {}
    for i in range(nv):
        cam = my_cams[i % len(my_cams)]
        tc0 = time.perf_counter()
        fw = ORA.forward(n(bg), a_xyz, a_col, a_op, a_sc, a_rot, 1.0, None, n(amaps[id(cam)]),
                         n(cam.world_view_transform), n(cam.full_proj_transform), tanx, tany, H, W, None, 0,
                         n(cam.camera_center))
        tc1 = time.perf_counter()
        gr = ORA.backward(fw, dcol, None, None)
        t_fwd += tc1 - tc0
        t_bwd += time.perf_counter() - tc1
        if i == 0:   # GPU vs CPU on the same inputs: worst error normalised by the tensor's max
            (R_g, color_g, radii_g, gB, bB, iB, invd_g, om_g) = _C.rasterize_gaussians(
                bg, xyz, colors, opac, scl, rotn, 1.0, empty, amaps[id(cam)], cam.world_view_transform,
                cam.full_proj_transform, tanx, tany, H, W, empty, 0, cam.camera_center, False, False, True, False)
            g_g = _C.rasterize_gaussians_backward(
                bg, empty, xyz, radii_g, colors, amaps[id(cam)], opac, scl, rotn, 1.0, empty,
                cam.world_view_transform, cam.full_proj_transform, tanx, tany, dL_dcolor, empty, empty, empty, 0,
                cam.camera_center, gB, R_g, bB, iB, False, True, False)
            def nrm(a, b):   # parity criterion of tests/util.py: fraction of elements off by > 1e-4 * max|ref|
                e = np.abs(np.asarray(a, np.float64) - b) / (np.abs(b).max() + 1e-30)
                return float((e > 1e-4).mean())
            err = {"color": nrm(n(color_g), fw.color), "all_map": nrm(n(om_g), fw.out_all_map),
                   "dL_dmeans3D": nrm(n(g_g[3]), gr["dL_dmeans3D"]), "dL_dscales": nrm(n(g_g[6]), gr["dL_dscales"]),
                   "dL_dopacity": nrm(n(g_g[2]), gr["dL_dopacity"]),
                   "radii_equal": bool((n(radii_g) == fw.radii).all())}
        fw.free()
    # the rest of the per-view path on the host: curve sampling (prepare_scaling_rot) and the splat attributes with their
    # backward, torch restatement (oracle/torch_ref.py) on the same thread count -- `value` times these too
    t_samp = 0.0
    if args.mode == "view":
        from oracle import torch_ref as TR
        torch.set_num_threads(threads)
        cpu = lambda t: t.detach().cpu()
        cam0 = my_cams[0]
        ts0 = time.perf_counter()
        lv = [cpu(curves[k]).clone().requires_grad_(True) for k in ("curve_points", "width", "opacity")]
        x_c, r_c, s_c = TR.prepare_scaling_rot(lv[0], lv[1], cpu(curves["is_bezier"]))
        rn_c = torch.nn.functional.normalize(r_c)
        op_c = torch.sigmoid(lv[2]).repeat_interleave(m, 0)
        am_c = TR.build_all_map(r_c.detach(), x_c.detach(), cpu(cam0.camera_center), cpu(cam0.world_view_transform))
        (x_c.sum() + s_c.sum() + rn_c.sum() + op_c.sum() + am_c.sum() * 0).backward()
        t_samp = time.perf_counter() - ts0
    tc = t_fwd + t_bwd + nv * t_samp
    out["cpu_baseline"] = {"value": round(P * nv / tc / 1e6, 4), "unit": "Msplats/s", "cores": threads,
                           "kind": "port", "fwd_ms_per_view": round(t_fwd / nv * 1e3, 1),
                           "bwd_ms_per_view": round(t_bwd / nv * 1e3, 1),
                           "sampling_attrs_fwd_bwd_ms_per_view": round(t_samp * 1e3, 1),
                           "gpu_vs_cpu_frac_over_1e-4_of_max": {k: (v if isinstance(v, bool) else float(f"{v:.2e}")) for k, v in err.items()},
                           "covers": ("the whole per-view path, like value: curve sampling + splat attributes + "
                                      "raster fwd + bwd + curve-parameter gradients" if args.mode == "view" else
                                      "rasterizer forward + backward"),
                           "sample": f"{nv} views of the same workload: raster fwd+bwd through oracle/raster_ref.c with "
                                     f"OpenMP ({threads} threads of {cores} host cores); sampling + attributes and their "
                                     f"backward through oracle/torch_ref.py (torch, {threads} threads), timed once and "
                                     f"charged per view; {tc:.1f} s"}
    if args.config == "cfg1" and args.torch_cpu_splats > 0:
        # SURVEY 8d / north_star: the "PyTorch-CPU raster fallback" beside the C port, on cfg1.  The reference itself has no
        # such fallback; this is the oracle's dense pure-PyTorch differentiable restatement, whose cost is
        # P x H x W whatever the splats' footprints, so a sample of S splats over the full image extrapolates linearly.
        from oracle import torch_ref as TR
        S_ = min(P, args.torch_cpu_splats)
        torch.set_num_threads(threads)
        cam = my_cams[0]
        c = lambda t: t.detach().cpu()
        leaves = [c(t)[:S_].clone().requires_grad_(True) for t in (xyz, opac, scl, rotn, colors)]
        tp0 = time.perf_counter()
        col_t, _, _, _ = TR.dense_render(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], c(amaps[id(cam)])[:S_],
                                         c(cam.world_view_transform), c(cam.full_proj_transform), tanx, tany, H, W, c(bg))
        tp1 = time.perf_counter()
        col_t.backward(c(dL_dcolor).reshape(col_t.shape))
        tp2 = time.perf_counter()
        out["cpu_baseline"]["torch_cpu_fallback"] = {
            "value": round(S_ / (tp2 - tp0) / 1e6, 8), "unit": "Msplats/s", "cores": threads,
            "fwd_ms_per_view_extrapolated": round((tp1 - tp0) / S_ * P * 1e3, 1),
            "bwd_ms_per_view_extrapolated": round((tp2 - tp1) / S_ * P * 1e3, 1),
            "sample": f"{S_} of the view's {P} splats over the full {W}x{H} image, oracle/torch_ref.dense_render + autograd "
                      f"(dense P x H x W: cost per splat does not depend on its footprint), torch.set_num_threads({threads}), "
                      f"{tp2 - tp0:.1f} s"}


def child_line(cfg, extra=()):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", cfg, "--steps", "8", "--warmup", "2",
           "--min-seconds", "1.0", "--no-cpu-baseline", "--no-general-route", "--no-realistic-size"] + list(extra)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return None, (r.stderr or r.stdout)[-300:]
    return json.loads(line[-1]), None


def realistic_size_fields(out):
    """The size the reference actually trains (BASELINE cfg2: ~50 k splats at 1600^2): the headline rate, the
    training iteration in its three forms and the literal drop-in render() + backward, from one child run on this GPU after
    everything else is done with it."""
    j, err = child_line("cfg2", ("--no-kernel-times",))
    if j is None:
        out["realistic_size"] = {"config": "cfg2", "error": err}
        return
    out["realistic_size"] = {
        "config": "cfg2", "splats": j["config"]["splats"], "value": j["value"], "ms_per_view": j["ms_per_view"],
        "train_step_ms": j.get("train_step_ms"), "train_step_eager_ms": j.get("train_step_eager_ms"),
        "train_step_eager_direct_ms": j.get("train_step_eager_direct_ms"), "dropin_view_ms": j.get("dropin_view_ms"),
        "note": "train_step_ms: graph replay; _eager_ms: render() + loss.backward() through Python autograd (the literal "
                "drop-in loop); _eager_direct_ms: the same iteration as plain library calls (TrainStep(direct=True)); "
                "dropin_view_ms: render() + backward alone"}


def all_configs_fields(out):
    """One child per BASELINE config on this same GPU, after everything else is done with it: the headline fields of each line,
    compact (the children skip the CPU baseline and the operator-instance block; their timed region is shorter)."""
    allc = {}
    for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        j, err = child_line(cfg)
        if j is None:
            allc[cfg] = {"error": err}
            continue
        allc[cfg] = {"value": j["value"], "ms_per_view": j["ms_per_view"],
                     "whole_path_hbm_frac": j.get("whole_path", {}).get("hbm_roofline_frac"),
                     "terminated_pixel_frac": j.get("whole_path", {}).get("terminated_pixel_frac"),
                     "serial_view_graph_ms": j.get("serial_view_graph_ms"), "train_step_ms": j.get("train_step_ms"),
                     "train_step_eager_ms": j.get("train_step_eager_ms"),
                     "train_step_eager_direct_ms": j.get("train_step_eager_direct_ms"), "dropin_view_ms": j.get("dropin_view_ms"),
                     "dropin_view_general_route_ms": j.get("dropin_view_general_route_ms"),
                     "splats": j["config"]["splats"], "instances_per_view_R": j["config"]["instances_per_view_R"]}
    out["all_configs"] = allc


def main():
    args = parse_args()
    who = start_rank(args)
    w = Workload(args, who)              # resident in HBM before timing
    pipe = ViewPipeline(w)               # streams, gradient sets, captured graphs
    head = time_headline(pipe)           # THE timed region; everything below runs after it
    grad_check = check_step_gradient(pipe)
    serial_ms, serial_graph_ms = time_serial_views(pipe)
    vp_train_ms = time_view_parallel_train_step(pipe)
    kernel_ms, R_mean, vis_mean = time_kernels(pipe)
    shared = None
    if kernel_ms and args.mode == "view" and pipe.cap and w.world == 1:
        shared = time_shared_sampling(pipe)
    dist = pipe.dist
    if w.rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    out, ms_per_view = headline_fields(pipe, head, R_mean, vis_mean)
    scaling_fields(out, pipe, head)
    if vp_train_ms is not None:   # one optimizer step = `world` views (one per rank), gradients summed by ONE all-reduce
        out["train_step_view_parallel_ms"] = round(vp_train_ms, 4)
    if grad_check is not None:
        out["step_gradient_rel_l2_vs_serial_eager"] = float(f"{grad_check:.3e}")
    if serial_ms is not None:
        out["serial_view_ms"] = round(serial_ms, 4)   # one view in flight, eager launches, all-reduce after every view
        if serial_graph_ms is not None:
            out["serial_view_graph_ms"] = round(serial_graph_ms, 4)   # same, one hipGraph replay per view
    if kernel_ms:
        roofline_fields(out, pipe, kernel_ms, shared, R_mean, ms_per_view)
    if w.world == 1 and not args.no_train_step:
        train_step_fields(out, w)
    if w.world == 1 and not args.no_cpu_baseline:
        cpu_baseline_fields(out, w)
    if (w.world == 1 and args.config == "cfg3" and args.mode == "view" and not args.no_realistic_size and not args.all_configs
            and not args.no_train_step):
        realistic_size_fields(out)
    if args.all_configs and w.world == 1:
        all_configs_fields(out)
    if dist is not None:
        dist.destroy_process_group()
    try:   # RCCL writes its banner through C stdio: flush that buffer so the JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
