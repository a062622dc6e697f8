"""Hot-path part of the reference's ``scene.gaussian_curve_model.GaussianCurveModel``
(/root/reference/scene/gaussian_curve_model.py:54-198): the learnable curve tensors, their layout, and the
per-step derivation of per-splat tensors (``prepare_scaling_rot``).  Topology edits (densify / split / prune / trim, :246-463) live in scene/topology.py and are
installed as methods below; merge_curves (:466ff) is not reproduced.

Tensor layout (kept exactly): ``_curve_points [B,4,3]``, ``_width [B,1]`` (log), ``_opacity [B,1]`` (logit),
``_mask [B,m,1]``, ``_features_dc [B,m,1,1]``, ``_features_rest [B,m,(D+1)^2-1,1]``, ``is_bezier [B]``; derived
(non-leaf, carry autograd): ``_xyz [P,3]``, ``_rotation [P,4]`` (w,x,y,z un-normalised), ``_scaling [P,3]`` with
splat index = b*m + i.
"""
import numpy as np
import torch
from torch import nn

from ..ops import curve_sampling


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with optional delayed warm-up (reference utils/general_utils.py:99-132)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        log_lerp = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
        return delay_rate * log_lerp
    return helper


def initialize_bezier_curves(points, bound, n_control_points=4):
    """One cubic Bezier per seed point, laid along +-Y (reference :27-51): P0 = p - (0, bound, 0), P3 = p + (0, bound, 0),
    P1 / P2 at half that distance.  points [B,3], bound [B,1] -> [B,4,3]."""
    assert n_control_points == 4
    direction = torch.cat([torch.zeros_like(bound), bound, torch.zeros_like(bound)], dim=1)
    return torch.stack([points - direction, points - 0.5 * direction, points + 0.5 * direction, points + direction], dim=1)


def _optimizer_state(opt):
    """Adam state by group name: {"step": int, "groups": {name: {"lr", "exp_avg", "exp_avg_sq"}}} from a torch.optim.Adam
    with named single-parameter groups or from ops.optim.FlatAdam (moments None before the first step)."""
    if opt is None:
        return None
    out = {"step": 0, "groups": {}}
    flat = hasattr(opt, "state_of")
    if flat:
        out["step"] = int(opt.step_count)
    for g in opt.param_groups:
        name, p = g["name"], g["params"][0]
        if flat:
            st = opt.state_of(name)
            m, v = (st[0].detach().clone(), st[1].detach().clone()) if st is not None else (None, None)
        else:
            st = opt.state.get(p, None)
            m, v = (st["exp_avg"].detach().clone(), st["exp_avg_sq"].detach().clone()) if st else (None, None)
            if st:
                out["step"] = int(st["step"])
        out["groups"][name] = {"lr": float(g["lr"]), "exp_avg": m, "exp_avg_sq": v}
    return out


def _load_optimizer_state(opt, st):
    if opt is None or st is None:
        return
    flat = hasattr(opt, "state_of")
    if flat:
        opt.step_count = int(st["step"])
    for g in opt.param_groups:
        e = st["groups"].get(g["name"])
        if e is None:
            continue
        g["lr"] = e["lr"]
        if e["exp_avg"] is None:
            continue
        p = g["params"][0]
        if flat:
            m, v = opt.state_of(g["name"])
            m.copy_(e["exp_avg"].to(m.device))
            v.copy_(e["exp_avg_sq"].to(v.device))
        else:
            opt.state[p] = {"step": torch.tensor(float(st["step"])), "exp_avg": e["exp_avg"].to(p.device).clone(),
                            "exp_avg_sq": e["exp_avg_sq"].to(p.device).clone()}


class GaussianCurveModel:
    def __init__(self, sh_degree: int = 0, n_gaussians: int = 12, optimizer_type: str = "default", device="cuda"):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.optimizer_type = optimizer_type
        self.n_gaussians = n_gaussians
        self.device = torch.device(device)
        self._curve_points = torch.empty(0)
        self._width = torch.empty(0)
        self._opacity = torch.empty(0)
        self._mask = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self.is_bezier = torch.empty(0)
        # Derived per-splat tensors.  lazy_derived (set by loops that own the parameter updates, train_step.TrainStep): a
        # prepare_scaling_rot() only records what to derive from, the sampling kernels run when `_xyz` / `_rotation` /
        # `_scaling` are first READ -- the fused render route samples inside its own kernels and never reads them, so a
        # training iteration runs the grid-wide norm pass once, not twice.  Off: derived on the spot, like the reference.
        self.lazy_derived = False
        self._derived_pending = None          # (eps, grad mode) of a prepare_scaling_rot() that has not run its kernels yet
        self._derived = [torch.empty(0), torch.empty(0), torch.empty(0)]
        self.optimizer = None
        self.exposure_optimizer = None
        self.exposure_mapping = {}
        self.pretrained_exposures = None
        self._exposure = nn.Parameter(torch.eye(3, 4)[None].repeat(0, 1, 1).requires_grad_(True))   # [n_cameras, 3, 4]
        self.spatial_lr_scale = 0
        # activations, scene/gaussian_model.py:38-53
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ------------------------------------------------------------------ construction
    def create_from_curves(self, curve_points, width, opacity, mask=None, is_bezier=None):
        """Install curve parameters (the reference builds them in create_from_pcd :142-178 from a point cloud)."""
        dev = self.device
        B = curve_points.shape[0]
        m = self.n_gaussians
        self._curve_points = nn.Parameter(curve_points.to(dev).float().contiguous().requires_grad_(True))
        self._width = nn.Parameter(width.to(dev).float().contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(opacity.to(dev).float().contiguous().requires_grad_(True))
        if mask is None:
            mask = torch.ones(B, m, 1)
        self._mask = nn.Parameter(mask.to(dev).float().contiguous().requires_grad_(True))
        self._features_dc = nn.Parameter(torch.zeros(B, m, 1, 1, device=dev).requires_grad_(True))
        self._features_rest = nn.Parameter(
            torch.zeros(B, m, (self.max_sh_degree + 1) ** 2 - 1, 1, device=dev).requires_grad_(True))
        if is_bezier is None:
            is_bezier = torch.ones(B, dtype=torch.bool)
        self.is_bezier = is_bezier.to(dev)
        self.max_radii2D = torch.zeros(B * m, device=dev)
        self.prepare_scaling_rot()
        return self

    def create_from_pcd(self, pcd, cam_infos, spatial_lr_scale: float, init_size: float = 0.5, n_control_points: int = 4):
        """Reference :142-178: one curve per point of the seed cloud.  bound = init_size * sqrt(mean squared distance to
        the 3 nearest neighbours) from the HIP ``simple_knn.distCUDA2``; control points along +-Y; opacity 0.6, width
        5e-3; DC feature = RGB2SH of the red channel, replicated over the curve's m samples; mask = 1; all curves Bezier;
        one 3x4 identity exposure per camera.  `pcd`: anything with ``.points`` / ``.colors`` ([N,3] arrays)."""
        from ..simple_knn import distCUDA2
        from .dataset_io import RGB2SH
        dev = self.device
        m = self.n_gaussians
        self.spatial_lr_scale = spatial_lr_scale
        fused_point_cloud = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        dist2 = torch.clamp_min(distCUDA2(torch.from_numpy(np.asarray(pcd.points)).float().to(dev)), 0.0000001)
        self.dist = torch.sqrt(dist2).mean()
        bound = init_size * torch.sqrt(dist2).unsqueeze(1)
        points_per_curve = initialize_bezier_curves(fused_point_cloud, bound, n_control_points)
        B = fused_point_cloud.shape[0]
        opacities = torch.logit(0.6 * torch.ones((B, 1), dtype=torch.float, device=dev))   # inverse_sigmoid
        widths = self.scaling_inverse_activation(5e-3 * torch.ones((B, 1), dtype=torch.float, device=dev))
        pcd_colors = np.asarray(pcd.colors)[:, None, :].repeat(m, axis=1)
        fused_color = RGB2SH(torch.tensor(np.asarray(pcd_colors[..., 0:1])).float().to(dev))
        features = torch.zeros((B, m, 1, (self.max_sh_degree + 1) ** 2)).float().to(dev)
        features[:, :, :1, 0] = fused_color
        features[:, :, 1:, 1:] = 0.0
        self._curve_points = nn.Parameter(points_per_curve.contiguous().requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, :, 0:1].transpose(2, 3).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, :, 1:].transpose(2, 3).contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self._width = nn.Parameter(widths.requires_grad_(True))
        self._mask = nn.Parameter(torch.ones((B, m, 1), device=dev).requires_grad_(True))
        self.max_radii2D = torch.zeros(B * m, device=dev)
        self.is_bezier = torch.ones(B, dtype=torch.bool, device=dev)
        cam_infos = list(cam_infos) if cam_infos is not None else []
        self.exposure_mapping = {cam_info.image_name: idx for idx, cam_info in enumerate(cam_infos)}
        self.pretrained_exposures = None
        exposure = torch.eye(3, 4, device=dev)[None].repeat(len(cam_infos), 1, 1)
        self._exposure = nn.Parameter(exposure.requires_grad_(True))
        self.prepare_scaling_rot()
        return self

    # hyper-parameters of training_setup with the reference's defaults (arguments/__init__.py:79-114)
    _TRAINING_DEFAULTS = dict(feature_lr=0.0025, opacity_lr=0.025, scaling_lr=0.005, lr_curve_points_init=0.0005,
                              mask_lr=0.01, lr_curve_points_final=0.000005, position_lr_delay_mult=0.01,
                              position_lr_max_steps=30000, exposure_lr_init=0.01, exposure_lr_final=0.001,
                              exposure_lr_delay_steps=0, exposure_lr_delay_mult=0.0, iterations=10000)

    def training_setup(self, training_args=None, **kw):
        """Adam groups of the reference (:200-232; lrs from arguments/__init__.py:83-114), the exposure optimizer and the two
        schedules; the densification statistics start at zero (:201-202).  `training_args`: the reference's
        OptimizationParams-style object (train.py:48 ``gaussians.training_setup(opt)``), attributes missing from it and
        keyword arguments fall back to / override the reference's defaults."""
        hp = dict(self._TRAINING_DEFAULTS)
        if training_args is not None:
            hp.update({k: getattr(training_args, k) for k in hp if hasattr(training_args, k)})
        unknown = set(kw) - set(hp)
        if unknown:
            raise TypeError(f"training_setup: unknown hyper-parameters {sorted(unknown)}")
        hp.update(kw)
        P = self._curve_points.shape[0] * self.n_gaussians
        self.denom = torch.zeros((P, 1), device=self._curve_points.device)
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self._curve_points.device)
        l = [
            {'params': [self._features_dc], 'lr': hp["feature_lr"], "name": "f_dc"},
            {'params': [self._features_rest], 'lr': hp["feature_lr"] / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': hp["opacity_lr"], "name": "opacity"},
            {'params': [self._width], 'lr': hp["scaling_lr"], "name": "width"},
            {'params': [self._curve_points], 'lr': hp["lr_curve_points_init"], "name": "curve_points"},
            {'params': [self._mask], 'lr': hp["mask_lr"], "name": "mask"},
        ]
        self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
        self.exposure_optimizer = torch.optim.Adam([self._exposure])                                    # :221
        self.curve_scheduler_args = get_expon_lr_func(lr_init=hp["lr_curve_points_init"], lr_final=hp["lr_curve_points_final"],
                                                      lr_delay_mult=hp["position_lr_delay_mult"],
                                                      max_steps=hp["position_lr_max_steps"])
        self.exposure_scheduler_args = get_expon_lr_func(hp["exposure_lr_init"], hp["exposure_lr_final"],    # :228-232
                                                         lr_delay_steps=hp["exposure_lr_delay_steps"],
                                                         lr_delay_mult=hp["exposure_lr_delay_mult"],
                                                         max_steps=hp["iterations"])
        return self.optimizer

    def update_learning_rate(self, iteration):
        """:234-244 (exposure schedule: scene/gaussian_model.py:257-259)"""
        if self.pretrained_exposures is None and self.exposure_optimizer is not None:
            for param_group in self.exposure_optimizer.param_groups:
                param_group['lr'] = self.exposure_scheduler_args(iteration)
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "curve_points":
                lr = self.curve_scheduler_args(iteration)
                param_group['lr'] = lr
                return lr

    def oneupSHdegree(self):
        """scene/gaussian_model.py:187-189 (train.py:81-82, every 1000 iterations; a no-op at the default sh_degree 0)"""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    @property
    def get_exposure(self):
        return self._exposure

    def get_exposure_from_name(self, image_name):
        """scene/gaussian_model.py:178-182"""
        if self.pretrained_exposures is None:
            return self._exposure[self.exposure_mapping[image_name]]
        return self.pretrained_exposures[image_name]

    # ------------------------------------------------------------------ checkpoint / resume
    def capture(self):
        """train.py:238-240 ``torch.save((gaussians.capture(), iteration), ...)``.  DEVIATION from the reference, whose
        GaussianCurveModel inherits the base class's capture / restore (scene/gaussian_model.py:74-106): those save the DERIVED
        splat tensors and drop `_curve_points / _width / _mask / is_bezier`, so a resumed run cannot continue (SURVEY quirk
        19).  Here the curve tensors, the statistics, both optimizers' state and the exposures round-trip; the optimizer state
        is stored per group NAME (torch.optim.Adam or the flat one-launch Adam: either can resume the other's checkpoint)."""
        t = lambda x: x.detach().clone()
        return {
            "format": "curvegs-checkpoint-1",
            "active_sh_degree": self.active_sh_degree, "max_sh_degree": self.max_sh_degree, "n_gaussians": self.n_gaussians,
            "curve_points": t(self._curve_points), "width": t(self._width), "opacity": t(self._opacity), "mask": t(self._mask),
            "features_dc": t(self._features_dc), "features_rest": t(self._features_rest), "is_bezier": t(self.is_bezier),
            "max_radii2D": t(self.max_radii2D),
            "xyz_gradient_accum": t(self.xyz_gradient_accum) if hasattr(self, "xyz_gradient_accum") else None,
            "denom": t(self.denom) if hasattr(self, "denom") else None,
            "optimizer": _optimizer_state(self.optimizer),
            "exposure": t(self._exposure), "exposure_mapping": dict(self.exposure_mapping),
            "exposure_optimizer": self.exposure_optimizer.state_dict() if self.exposure_optimizer is not None else None,
            "spatial_lr_scale": self.spatial_lr_scale,
        }

    def restore(self, model_args, training_args=None):
        """train.py:49-51 ``gaussians.restore(model_params, opt)``: the inverse of ``capture`` (see there)."""
        a = model_args
        if isinstance(a, (tuple, list)) and len(a) == 12:
            # the reference's GaussianModel.capture() tuple (scene/gaussian_model.py:74-88: active_sh_degree, _xyz, _features_dc,
            # _features_rest, _scaling, _rotation, _opacity, max_radii2D, xyz_gradient_accum, denom, optimizer state_dict,
            # spatial_lr_scale), i.e. a chkpnt*.pth written by the reference's train.py
            raise ValueError(
                "restore: this is a checkpoint in the reference's GaussianModel.capture() layout (12-tuple of DERIVED splat "
                "tensors).  It does not contain the curve parameters (_curve_points, _width, _mask, is_bezier) -- the reference "
                "itself cannot resume a curve model from it (scene/gaussian_model.py:74-106 vs gaussian_curve_model.py:54-64) "
                "-- so there is nothing to rebuild the model from.  Checkpoints written by this package's capture() "
                "('curvegs-checkpoint-1') round-trip; see INTEGRATION.md, 'Checkpoints'.")
        if not (isinstance(a, dict) and a.get("format") == "curvegs-checkpoint-1"):
            raise ValueError("restore: not a checkpoint written by GaussianCurveModel.capture() (expected a dict with "
                             "format == 'curvegs-checkpoint-1')")
        if a["n_gaussians"] != self.n_gaussians:
            raise ValueError(f"restore: checkpoint has {a['n_gaussians']} samples per curve, the model {self.n_gaussians}")
        dev = self.device
        par = lambda x: nn.Parameter(x.to(dev).float().contiguous().requires_grad_(True))
        self.active_sh_degree, self.max_sh_degree = a["active_sh_degree"], a["max_sh_degree"]
        self._curve_points, self._width, self._opacity, self._mask = (par(a[k]) for k in ("curve_points", "width", "opacity", "mask"))
        self._features_dc, self._features_rest = par(a["features_dc"]), par(a["features_rest"])
        self.is_bezier = a["is_bezier"].to(dev)
        self.max_radii2D = a["max_radii2D"].to(dev)
        self._exposure = par(a["exposure"])
        self.exposure_mapping = dict(a["exposure_mapping"])
        self.spatial_lr_scale = a["spatial_lr_scale"]
        self.prepare_scaling_rot()
        self.training_setup(training_args)
        if a["xyz_gradient_accum"] is not None:
            self.xyz_gradient_accum, self.denom = a["xyz_gradient_accum"].to(dev), a["denom"].to(dev)
        _load_optimizer_state(self.optimizer, a["optimizer"])
        if a["exposure_optimizer"] is not None:
            self.exposure_optimizer.load_state_dict(a["exposure_optimizer"])
        return self

    # ------------------------------------------------------------------ per-step derivation
    def prepare_scaling_rot(self, eps=1e-8):
        """:180-198 -- fused HIP sampling kernel (forward + hand-written backward)."""
        if self.lazy_derived and self._curve_points.is_cuda:
            self._derived_pending = (float(eps), torch.is_grad_enabled())
        else:
            self._derived_pending = None
            self._derived = list(curve_sampling.sample_curves(self._curve_points, self._width, self.is_bezier,
                                                              self.n_gaussians, eps))
        # which parameter state the derived tensors belong to: render() takes its fused per-view route (which samples the
        # curves itself) only while they are current, so both routes draw the same splats
        self._derived_eps = float(eps)
        self._derived_from = self._param_stamp()

    def _param_stamp(self):
        return (self._curve_points.data_ptr(), self._curve_points._version, self._width.data_ptr(), self._width._version,
                tuple(self._curve_points.shape), self.is_bezier.data_ptr(), self.is_bezier._version)

    def _derive(self, k):
        """The k-th derived tensor; runs the deferred sampling first (lazy_derived).  The deferred tensors are derived from the
        parameters as they are when first READ.  The loop that switched laziness on only changes the curve tensors through its
        optimizer step, which is followed by the next prepare_scaling_rot(), so that is the state the call saw -- except when
        a parameter's STORAGE was replaced with equal values in between (reset_opacity / a topology edit rebuilding the flat
        buffers): the stamp then names the storage that was actually sampled, and render() may keep its fused route."""
        pend = self._derived_pending
        if pend is not None:
            self._derived_pending = None
            # (grad mode: the prepare_scaling_rot() call's, or the reader's -- a topology edit made under no_grad followed by a
            # training render must still reach the curve parameters through the derived tensors)
            with torch.set_grad_enabled(pend[1] or torch.is_grad_enabled()):
                self._derived = list(curve_sampling.sample_curves(self._curve_points, self._width, self.is_bezier,
                                                                  self.n_gaussians, pend[0]))
            self._derived_from = self._param_stamp()
        return self._derived[k]

    def _set_derived(self, k, v):
        if self._derived_pending is not None:
            self._derive(k)                      # materialise the other two before one of the three is replaced
        self._derived[k] = v

    _xyz = property(lambda self: self._derive(0), lambda self, v: self._set_derived(0, v))
    _rotation = property(lambda self: self._derive(1), lambda self, v: self._set_derived(1, v))
    _scaling = property(lambda self: self._derive(2), lambda self, v: self._set_derived(2, v))

    @property
    def n_splats(self):
        """Number of splats (rows of `_xyz`) without touching the derived tensors."""
        return int(self._curve_points.shape[0]) * self.n_gaussians if self._curve_points.dim() == 3 else 0

    # ------------------------------------------------------------------ accessors (:66-140)
    @property
    def get_curve_points(self):
        return self._curve_points

    @property
    def get_scaling(self):
        return self._scaling

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity.unsqueeze(1).expand(-1, self.n_gaussians, -1).reshape(-1, 1))

    @property
    def get_curve_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_curve_width(self):
        return self.scaling_activation(self._width)

    def get_covariance(self, scaling_modifier=1):
        """scene/gaussian_model.py:32-36,184-185 with utils/general_utils.py:134-181: Sigma = (R S)(R S)^T of the
        normalised quaternion and `scaling_modifier * scaling`, packed as the upper triangle [xx, xy, xz, yy, yz, zz]
        (pipe.compute_cov3D_python, gaussian_renderer/__init__.py:67-68)."""
        q = self._rotation / torch.sqrt((self._rotation * self._rotation).sum(-1, keepdim=True))
        r, x, y, z = q.unbind(-1)
        R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), -1).view(-1, 3, 3)
        Lm = R * (scaling_modifier * self.get_scaling).unsqueeze(1)          # R @ diag(s)
        cov = Lm @ Lm.transpose(1, 2)
        return torch.stack((cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]), -1)

    @property
    def get_features(self):
        return torch.cat((self._features_dc.flatten(0, 1), self._features_rest.flatten(0, 1)), dim=1)

    @property
    def get_rotation_matrix(self):
        return curve_sampling.quaternion_to_matrix(self.get_rotation)

    def get_main_axis(self, view_cam):
        """:99-105 (the in-place masked negation is written as a where; same values)."""
        d = self.get_rotation_matrix[..., 0]
        to_cam = view_cam.camera_center - self._xyz
        neg = (d * to_cam).sum(-1) < 0.0
        return torch.where(neg[:, None], -d, d)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """scene/gaussian_model.py:618-620 -- consumer of means2D.grad[:, :2] (NDC-scaled, quirk 9)."""
        if not hasattr(self, "xyz_gradient_accum") or self.xyz_gradient_accum.shape[0] != self.n_splats:
            dev = self._curve_points.device
            self.xyz_gradient_accum = torch.zeros((self.n_splats, 1), device=dev)
            self.denom = torch.zeros((self.n_splats, 1), device=dev)
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1,
                                                             keepdim=True)
        self.denom[update_filter] += 1


class Scene:
    """What the reference's ``Scene(args, gaussians)`` (scene/__init__.py:27-92) does for an EMAP scan, minus the file
    copies: read the cameras (dataset_io.read_emap = readEMAP + loadCam), build the seed cloud of rendemapInfo
    (dataset_readers.py:404-441: the 15^3 grid when ``init_random_init``) and call ``gaussians.create_from_pcd``."""

    def __init__(self, source_path, gaussians, detector="DexiNed", num_pts_per_axis=15, cameras_extent=None, rng=None,
                 device=None):
        from . import dataset_io
        self.gaussians = gaussians
        self.train_cameras = dataset_io.read_emap(source_path, detector=detector)
        self.point_cloud = dataset_io.grid_point_cloud(num_pts_per_axis, rng)
        if cameras_extent is None:   # getNerfppNorm (dataset_readers.py:46-67): 1.1 x the largest distance to the mean centre
            centres = torch.stack([c.camera_center for c in self.train_cameras]).double()
            cameras_extent = float((centres - centres.mean(0)).norm(dim=1).max() * 1.1)
        self.cameras_extent = cameras_extent
        if device is not None:
            self.train_cameras = [c.to(device) for c in self.train_cameras]
        gaussians.create_from_pcd(self.point_cloud, self.train_cameras, self.cameras_extent)

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras


def _install_topology():
    from . import topology
    for name, fn in topology.METHODS.items():
        setattr(GaussianCurveModel, name, fn)

    def add_topology_listener(self, cb):
        """cb() runs after every topology edit (the train step rebinds its flat buffers / drops its captured graph)."""
        self.__dict__.setdefault("_topology_listeners", []).append(cb)
    GaussianCurveModel.add_topology_listener = add_topology_listener


_install_topology()
