"""Per-iteration training sequence of the reference (train.py:75-148,226-243) on the HIP hot path:

    update lr -> pick view -> render (fused attrs + rasterizer) -> edge_aware_loss + fused_ssim -> backward
    -> [view-parallel all-reduce] -> Adam step -> zero grads -> prepare_scaling_rot

Only the photometric terms and (from ``densify_until_iter``) the mask regulariser are included; the O(B^2) connection
loss and the topology edits are out of scope."""
import os
import random

import torch

from .fused_ssim import fused_ssim
from .gaussian_renderer import PipelineParams, render
from .ops.losses import edge_aware_loss, photometric_loss, unit_grad
from .ops.optim import FlatAdam
from .view_parallel import FlatGrads, StaticCamera as _StaticCamera, no_gc


class TrainStep:
    def __init__(self, gaussians, cameras, gt_images, lambda_mse=10.0, lambda_dssim=0.1, lambda_mask=0.0005,
                 densify_until_iter=7000, mask_threshold=0.01, seed=0, rank=0, world=1, fused=True,
                 regularisers=False, opacity_loss_weight=0.01, lambda_curve_smo=0.1, lambda_width=0.01,
                 lambda_points_conn=0.1, conn_from_iter=7000, direct=False):
        self.g = gaussians
        # direct=True (fused only, compiled host shim): the iteration without autograd -- the checked view forward, the
        # photometric loss kernels, the view backward (adding into the flat gradient buffer), Adam and prepare_scaling_rot
        # called one after the other.  Same kernels and numbers as the autograd form; a third of its host time.  Unlike
        # GraphedTrainStep nothing is captured: cameras may differ in size and field of view, binning stays exact.
        self._eager_direct = bool(direct)
        self._direct_ws = {}
        self.cams = cameras
        self.gts = gt_images                      # list of [1,H,W] edge maps on the device
        self.lambda_mse, self.lambda_dssim, self.lambda_mask = lambda_mse, lambda_dssim, lambda_mask
        self.densify_until_iter, self.mask_threshold = densify_until_iter, mask_threshold
        # train.py:113-131 (off by default: the BASELINE train-step metric excludes them)
        self.regularisers = regularisers
        self.opacity_loss_weight, self.lambda_curve_smo, self.lambda_width = opacity_loss_weight, lambda_curve_smo, lambda_width
        # train.py:133-146: end-point connection loss, from iteration conn_from_iter + 1 on (part of `regularisers`)
        self.lambda_points_conn, self.conn_from_iter = lambda_points_conn, conn_from_iter
        self.reset_timestep = 0    # train.py:74-76: 0 before the loop, += 1 at the top of EVERY iteration -- the
                                   # `reset_timestep > 0` gate of the opacity term (train.py:114) is true from iteration 1
        self.pipe = PipelineParams()
        self.bg = torch.zeros(3, device=gaussians.device)
        self.rng = random.Random(seed)            # the SAME stream on every rank: rank r takes the r-th of each group of
                                                  # `world` draws (train.py:85-90 pops one random view per iteration)
        self.stack = []
        self.rank, self.world = rank, world
        if gaussians.optimizer is None:
            gaussians.training_setup()
        named = {"curve_points": gaussians._curve_points, "width": gaussians._width, "opacity": gaussians._opacity,
                 "mask": gaussians._mask, "f_dc": gaussians._features_dc, "f_rest": gaussians._features_rest}
        self.flat = FlatGrads(named)
        self.fused = fused
        if fused:   # one-launch Adam with the reference's per-group learning rates (training_setup :203-213)
            from .scene.gaussian_curve_model import _load_optimizer_state, _optimizer_state
            lrs = {g["name"]: g["lr"] for g in gaussians.optimizer.param_groups}
            carried = _optimizer_state(gaussians.optimizer)     # a run resumed with restore(): moments and step count
            gaussians.optimizer = FlatAdam(named, lrs, self.flat, eps=1e-15)
            if carried is not None and carried["step"] > 0:
                _load_optimizer_state(gaussians.optimizer, carried)
            # this loop owns the parameter updates (optimizer step -> prepare_scaling_rot, train.py:235-243): the derived splat
            # tensors are only computed when something reads them.  The fused render route samples inside its own kernels, so
            # an iteration runs the grid-wide norm pass of prepare_scaling_rot once instead of twice.
            gaussians.lazy_derived = True
            gaussians.prepare_scaling_rot()   # parameters moved into the flat buffer: rebuild the derived tensors
        self.iteration = 0
        if hasattr(gaussians, "add_topology_listener"):
            gaussians.add_topology_listener(self._on_topology_change)

    def _on_topology_change(self):
        """The per-curve tensors were resized (scene/topology.py): rebind the flat gradient buffer."""
        g = self.g
        named = {"curve_points": g._curve_points, "width": g._width, "opacity": g._opacity, "mask": g._mask,
                 "f_dc": g._features_dc, "f_rest": g._features_rest}
        if self.fused:
            self.flat = g.optimizer.grads          # FlatAdam.rebuild made new flat buffers and views
        else:
            self.flat = FlatGrads(named)

    def _conn_active(self, iteration):
        """train.py:133: `opt.lambda_points_conn > 0 and iteration > opt.conn_from_iter`."""
        return self.regularisers and self.lambda_points_conn > 0 and iteration > self.conn_from_iter

    def _regulariser_terms(self, radii, opacity_gate, with_conn=False):
        """train.py:113-146; opacity_gate (float or device scalar) switches the opacity term (reset_timestep > 0)."""
        from .ops import regularizers as RG
        g = self.g
        if self.fused:   # one HIP op (three launches)
            reg = RG.curve_regularizers(g, radii, self.opacity_loss_weight, opacity_gate, self.lambda_curve_smo,
                                        self.lambda_width)
            if with_conn:
                reg = reg + RG.connection_loss(g, self.lambda_points_conn)
            return reg
        reg = RG.opacity_loss(g, radii, self.opacity_loss_weight) * opacity_gate
        if self.lambda_curve_smo > 0:
            reg = reg + RG.curve_smoothness_loss(g, radii, self.lambda_curve_smo)
        if self.lambda_width > 0:
            reg = reg + RG.width_loss(g, self.lambda_width)
        if with_conn:
            reg = reg + RG.connection_loss_reference(g, self.lambda_points_conn)
        return reg

    def _next_view(self):
        mine = None
        for r in range(max(1, self.world)):      # every rank draws the whole group: the streams stay in lock step
            if not self.stack:
                self.stack = list(range(len(self.cams)))
            v = self.stack.pop(self.rng.randint(0, len(self.stack) - 1))   # train.py:85-90
            if r == self.rank:
                mine = v
        return mine

    def step(self, view_index=None):
        g = self.g
        self.iteration += 1
        self.reset_timestep += 1                   # train.py:76
        it = self.iteration
        g.update_learning_rate(it)
        vi = self._next_view() if view_index is None else view_index
        cam, gt = self.cams[vi], self.gts[vi]
        use_mask = it >= self.densify_until_iter
        if self._eager_direct and self.fused:
            return self._step_direct(cam, gt, use_mask)
        pkg = render(cam, g, self.pipe, self.bg, use_mask=use_mask, mask_thr=self.mask_threshold,
                     compute_visibility=not self.fused, clamp=not self.fused, compute_rend_dir=not self.fused,
                     grad_sinks=self.fused)   # (fused: the backward kernels add into the flat gradient buffer themselves)
        image = pkg["render"]
        if self.fused:   # raw composite in, render()'s clamp applied inside the loss kernels
            loss = photometric_loss(image, gt[:1], self.lambda_mse, self.lambda_dssim, clamp=True)
        else:
            Ll1 = edge_aware_loss(image, gt[:1])
            ssim_value = fused_ssim(image.unsqueeze(0), gt[:1].unsqueeze(0))
            loss = self.lambda_mse * ((1.0 - self.lambda_dssim) * Ll1 + self.lambda_dssim * (1.0 - ssim_value))
        if use_mask:
            loss = loss + self.lambda_mask * torch.mean(torch.sigmoid(g._mask))
        if self.regularisers:
            loss = loss + self._regulariser_terms(pkg["radii"], 1.0 if self.reset_timestep > 0 else 0.0,
                                                  with_conn=self._conn_active(self.iteration))
        loss.backward(gradient=unit_grad(loss.device) if self.fused else None)
        self.flat.all_reduce()
        if self.fused:
            g.optimizer.step(zero_grad=True)       # Adam + zero_grad in one launch
        else:
            g.optimizer.step()
            self.flat.zero_()                      # grads are views of the flat buffer: keep them, zero in place
        g.prepare_scaling_rot()                    # train.py:242-243
        return loss.detach(), pkg


    def _step_direct(self, cam, gt, use_mask):
        """TrainStep.step() from the render on, without autograd (TrainStep(direct=True)): train.py:95-107, :110-146, :235,
        :242-243 as eight library calls.  The forward is the checked one (exact binning: a bucket overflow re-renders with the
        raised capacity before anything reached the gradient buffer)."""
        import ctypes as C
        import math
        from . import _lib as L
        from .ops import view_render as VR
        from .ops.curve_sampling import _bezier_mask, sample_coefficients
        from .ops.losses import edge_pixel_count
        g = self.g
        lib = L.load()
        if not L.use_shim():
            raise L.CurveGSError("TrainStep(direct=True) needs the compiled host shim (curve_gaussian_amd/_cgs_torch.so)")
        from .gaussian_renderer import _fused_route_ok
        if not _fused_route_ok(g, self.pipe, 1.0, None):
            # a parameter tensor was replaced since the last prepare_scaling_rot (reset_opacity, a topology edit made outside
            # the loop): train.py refreshes the derived tensors at the end of every iteration (:242-243), so they are
            # current whenever its next render starts -- same state here
            g.prepare_scaling_rot()
        dev = g._curve_points.device
        m = g.n_gaussians
        cp, wl, ol = g._curve_points.detach(), g._width.detach(), g._opacity.detach()
        mask = g._mask.detach() if use_mask else None
        B = cp.shape[0]
        P = B * m
        H, W = int(cam.image_height), int(cam.image_width)
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        isb, coef = _bezier_mask(g.is_bezier, dev), sample_coefficients(m, dev)
        eps = getattr(g, "_derived_eps", 1e-8)
        p, cf = L.ptr, C.c_float
        gt1 = gt[:1].detach().float().contiguous()
        n_pos = edge_pixel_count(gt1)
        with L.device_guard(dev):
            st = L.raw_stream(dev)
            key = (str(dev), H, W, st)
            ws = self._direct_ws.get(key)
            if ws is None:
                if len(self._direct_ws) >= 8:
                    self._direct_ws.pop(next(iter(self._direct_ws)))
                ws = self._direct_ws[key] = dict(
                    photo=torch.zeros(int(lib.cgs_photometric_workspace_bytes(H, W)), dtype=torch.uint8, device=dev),
                    reg=torch.zeros(int(lib.cgs_curve_regularizers_workspace_bytes()), dtype=torch.uint8, device=dev),
                    gate={v: torch.full((1,), v, dtype=torch.float32, device=dev) for v in (0.0, 1.0)})
            a, bb = self.lambda_mse * (1.0 - self.lambda_dssim), self.lambda_mse * self.lambda_dssim
            while True:
                cap = VR._capacity(lib, dev, P, W, H)
                color, invd, amap, radii, _dir, _raw, saved, handle = L.shim().view_forward(
                    cp, wl, ol, mask, isb, coef, m, self.mask_threshold, self.bg, cam.world_view_transform, cam.full_proj_transform,
                    cam.camera_center, tanx, tany, H, W, cap, False, False, False, eps)
                pend = VR.Pending(handle, (dev.index, P, W, H), cap, saved[6])
                # value and d loss / d image behind the compositor (render()'s clamp inside the loss kernels), then the wait for
                # the forward's 16-byte status readback -- the compositor and the loss are already queued behind it
                g_img = torch.empty_like(color)
                loss = torch.empty((), dtype=torch.float32, device=dev)
                L.check(lib.cgs_photometric_loss(H, W, p(color), p(gt1), cf(0.1), p(n_pos), cf(a), cf(bb), 1, p(ws["photo"]), p(g_img),
                                                 p(loss), st), "cgs_photometric_loss")
                ok, _nvis = VR.finish(pend)
                if ok:
                    break
            grads = self.flat
            sinks = [grads.view("curve_points"), grads.view("width"), grads.view("opacity")] + ([grads.view("mask")] if use_mask else [])
            extra = None
            if self.regularisers:   # (train.py:113-131) forward quantities only: queued before the backward, which takes dL/drotation
                f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
                r_rot, r_op, r_w, reg_loss = f32(P, 4), f32(B, 1), f32(B, 1), f32(())
                gate = ws["gate"][1.0 if self.reset_timestep > 0 else 0.0]
                L.check(lib.cgs_curve_regularizers(B, m, p(g._rotation.detach()), p(ol), p(wl), p(radii), cf(self.opacity_loss_weight),
                                                   p(gate), cf(self.lambda_curve_smo), cf(self.lambda_width), cf(0.005), p(ws["reg"]),
                                                   p(reg_loss), p(r_rot), p(r_op), p(r_w), st), "cgs_curve_regularizers")
                extra = r_rot
                loss = loss + reg_loss
            _n, _n, _n, _n, g_m2d = L.shim().view_backward(*saved[:4], isb, coef, *saved[4:], m, self.mask_threshold, tanx, tany, H, W,
                                                           eps, g_img, None, sinks, extra)
            if self.regularisers:
                sinks[2].add_(r_op)
                sinks[1].add_(r_w)
                if self._conn_active(self.iteration):
                    conn_ws = torch.zeros(int(lib.cgs_endpoint_connection_workspace_bytes(B)), dtype=torch.uint8, device=dev)
                    conn_loss = torch.zeros((), dtype=torch.float32, device=dev)
                    L.check(lib.cgs_endpoint_connection_loss(B, p(cp), cf(0.05), cf(self.lambda_points_conn), p(conn_ws), p(conn_loss),
                                                             p(sinks[0]), 1, st), "cgs_endpoint_connection_loss")
                    loss = loss + conn_loss
            if use_mask:            # train.py:110-111: lambda_mask * mean(sigmoid(mask)), gradient added by hand
                sg = torch.sigmoid(mask)
                loss = loss + self.lambda_mask * sg.mean()
                sinks[3].add_(sg * (1 - sg), alpha=self.lambda_mask / mask.numel())
        self.flat.all_reduce()
        g.optimizer.step(zero_grad=True)
        g.prepare_scaling_rot()
        pkg = {"render": color, "viewspace_points": _GradHolder(g_m2d), "visibility_filter": None, "radii": radii, "depth": invd,
               "rend_dir": None, "rend_alpha": amap[3:4]}
        return loss, pkg


class _GradHolder:
    """Stands where render()'s `viewspace_points` leaf stands in the result dict: `.grad` is dL/dmeans2D [P,3] of the view
    (what add_densification_stats reads, scene/gaussian_curve_model.py:604-607 of the reference)."""

    def __init__(self, grad):
        self.grad = grad


class GraphedTrainStep(TrainStep):
    """The same iteration as ``TrainStep(fused=True)`` replayed as ONE hipGraph launch.

    An eager iteration costs ~30 kernel launches through Python autograd, ctypes and the HIP runtime (~0.4 ms of host
    time); small scenes are bound by that, large ones leave the GPU idle between kernels.  Here the whole sequence
    (render with the sync-free forward -> fused photometric loss -> backward -> Adam -> prepare_scaling_rot) is captured
    once; per step the host sends ONE 432-byte pinned staging slot (Adam scalars, camera pose, edge-pixel count, view
    index) with a stream-ordered copy and replays the graph; the gt edge map of the step is picked on the device from
    a [V,H,W] stack (the autograd body, ``direct=False``, still copies it into a staging image).

    The captured forward bins into fixed-capacity tile buckets sized from an eager probe (x ``cap_margin``).  If a
    bucket still overflows, the device-side flag makes the captured Adam skip its update (gradients are cleared);
    the host sees the flag one step later, redoes that view eagerly through the exact path, enlarges the buckets and
    re-captures.  All cameras must share the image size and field of view (they are graph constants).

    ``step()`` returns the graph's static loss tensor: it is overwritten by the next step (copy or ``float()`` it
    to keep a value).  Call ``finish()`` before reading the model from outside or editing its topology."""

    def __init__(self, *args, cap_margin=1.5, direct=True, collectives=None, fused_view=True, aux_outputs=True,
                 capture_collectives=False, **kw):
        kw["fused"] = True
        super().__init__(*args, **kw)
        # aux_outputs=False (fused direct body only): the forward writes `render` alone -- a training iteration reads nothing
        # else (train.py:98-107); `last["depth"]` / `last["all_map"]` are then None.  Default: every output of render().
        self.aux_outputs = bool(aux_outputs)
        # direct body only: cgs_view_forward / cgs_view_backward (the per-splat chains fused, csrc/view.hip) instead of the
        # sampling -> attributes -> rasterizer calls one by one; same results, five launches fewer per iteration
        self.fused_view = bool(fused_view)
        # View-parallel runs (world > 1; `collectives=True` forces the same code path on a single-rank group): the graph
        # ends before the optimizer; the host then all-reduces the flat gradient buffer and the overflow flag (max) and
        # launches the Adam kernel.  Overflow handling is made deterministic across ranks by looking at the flag of
        # iteration k - 2 (blocking, long finished) at the start of iteration k, so every rank redoes the same iteration
        # at the same point of its collective sequence.
        self._collective = (self.world > 1) if collectives is None else bool(collectives)
        # capture_collectives=True: the two all-reduces and the Adam kernel are captured with the rest of the iteration (one
        # graph replay per step, no host touch between backward and optimizer).  Opt-in and UNVERIFIED for world > 1: exercised
        # on a single-rank RCCL group only, there also across forced bucket overflows (no multi-GPU box in the build
        # environment).  With more ranks every rank must run the same number of warm-up executions and re-captures -- the
        # captured body issues collectives -- which the fixed two-iteration overflow lag is designed to guarantee.
        self._capture_coll = bool(capture_collectives) and self._collective
        if self._capture_coll and self.world > 1 and os.environ.get("CGS_ALLOW_CAPTURED_COLLECTIVES") != "1":
            # never run with two ranks (no multi-GPU box in the build environment): opt in explicitly
            raise ValueError("GraphedTrainStep(capture_collectives=True) has only been exercised on a single-rank RCCL group; "
                             "set CGS_ALLOW_CAPTURED_COLLECTIVES=1 to use it with world > 1")
        if self._collective:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise ValueError("GraphedTrainStep: view-parallel mode needs an initialised process group")
            if not direct:
                raise ValueError("GraphedTrainStep: view-parallel mode uses the direct body")
        dev = self.g.device
        c0 = self.cams[0]
        for c in self.cams:
            if (c.image_height, c.image_width, c.FoVx, c.FoVy) != (c0.image_height, c0.image_width, c0.FoVx, c0.FoVy):
                raise ValueError("GraphedTrainStep: all cameras must share image size and field of view")
        self.cap_margin = float(cap_margin)
        self.direct = bool(direct)   # True: the captured sequence calls the C ABI directly (no autograd inside the graph)
        self._bufs = None
        # per-step inputs of the graph: camera pose (35 floats) | edge-pixel count (int32 bits) | view index (int32),
        # 160 bytes per view kept as host bytes; they travel behind the Adam scalars in the optimizer's ONE pinned
        # staging copy per step (FlatAdam.stage_step(extra=...)) and land in the tail of its device state, which the
        # captured kernels read.  The gt edge maps live in one [V,H,W] stack and the direct body picks the step's map
        # on the device (cgs_photometric_loss_indexed): no per-step 4*H*W-byte copy.
        import struct
        from .ops.losses import edge_pixel_count
        self._cam = _StaticCamera(c0, dev)
        H, W = int(c0.image_height), int(c0.image_width)
        self._gt_stack = torch.stack([g[:1].reshape(H, W).float() for g in self.gts]).contiguous()
        self.gts = [self._gt_stack[v:v + 1] for v in range(len(self.gts))]          # same storage, eager path
        self._npos_table = torch.cat([edge_pixel_count(g) for g in self.gts]).to(torch.int32).contiguous()
        npos_host = self._npos_table.cpu().tolist()
        self._host_packs = []
        for v, c in enumerate(self.cams):
            pose = _StaticCamera.packed(c).cpu().numpy().astype("<f4").tobytes()
            self._host_packs.append(pose + struct.pack("<ii", int(npos_host[v]), v) + b"\0" * (self.INPUT_BYTES - 148))
        self._bind_inputs()
        self._gt = torch.empty_like(self.gts[0][:1]).contiguous()      # staging copy, autograd body (direct=False) only
        self._opa_gate = torch.zeros((), dtype=torch.float32, device=dev)
        self._gate_value = 0.0
        self._graph = None
        self._cap = 0
        self._loss = None
        self._status = None
        self._use_mask = False
        self._use_conn = False      # graph constant like _use_mask: the switch at conn_from_iter re-captures
        self._flag_host = torch.zeros(64, dtype=torch.int32).pin_memory()
        # single-GPU replays: the captured Adam kernel reports "this iteration was skipped" straight into that pinned ring
        # (entry = its own execution count, kept in device memory, modulo 64) -- no device-to-host copy queued between one
        # replay and the next.  _report_next mirrors the device counter on the host.
        self._report_seq = torch.zeros(1, dtype=torch.int32, device=self.g.device)
        self._report_next = 0
        self._inflight = []   # (event, slot, view index, iteration)
        self.recaptures = 0
        self._t0 = self.g.optimizer.step_count - self.iteration   # Adam step number = _t0 + iteration

    # -- the captured sequence ---------------------------------------------------------------------------------
    def _body(self):
        if self.direct:
            return self._body_direct()
        g = self.g
        sink = []
        # prepare_scaling_rot opens the captured sequence (the reference runs it at the END of the previous iteration,
        # train.py:242-243 -- same data flow, but every replay must read the parameters, not a tensor of an older replay)
        g.prepare_scaling_rot()
        pkg = render(self._cam, g, self.pipe, self.bg, use_mask=self._use_mask, mask_thr=self.mask_threshold,
                     compute_visibility=False, clamp=False, compute_rend_dir=False,
                     static_bucket_cap=self._cap, status_sink=sink)
        loss = photometric_loss(pkg["render"], self._gt, self.lambda_mse, self.lambda_dssim, clamp=True, n_pos=self._npos)
        if self._use_mask:      # train.py:110-111 (a graph constant: the switch at densify_until_iter re-captures)
            loss = loss + self.lambda_mask * torch.mean(torch.sigmoid(g._mask))
        if self.regularisers:   # sync-free torch ops; the opacity term is gated by a device scalar refreshed per step
            loss = loss + self._regulariser_terms(pkg["radii"], self._opa_gate, with_conn=self._use_conn)
        loss.backward(gradient=unit_grad(loss.device))
        status = sink[0]
        g.optimizer.step_dev(zero_grad=True, skip_flag=status[2:3], report=self._report())
        return loss.detach(), status

    # -- the same sequence without autograd: every kernel of the iteration called through the C ABI on preallocated
    # buffers, gradients written straight into the flat gradient buffer.  Inside a graph the autograd bookkeeping
    # costs nothing on the host, but it does cost GPU launches (gradient accumulation adds, ones/zeros fills,
    # grad * 1 multiplies: ~10 of 41 launches, ~45 us at cfg3).
    def _alloc_direct(self):
        import ctypes as C
        from . import _lib as L
        from .ops.curve_sampling import sample_coefficients, _bezier_mask
        lib = L.load()
        g = self.g
        dev = g.device
        m = g.n_gaussians
        B = g._curve_points.shape[0]
        P = B * m
        H, W = self._cam.image_height, self._cam.image_width
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        b = dict(B=B, P=P, m=m, H=H, W=W)
        b["coef"] = sample_coefficients(m, dev)
        b["isb"] = _bezier_mask(g.is_bezier, dev)
        b["norms"] = torch.empty(384, dtype=torch.float64, device=dev)
        b["xyz"], b["rot"], b["scl"] = f(P, 3), f(P, 4), f(P, 3)
        b["rot_n"], b["opac"], b["amap"], b["scl_m"] = f(P, 4), f(P, 1), f(P, 4), f(P, 3)
        b["colors"] = torch.ones(P, 1, device=dev)
        b["geom"] = u8(lib.cgs_geometry_bytes(P))
        b["nbin"] = int(lib.cgs_binning_bytes(self._cap * tiles))
        b["bin"] = u8(b["nbin"])
        b["img"] = u8(lib.cgs_image_bytes(W, H))
        off, n = int(lib.cgs_image_status_offset(W, H)), int(lib.cgs_status_words())
        b["status"] = b["img"][off:off + 4 * n].view(torch.int32)
        b["color"], b["invd"], b["omap"] = f(1, H, W), f(1, H, W), f(4, H, W)
        b["radii"] = torch.empty(P, dtype=torch.int32, device=dev)
        b["photo_ws"] = torch.zeros(int(lib.cgs_photometric_workspace_bytes(H, W)), dtype=torch.uint8, device=dev)
        b["g_img"] = f(1, H, W)
        b["loss"] = torch.zeros((), dtype=torch.float32, device=dev)
        b["g_m2d"], b["g_conic"], b["g_opac"] = f(P, 3), f(P, 2, 2), f(P, 1)
        b["g_m3d"], b["g_cov"], b["g_scl"], b["g_rotn"], b["g_amap"] = f(P, 3), f(P, 6), f(P, 3), f(P, 4), f(P, 4)
        b["g_rot_raw"], b["g_scaling"], b["gv"] = f(P, 4), f(P, 3), f(P, 9)
        b["view_scratch"] = f(int(lib.cgs_view_backward_scratch_floats(B, m)))
        b["reg_ws"] = torch.zeros(int(lib.cgs_curve_regularizers_workspace_bytes()), dtype=torch.uint8, device=dev)
        b["reg_loss"] = torch.zeros((), dtype=torch.float32, device=dev)
        b["r_rot"], b["r_op"], b["r_w"] = f(P, 4), f(B, 1), f(B, 1)
        b["mask_loss"] = torch.zeros((), dtype=torch.float32, device=dev)
        b["conn_ws"] = torch.zeros(int(lib.cgs_endpoint_connection_workspace_bytes(B)), dtype=torch.uint8, device=dev)
        b["conn_loss"] = torch.zeros((), dtype=torch.float32, device=dev)
        b["bg"] = self.bg.float().contiguous()
        self._bufs = b
        return b

    def _body_direct(self):
        import ctypes as C
        import math
        from . import _lib as L
        lib = L.load()
        g = self.g
        b = self._bufs if self._bufs is not None else self._alloc_direct()
        B, P, m, H, W = b["B"], b["P"], b["m"], b["H"], b["W"]
        p, cf, s = L.ptr, C.c_float, L.raw_stream(g.device)
        grads = g.optimizer.grads                       # FlatGrads: .view(name) are the parameters' .grad
        cp, wl, ol = g._curve_points.detach(), g._width.detach(), g._opacity.detach()
        mask = g._mask.detach() if self._use_mask else None
        cam = self._cam
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        chk = L.check
        if self.fused_view:
            return self._body_direct_fused(lib, b, cam, cp, wl, ol, mask, grads, tanx, tany, s)
        # ---- forward: curves -> splats -> per-view attributes -> rasterizer (sync-free) -> loss
        chk(lib.cgs_sample_curves_forward(B, m, p(cp), p(wl), p(b["isb"]), p(b["coef"]), cf(1e-8), p(b["norms"]),
                                          p(b["xyz"]), p(b["rot"]), p(b["scl"]), s), "sample_curves_forward")
        chk(lib.cgs_splat_attrs_forward(B, m, p(b["rot"]), p(b["xyz"]), p(ol), p(mask), cf(self.mask_threshold),
                                        p(b["scl"]), p(cam.camera_center), p(cam.world_view_transform), p(b["rot_n"]),
                                        p(b["opac"]), p(b["scl_m"]) if mask is not None else None, p(b["amap"]), s),
            "splat_attrs_forward")
        scales = b["scl_m"] if mask is not None else b["scl"]
        chk(lib.cgs_rasterize_forward_static(
            p(b["geom"]), p(b["bin"]), b["nbin"], p(b["img"]), self._cap, P, 0, 0, p(b["bg"]), W, H, p(b["xyz"]), None,
            p(b["colors"]), p(b["opac"]), p(scales), 1.0, p(b["rot_n"]), None, p(b["amap"]),
            p(cam.world_view_transform), p(cam.full_proj_transform), p(cam.camera_center), tanx, tany, p(b["color"]),
            p(b["invd"]), p(b["omap"]), 0, 1, p(b["radii"]), s), "rasterize_forward_static")
        a = self.lambda_mse * (1.0 - self.lambda_dssim)
        bb = self.lambda_mse * self.lambda_dssim
        chk(lib.cgs_photometric_loss_indexed(H, W, p(b["color"]), p(self._gt_stack), p(self._view_idx), cf(0.1),
                                             p(self._npos_table), cf(a), cf(bb), 1, p(b["photo_ws"]), p(b["g_img"]),
                                             p(b["loss"]), s), "photometric_loss_indexed")
        # ---- backward: rasterizer -> attributes -> sampling, straight into the flat gradient views
        chk(lib.cgs_rasterize_backward(
            P, 0, 0, 1, p(b["bg"]), W, H, p(b["xyz"]), None, p(b["colors"]), p(b["amap"]), p(b["opac"]), p(scales), 1.0,
            p(b["rot_n"]), None, p(cam.world_view_transform), p(cam.full_proj_transform), p(cam.camera_center), tanx,
            tany, p(b["radii"]), p(b["geom"]), p(b["bin"]), p(b["img"]), p(b["g_img"]), None, None, p(b["g_m2d"]),
            p(b["g_conic"]), p(b["g_opac"]), None, None, p(b["g_m3d"]), p(b["g_cov"]), None, p(b["g_scl"]),
            p(b["g_rotn"]), p(b["g_amap"]), 0, 1, 0, s), "rasterize_backward")
        chk(lib.cgs_splat_attrs_backward(
            B, m, p(b["rot"]), p(b["xyz"]), p(ol), p(mask), cf(self.mask_threshold), p(b["scl"]), p(cam.camera_center),
            p(cam.world_view_transform), p(b["g_rotn"]), p(b["g_opac"]), p(b["g_scl"]) if mask is not None else None,
            p(b["g_amap"]), p(b["g_rot_raw"]), p(grads.view("opacity")), p(grads.view("mask")) if mask is not None else None,
            p(b["g_scaling"]) if mask is not None else None, s), "splat_attrs_backward")
        g_scaling = b["g_scaling"] if mask is not None else b["g_scl"]
        loss = b["loss"]
        if self.regularisers:
            chk(lib.cgs_curve_regularizers(B, m, p(b["rot"]), p(ol), p(wl), p(b["radii"]), cf(self.opacity_loss_weight),
                                           p(self._opa_gate), cf(self.lambda_curve_smo), cf(self.lambda_width), cf(0.005),
                                           p(b["reg_ws"]), p(b["reg_loss"]), p(b["r_rot"]), p(b["r_op"]), p(b["r_w"]), s),
                "curve_regularizers")
            b["g_rot_raw"].add_(b["r_rot"])
            grads.view("opacity").add_(b["r_op"])
            loss = loss + b["reg_loss"]
        chk(lib.cgs_sample_curves_backward(B, m, p(cp), p(wl), p(b["isb"]), p(b["coef"]), cf(1e-8), p(b["norms"]),
                                           p(b["g_m3d"]), p(b["g_rot_raw"]), p(g_scaling), p(grads.view("curve_points")),
                                           p(grads.view("width")), p(b["gv"]), s), "sample_curves_backward")
        if self.regularisers:
            grads.view("width").add_(b["r_w"])
            if self._use_conn:   # adds to the curve-point gradient the sampling backward just wrote
                chk(lib.cgs_endpoint_connection_loss(B, p(cp), cf(0.05), cf(self.lambda_points_conn), p(b["conn_ws"]),
                                                     p(b["conn_loss"]), p(grads.view("curve_points")), 1, s),
                    "endpoint_connection_loss")
                loss = loss + b["conn_loss"]
        if self._use_mask:      # train.py:110-111: lambda_mask * mean(sigmoid(mask)), gradient added by hand
            sg = torch.sigmoid(mask)
            loss = loss + self.lambda_mask * sg.mean()
            grads.view("mask").add_(sg * (1 - sg), alpha=self.lambda_mask / mask.numel())
        status = b["status"]
        if self._capture_coll:
            import torch.distributed as dist
            dist.all_reduce(g.optimizer.grads.flat)
            dist.all_reduce(status[2:3], op=dist.ReduceOp.MAX)   # any rank overflowed -> every rank skips
        if not self._collective or self._capture_coll:   # view-parallel: the all-reduce sits between backward and optimizer
            g.optimizer.step_dev(zero_grad=True, skip_flag=status[2:3], report=self._report())
        self.last = dict(radii=b["radii"], dL_dmeans2D=b["g_m2d"], render=b["color"], depth=b["invd"], all_map=b["omap"])
        return loss, status

    def _body_direct_fused(self, lib, b, cam, cp, wl, ol, mask, grads, tanx, tany, s):
        """The direct body on the fused per-view entry points: 8 launches for render + its backward instead of 13."""
        import ctypes as C
        from . import _lib as L
        g = self.g
        B, P, m, H, W = b["B"], b["P"], b["m"], b["H"], b["W"]
        p, cf, chk = L.ptr, C.c_float, L.check
        chk(lib.cgs_view_forward(
            B, m, p(cp), p(wl), p(b["isb"]), p(b["coef"]), cf(1e-8), p(b["norms"]), p(ol), p(mask), cf(self.mask_threshold),
            None, p(b["geom"]), p(b["bin"]), b["nbin"], p(b["img"]), self._cap, p(b["bg"]), W, H,
            p(cam.world_view_transform), p(cam.full_proj_transform), p(cam.camera_center), tanx, tany, p(b["color"]),
            p(b["invd"]) if self.aux_outputs else None, p(b["omap"]) if self.aux_outputs else None, p(b["radii"]),
            p(b["xyz"]), p(b["rot"]), p(b["scl"]), s), "view_forward")
        a = self.lambda_mse * (1.0 - self.lambda_dssim)
        bb = self.lambda_mse * self.lambda_dssim
        chk(lib.cgs_photometric_loss_indexed(H, W, p(b["color"]), p(self._gt_stack), p(self._view_idx), cf(0.1),
                                             p(self._npos_table), cf(a), cf(bb), 1, p(b["photo_ws"]), p(b["g_img"]),
                                             p(b["loss"]), s), "photometric_loss_indexed")
        loss = b["loss"]
        extra = None
        if self.regularisers:   # needs only forward quantities: runs before the backward and hands it dL/drotation_raw
            chk(lib.cgs_curve_regularizers(B, m, p(b["rot"]), p(ol), p(wl), p(b["radii"]), cf(self.opacity_loss_weight),
                                           p(self._opa_gate), cf(self.lambda_curve_smo), cf(self.lambda_width), cf(0.005),
                                           p(b["reg_ws"]), p(b["reg_loss"]), p(b["r_rot"]), p(b["r_op"]), p(b["r_w"]), s),
                "curve_regularizers")
            extra = b["r_rot"]
            loss = loss + b["reg_loss"]
        chk(lib.cgs_view_backward(
            B, m, p(cp), p(wl), p(b["isb"]), p(b["coef"]), cf(1e-8), p(b["norms"]), p(ol), p(mask), cf(self.mask_threshold),
            None, p(b["geom"]), p(b["bin"]), p(b["img"]), p(b["bg"]), W, H, p(cam.world_view_transform),
            p(cam.full_proj_transform), p(cam.camera_center), tanx, tany, p(b["radii"]), p(b["g_img"]), p(extra),
            p(b["g_m2d"]), p(grads.view("curve_points")), p(grads.view("width")), p(grads.view("opacity")),
            p(grads.view("mask")) if mask is not None else None, p(b["view_scratch"]), 0, s), "view_backward")
        if self.regularisers:
            grads.view("opacity").add_(b["r_op"])
            grads.view("width").add_(b["r_w"])
            if self._use_conn:
                chk(lib.cgs_endpoint_connection_loss(B, p(cp), cf(0.05), cf(self.lambda_points_conn), p(b["conn_ws"]),
                                                     p(b["conn_loss"]), p(grads.view("curve_points")), 1, s),
                    "endpoint_connection_loss")
                loss = loss + b["conn_loss"]
        if self._use_mask:      # train.py:110-111: lambda_mask * mean(sigmoid(mask)), gradient added by hand
            sg = torch.sigmoid(mask)
            loss = loss + self.lambda_mask * sg.mean()
            grads.view("mask").add_(sg * (1 - sg), alpha=self.lambda_mask / mask.numel())
        status = b["status"]
        if self._capture_coll:
            import torch.distributed as dist
            dist.all_reduce(g.optimizer.grads.flat)
            dist.all_reduce(status[2:3], op=dist.ReduceOp.MAX)   # any rank overflowed -> every rank skips
        if not self._collective or self._capture_coll:
            g.optimizer.step_dev(zero_grad=True, skip_flag=status[2:3], report=self._report())
        self.last = dict(radii=b["radii"], dL_dmeans2D=b["g_m2d"], render=b["color"],
                         depth=b["invd"] if self.aux_outputs else None, all_map=b["omap"] if self.aux_outputs else None)
        return loss, status

    def _report(self):
        """(execution counter, pinned flag ring) for the captured Adam kernel, or None where the optimizer runs outside the
        replayed sequence (view-parallel mode without captured collectives: the flag is all-reduced first)."""
        if self._collective and not self._capture_coll:
            return None
        return self._report_seq, self._flag_host

    def _probe_capacity(self):
        """Longest tile list over a few eager (exact-path) renders -> bucket capacity."""
        from . import _lib as L
        import ctypes
        lib = L.load()
        longest = 1
        with torch.no_grad():
            for cam in self.cams[:min(8, len(self.cams))]:
                render(cam, self.g, self.pipe, self.bg, compute_visibility=False, clamp=False, compute_rend_dir=False)
                m = ctypes.c_int64()
                lib.cgs_last_forward_stats(None, ctypes.byref(m), None)
                longest = max(longest, int(m.value))
        limit = int(lib.cgs_bucket_capacity_limit())
        cap = (int(longest * self.cap_margin) + 64 + 63) // 64 * 64
        if cap > limit:
            raise RuntimeError(f"GraphedTrainStep: tile lists of {longest} entries need buckets beyond the limit "
                               f"({limit}); use TrainStep for this scene")
        return cap

    INPUT_BYTES = 160

    def _bind_inputs(self):
        """(Re)derive the views of the per-step input block from the current optimizer's device state."""
        opt = self.g.optimizer
        opt.device_state(extra_bytes=self.INPUT_BYTES)
        tail = opt.state_extra()
        self._inputs = tail[:144].view(torch.float32)
        self._cam.pack = self._inputs[0:35]
        self._cam.world_view_transform = self._cam.pack[0:16].view(4, 4)
        self._cam.full_proj_transform = self._cam.pack[16:32].view(4, 4)
        self._cam.camera_center = self._cam.pack[32:35]
        self._npos = self._inputs[35:36].view(torch.int32)
        self._view_idx = tail[144:148].view(torch.int32)
        self._bound_state = opt._state_dev

    def _stage(self, vi):
        """Everything the replay of view `vi` needs, stream-ordered: ONE host-to-device copy (Adam scalars + inputs)."""
        self.g.optimizer.stage_step(extra=self._host_packs[vi])
        if not self.direct:
            self._gt.copy_(self.gts[vi][:1], non_blocking=True)
        gate = 1.0 if self.reset_timestep > 0 else 0.0
        if gate != self._gate_value:
            self._opa_gate.fill_(gate)
            self._gate_value = gate

    def _capture(self, vi):
        if self._cap == 0:
            self._cap = self._probe_capacity()
        opt = self.g.optimizer
        if getattr(opt, "_state_dev", None) is not self._bound_state:   # optimizer (re)built since the last capture
            self._bind_inputs()
        # warm-up replicas of the body must not change the model: snapshot, run on a side stream, restore
        snap = (opt.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count)
        # The gradient accumulators of the parameters live as long as an autograd graph references them, and they keep
        # the stream they were created on.  Eager steps create them on the default stream, and a captured backward
        # that has to synchronise with the default stream cannot be captured: drop the old graph (the derived splat
        # tensors of the last prepare_scaling_rot hold it) so the warm-up below re-creates them on a side stream.
        g = self.g
        g._xyz, g._rotation, g._scaling = g._xyz.detach(), g._rotation.detach(), g._scaling.detach()
        self._loss = self._status = None
        self._stage(vi)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # capture on the warm-up stream: the gradient accumulators created there then share the capture stream, and the
        # captured backward stays single-stream (a cross-stream AccumulateGrad inside the capture lets the allocator
        # recycle blocks the other stream still uses -- later replays then read clobbered intermediates)
        with no_gc(), torch.cuda.graph(graph, stream=side):
            self._loss, self._status = self._body()
        opt.flat.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2])
        opt.step_count = snap[3]
        opt.grads.zero_()
        self.g.prepare_scaling_rot()
        self._report_seq.fill_(self._report_next)   # (the warm-up executions of the body counted too)
        self._graph = graph
        self.recaptures += 1

    def _check_overflow(self, block=False):
        """Steps whose status has arrived: a raised flag means that step was skipped on the device -> redo it."""
        redo = []
        keep = []
        for ev, slot, vi, it in self._inflight:
            if self._collective and not block:
                if it > self.iteration - 1:       # called before iteration += 1: entries of iteration <= k - 2 ...
                    keep.append((ev, slot, vi, it))
                    continue
                ev.synchronize()                  # ... are examined by every rank at the same point (long finished)
            if block:
                ev.synchronize()
            if ev.query():
                if int(self._flag_host[slot]) != 0:
                    redo.append((vi, it))
            else:
                keep.append((ev, slot, vi, it))
        self._inflight = keep
        opt = self.g.optimizer
        for vi, it in redo:
            now = self.iteration
            self.iteration = it - 1                    # redo with the learning rate ...
            opt.step_count = self._t0 + it - 1         # ... and the Adam step number of its own iteration
            self._refresh_derived()
            TrainStep.step(self, view_index=vi)        # exact path, eager
            self.iteration = now
            opt.step_count = self._t0 + now            # every iteration up to `now` is applied, redone or in flight
            self._cap = 0                              # re-probe and re-capture with larger buckets
            self._graph = None
            self._bufs = None
        return len(redo)

    def step(self, view_index=None):
        g = self.g
        use_mask = self.iteration + 1 >= self.densify_until_iter
        use_conn = self._conn_active(self.iteration + 1)
        if use_mask != self._use_mask or use_conn != self._use_conn:   # a phase of train.py starts: re-capture
            self.finish()
            self._use_mask, self._use_conn = use_mask, use_conn
            self._graph = None
        self._check_overflow()
        self.iteration += 1
        self.reset_timestep += 1                   # train.py:76
        g.update_learning_rate(self.iteration)
        vi = self._next_view() if view_index is None else view_index
        if self._graph is None:
            self._capture(vi)
        self._stage(vi)
        self._graph.replay()
        if self._collective and not self._capture_coll:
            import torch.distributed as dist
            dist.all_reduce(g.optimizer.grads.flat)
            dist.all_reduce(self._status[2:3], op=dist.ReduceOp.MAX)   # any rank overflowed -> every rank skips
            g.optimizer.step_dev(zero_grad=True, skip_flag=self._status[2:3])
        self._derived_stale = True   # g._xyz/_rotation/_scaling now hold the values of BEFORE this step's update
        if self._report() is not None:
            slot = self._report_next % 64             # written by the replay's own Adam kernel
            self._report_next += 1
        else:
            slot = self.iteration % 64
            self._flag_host[slot:slot + 1].copy_(self._status[2:3], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._inflight.append((ev, slot, vi, self.iteration))
        if len(self._inflight) > 32:
            self._check_overflow(block=True)
        return self._loss, None

    def _on_topology_change(self):
        TrainStep._on_topology_change(self)
        self._graph = None      # sizes are graph constants: re-probe the bucket capacity and re-capture
        self._cap = 0
        self._bufs = None
        self._loss = self._status = None
        self._derived_stale = False

    def _refresh_derived(self):
        if getattr(self, "_derived_stale", False):
            self.g.prepare_scaling_rot()
            self._derived_stale = False

    def finish(self):
        """Drain: no skipped step left behind and the derived splat tensors match the parameters (call before reading
        the model from outside)."""
        while self._inflight:
            self._check_overflow(block=True)
        self._refresh_derived()
