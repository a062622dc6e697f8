"""Per-iteration training sequence of the reference (train.py:75-148,226-243) on the HIP hot path:

    update lr -> pick view -> render (fused attrs + rasterizer) -> edge_aware_loss + fused_ssim -> backward
    -> [view-parallel all-reduce] -> Adam step -> zero grads -> prepare_scaling_rot

Only the photometric terms and (from ``densify_until_iter``) the mask regulariser are included; the O(B^2) connection
loss and the topology edits are out of scope."""
import random

import torch

from .fused_ssim import fused_ssim
from .gaussian_renderer import PipelineParams, render
from .ops.losses import edge_aware_loss, photometric_loss
from .ops.optim import FlatAdam
from .view_parallel import FlatGrads


class TrainStep:
    def __init__(self, gaussians, cameras, gt_images, lambda_mse=10.0, lambda_dssim=0.1, lambda_mask=0.0005,
                 densify_until_iter=7000, mask_threshold=0.01, seed=0, rank=0, world=1, fused=True):
        self.g = gaussians
        self.cams = cameras
        self.gts = gt_images                      # list of [1,H,W] edge maps on the device
        self.lambda_mse, self.lambda_dssim, self.lambda_mask = lambda_mse, lambda_dssim, lambda_mask
        self.densify_until_iter, self.mask_threshold = densify_until_iter, mask_threshold
        self.pipe = PipelineParams()
        self.bg = torch.zeros(3, device=gaussians.device)
        self.rng = random.Random(seed + rank)     # rank-dependent view choice
        self.stack = []
        self.rank, self.world = rank, world
        if gaussians.optimizer is None:
            gaussians.training_setup()
        named = {"curve_points": gaussians._curve_points, "width": gaussians._width, "opacity": gaussians._opacity,
                 "mask": gaussians._mask, "f_dc": gaussians._features_dc, "f_rest": gaussians._features_rest}
        self.flat = FlatGrads(named)
        self.fused = fused
        if fused:   # one-launch Adam with the reference's per-group learning rates (training_setup :203-213)
            lrs = {g["name"]: g["lr"] for g in gaussians.optimizer.param_groups}
            gaussians.optimizer = FlatAdam(named, lrs, self.flat, eps=1e-15)
            gaussians.prepare_scaling_rot()   # parameters moved into the flat buffer: rebuild the derived tensors
        self.iteration = 0

    def step(self):
        g = self.g
        self.iteration += 1
        it = self.iteration
        g.update_learning_rate(it)
        if not self.stack:
            self.stack = list(range(len(self.cams)))
        vi = self.stack.pop(self.rng.randint(0, len(self.stack) - 1))   # train.py:85-90
        cam, gt = self.cams[vi], self.gts[vi]
        use_mask = it >= self.densify_until_iter
        pkg = render(cam, g, self.pipe, self.bg, use_mask=use_mask, mask_thr=self.mask_threshold,
                     compute_visibility=not self.fused, clamp=not self.fused, compute_rend_dir=not self.fused)
        image = pkg["render"]
        if self.fused:   # raw composite in, render()'s clamp applied inside the loss kernels
            loss = photometric_loss(image, gt[:1], self.lambda_mse, self.lambda_dssim, clamp=True)
        else:
            Ll1 = edge_aware_loss(image, gt[:1])
            ssim_value = fused_ssim(image.unsqueeze(0), gt[:1].unsqueeze(0))
            loss = self.lambda_mse * ((1.0 - self.lambda_dssim) * Ll1 + self.lambda_dssim * (1.0 - ssim_value))
        if use_mask:
            loss = loss + self.lambda_mask * torch.mean(torch.sigmoid(g._mask))
        loss.backward()
        self.flat.all_reduce()
        if self.fused:
            g.optimizer.step(zero_grad=True)       # Adam + zero_grad in one launch
        else:
            g.optimizer.step()
            self.flat.zero_()                      # grads are views of the flat buffer: keep them, zero in place
        g.prepare_scaling_rot()                    # train.py:242-243
        return loss.detach(), pkg
