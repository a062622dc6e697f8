"""One-launch Adam over flat parameter / gradient buffers (csrc/loss.hip::k_adam_flat).

The reference builds ``torch.optim.Adam`` over six parameter groups (scene/gaussian_curve_model.py:200-213) and steps
them with ~8 foreach kernels per group (train.py:235).  ``FlatAdam`` keeps every learnable tensor as a view of ONE flat
buffer (the same layout as the flat gradient buffer of view_parallel.FlatGrads) and updates all of them, with per-group
learning rates, in a single kernel.  Semantics = torch.optim.Adam(lr per group, betas=(0.9,0.999), eps) without weight
decay / amsgrad."""
import ctypes as C
import struct

import torch

from .. import _lib as L


class FlatAdam:
    STATE_RING = 64

    def __init__(self, named_params, lrs, flat_grads, betas=(0.9, 0.999), eps=1e-15):
        """named_params: dict name -> nn.Parameter (insertion order = flat layout, must match flat_grads.names);
        lrs: dict name -> lr; flat_grads: view_parallel.FlatGrads built over the same dict."""
        self.names = list(named_params)
        assert self.names == flat_grads.names
        self.params = named_params
        self.grads = flat_grads
        any_p = next(iter(named_params.values()))
        L.require_gpu_tensor(any_p, "parameters")
        self.device = any_p.device
        n = flat_grads.flat.numel()
        self.flat = torch.empty(n, dtype=torch.float32, device=self.device)
        for name, p in named_params.items():      # adopt: parameters become views of the flat buffer
            a, b = flat_grads.slices[name]
            self.flat[a:b].copy_(p.data.reshape(-1))
            p.data = self.flat[a:b].view_as(p)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.param_groups = [{"name": nm, "lr": float(lrs[nm]), "params": [named_params[nm]]} for nm in self.names]

    def _segments(self):
        """Host-side {begin, lr} table, rebuilt every step: learning rates may have been changed through param_groups
        (update_learning_rate).  It is passed to the kernel by value -- no host-to-device copy, no stream sync."""
        return b"".join(struct.pack("<qff", self.grads.slices[g["name"]][0], float(g["lr"]), 0.0) for g in self.param_groups)

    def step(self, zero_grad=False):
        """One Adam step over all groups; zero_grad=True also clears the flat gradient buffer in the same kernel."""
        self.step_count += 1
        lib = L.load()
        with L.device_guard(self.device):
            rc = lib.cgs_adam_step_flat(self.flat.numel(), L.ptr(self.flat), L.ptr(self.grads.flat), L.ptr(self.exp_avg),
                                        L.ptr(self.exp_avg_sq), self._segments(), len(self.param_groups),
                                        C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                        self.step_count, 1 if zero_grad else 0,
                                        L.raw_stream(self.device))
        L.check(rc, "cgs_adam_step_flat")

    # ---- topology edits (scene/topology.py): per-curve tensors change their first dimension
    def state_of(self, name):
        """(exp_avg, exp_avg_sq) views shaped like the parameter, or None before the first step (torch.optim.Adam
        creates its state lazily; the reference's surgery code branches on that)."""
        if self.step_count == 0:
            return None
        a, b = self.grads.slices[name]
        p = self.params[name]
        return self.exp_avg[a:b].view_as(p), self.exp_avg_sq[a:b].view_as(p)

    def rebuild(self, new_params, new_state):
        """Replace every group's values (dict name -> tensor, new leading dimension allowed) and Adam moments (dict name
        -> (exp_avg, exp_avg_sq) or None = zeros): new flat parameter / gradient / moment buffers, new nn.Parameters
        that are views of them.  Returns name -> nn.Parameter; step count and learning rates are kept."""
        from ..view_parallel import FlatGrads
        import torch.nn as nn
        shapes = {n: tuple(new_params[n].shape) for n in self.names}
        total = sum(int(torch.tensor(s).prod()) if len(s) else 1 for s in shapes.values())
        flat = torch.empty(total, dtype=torch.float32, device=self.device)
        m = torch.zeros(total, dtype=torch.float32, device=self.device)
        v = torch.zeros(total, dtype=torch.float32, device=self.device)
        out, o = {}, 0
        for n in self.names:
            t = new_params[n].detach().float()
            k = t.numel()
            flat[o:o + k].copy_(t.reshape(-1))
            st = new_state.get(n) if new_state else None
            if st is not None:
                m[o:o + k].copy_(st[0].reshape(-1))
                v[o:o + k].copy_(st[1].reshape(-1))
            out[n] = nn.Parameter(flat[o:o + k].view(shapes[n]))
            o += k
        self.flat, self.exp_avg, self.exp_avg_sq = flat, m, v
        self.params = out
        self.grads = FlatGrads(out)
        for g in self.param_groups:
            g["params"] = [out[g["name"]]]
        return out

    # ---- graph-replayable variant: per-step scalars in device memory, optional device-side skip flag
    def device_state(self, extra_bytes=None):
        """(device uint8 tensor, pinned host mirror) holding {segments, 1 - b1^t, sqrt(1 - b2^t)} for
        cgs_adam_step_flat_dev, followed by `extra_bytes` of caller data that ride in the same per-step copy
        (``stage_step(extra=...)``; ``state_extra()`` is the device view of that tail)."""
        n = int(L.load().cgs_adam_state_bytes())
        have = getattr(self, "_state_dev", None)
        if have is None or (extra_bytes is not None and have.numel() != n + int(extra_bytes)):
            total = n + int(extra_bytes or 0)
            self._state_dev = torch.zeros(total, dtype=torch.uint8, device=self.device)
            # ring of pinned staging slots: the host-to-device copy is asynchronous, so the slot of step t must not be
            # rewritten before that copy has run (callers keep fewer than STATE_RING steps in flight)
            self._state_host = torch.zeros(self.STATE_RING, total, dtype=torch.uint8).pin_memory()
        return self._state_dev, self._state_host

    def state_extra(self):
        dev, _ = self.device_state()
        return dev[int(L.load().cgs_adam_state_bytes()):]

    def stage_step(self, extra=None):
        """Advance the step count and enqueue (stream-ordered, non-blocking) the scalars of that step -- ONE copy from a
        pinned slot; `extra` (bytes, at most the size reserved by device_state) lands behind the Adam scalars."""
        dev, host = self.device_state()
        self.step_count += 1
        b1, b2 = self.betas
        segs = self._segments()
        blob = segs + b"\0" * (16 * 16 - len(segs)) + struct.pack("<ffff", 1.0 - b1 ** self.step_count,
                                                                  (1.0 - b2 ** self.step_count) ** 0.5, 0.0, 0.0)
        if extra is not None:
            blob = blob + extra
        if len(blob) > dev.numel():
            raise ValueError("FlatAdam.stage_step: extra bytes exceed the reserved tail")
        slot = host[self.step_count % self.STATE_RING]
        slot[:len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        dev.copy_(slot, non_blocking=True)

    def step_dev(self, zero_grad=True, skip_flag=None, report=None):
        """Adam step whose scalars come from ``device_state()`` (call ``stage_step()`` first, outside any graph
        capture); skip_flag: optional device int32/uint32 scalar -- non-zero leaves parameters and moments untouched.
        report = (seq, ring): device int32 scalar counting the executions and an int32 ring (device or PINNED HOST memory)
        whose entry ``n % len(ring)`` receives 1 if execution n was skipped, else 0 (cgs_adam_step_flat_dev_report)."""
        dev, _ = self.device_state()
        lib = L.load()
        if report is not None:
            seq, ring = report
            rc = lib.cgs_adam_step_flat_dev_report(
                self.flat.numel(), L.ptr(self.flat), L.ptr(self.grads.flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), L.ptr(dev),
                len(self.param_groups), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                1 if zero_grad else 0, L.ptr(skip_flag) if skip_flag is not None else None, L.ptr(seq), L.ptr(ring),
                int(ring.numel()), L.raw_stream(self.device))
            L.check(rc, "cgs_adam_step_flat_dev_report")
            return
        rc = lib.cgs_adam_step_flat_dev(self.flat.numel(), L.ptr(self.flat), L.ptr(self.grads.flat), L.ptr(self.exp_avg),
                                        L.ptr(self.exp_avg_sq), L.ptr(dev), len(self.param_groups),
                                        C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                        1 if zero_grad else 0, L.ptr(skip_flag) if skip_flag is not None else None,
                                        L.raw_stream(self.device))
        L.check(rc, "cgs_adam_step_flat_dev")

    def zero_grad(self, set_to_none=False):
        self.grads.zero_()
