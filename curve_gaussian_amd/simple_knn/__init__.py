"""Drop-in for the reference's ``simple_knn`` package: ``simple_knn._C.distCUDA2`` (submodules/simple-knn/ext.cpp:15-17)."""
import torch

from .. import _lib as L


class _Ext:
    @staticmethod
    def distCUDA2(points):
        L.require_gpu_tensor(points, "points")
        lib = L.load()
        dev = points.device
        with L.device_guard(dev):
            pts = points.float().contiguous()
            P = pts.size(0)
            means = torch.full((P,), 0.0, dtype=torch.float32, device=dev)
            if P > 0:
                ws = torch.empty((int(lib.cgs_knn_workspace_bytes(P)),), dtype=torch.uint8, device=dev)
                rc = lib.cgs_knn_mean_dist2(P, L.ptr(pts), L.ptr(means), L.ptr(ws),
                                            L.raw_stream(dev))
                L.check(rc, "cgs_knn_mean_dist2")
        return means


_C = _Ext()
distCUDA2 = _C.distCUDA2
