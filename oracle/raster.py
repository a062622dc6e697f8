"""numpy front-end of oracle/raster_ref.c (the CPU restatement of the reference rasterizer).

Mirrors the reference's ``_C.rasterize_gaussians`` / ``_C.rasterize_gaussians_backward`` argument meaning
(/root/reference/submodules/diff-cur-rasterization/rasterize_points.cu:35-130, :132-239) on numpy arrays.
"""
import ctypes as C

import numpy as np

from . import lib as _lib

_f32p = C.POINTER(C.c_float)


def _fp(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def _arr(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if a.size == 0:
        return None
    return a


_sigs_done = False


def _sigs():
    global _sigs_done
    L = _lib()
    if _sigs_done:
        return L
    L.ora_forward.restype = C.c_void_p
    L.ora_forward.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p,
                              C.c_float, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_int,
                              C.c_int, C.c_int, _f32p, _f32p, _f32p, C.POINTER(C.c_int)]
    L.ora_backward.restype = None
    L.ora_backward.argtypes = [C.c_void_p] + [_f32p] * 7 + [C.c_float] + [_f32p] * 5 + [C.c_float, C.c_float] + \
                              [_f32p] * 3 + [C.c_int, C.c_int] + [_f32p] * 11
    L.ora_free.argtypes = [C.c_void_p]
    L.ora_free.restype = None
    L.ora_num_rendered.argtypes = [C.c_void_p]
    L.ora_num_rendered.restype = C.c_int64
    for name, rt in [("ora_means2D", _f32p), ("ora_depths", _f32p), ("ora_cov3D", _f32p), ("ora_conic_opacity", _f32p),
                     ("ora_final_T", _f32p), ("ora_rgb", _f32p), ("ora_tiles_touched", C.POINTER(C.c_uint32)),
                     ("ora_point_list", C.POINTER(C.c_uint32)), ("ora_keys", C.POINTER(C.c_uint64)),
                     ("ora_ranges", C.POINTER(C.c_uint32)), ("ora_n_contrib", C.POINTER(C.c_uint32))]:
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = rt
    L.ora_mark_visible.argtypes = [C.c_int, _f32p, _f32p, _f32p, C.POINTER(C.c_uint8)]
    L.ora_set_num_threads.argtypes = [C.c_int]
    L.ora_num_threads.restype = C.c_int
    L.ora_get_higher_msb.argtypes = [C.c_uint32]
    L.ora_get_higher_msb.restype = C.c_uint32
    _sigs_done = True
    return L


class ForwardResult:
    """Outputs + saved state of one oracle forward.  Call ``.free()`` (or rely on __del__)."""

    def __init__(self, ctx, P, H, W, color, invdepth, all_map, radii, inputs):
        self._ctx = ctx
        self.P, self.H, self.W = P, H, W
        self.color, self.invdepth, self.out_all_map, self.radii = color, invdepth, all_map, radii
        self.inputs = inputs
        L = _sigs()
        self.num_rendered = int(L.ora_num_rendered(ctx))

    def _view(self, fn, n, dtype):
        L = _sigs()
        if n == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(getattr(L, fn)(self._ctx), shape=(n,)).astype(dtype, copy=True)

    @property
    def means2D(self): return self._view("ora_means2D", 2 * self.P, np.float32).reshape(-1, 2)
    @property
    def depths(self): return self._view("ora_depths", self.P, np.float32)
    @property
    def cov3D(self): return self._view("ora_cov3D", 6 * self.P, np.float32).reshape(-1, 6)
    @property
    def conic_opacity(self): return self._view("ora_conic_opacity", 4 * self.P, np.float32).reshape(-1, 4)
    @property
    def tiles_touched(self): return self._view("ora_tiles_touched", self.P, np.uint32)
    @property
    def point_list(self): return self._view("ora_point_list", self.num_rendered, np.uint32)
    @property
    def keys(self): return self._view("ora_keys", self.num_rendered, np.uint64)
    @property
    def final_T(self): return self._view("ora_final_T", self.H * self.W, np.float32).reshape(self.H, self.W)
    @property
    def n_contrib(self): return self._view("ora_n_contrib", self.H * self.W, np.uint32).reshape(self.H, self.W)
    @property
    def ranges(self):
        tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        return self._view("ora_ranges", 2 * tiles, np.uint32).reshape(-1, 2)

    def free(self):
        if self._ctx:
            _sigs().ora_free(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def forward(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp, all_map,
            viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered=False,
            antialiasing=False, render_geo=True) -> ForwardResult:
    L = _sigs()
    means3D = _arr(means3D)
    P = 0 if means3D is None else means3D.shape[0]
    sh_a = _arr(sh)
    M = 0 if sh_a is None else sh_a.shape[1]
    a = dict(bg=_arr(bg), means3D=means3D, sh=sh_a, colors=_arr(colors_precomp), opac=_arr(opacities),
             scales=_arr(scales), rots=_arr(rotations), cov3D=_arr(cov3D_precomp), all_map=_arr(all_map),
             view=_arr(viewmatrix), proj=_arr(projmatrix), campos=_arr(campos), scale_modifier=float(scale_modifier),
             tan_fovx=float(tan_fovx), tan_fovy=float(tan_fovy), degree=int(degree), M=M,
             antialiasing=bool(antialiasing), render_geo=bool(render_geo))
    color = np.zeros((1, H, W), np.float32)
    invd = np.zeros((1, H, W), np.float32)
    amap = np.zeros((4, H, W), np.float32)
    radii = np.zeros(max(P, 1), np.int32)
    ctx = L.ora_forward(P, int(degree), M, _fp(a["bg"]), W, H, _fp(means3D), _fp(sh_a), _fp(a["colors"]), _fp(a["opac"]),
                        _fp(a["scales"]), a["scale_modifier"], _fp(a["rots"]), _fp(a["cov3D"]), _fp(a["all_map"]),
                        _fp(a["view"]), _fp(a["proj"]), _fp(a["campos"]), a["tan_fovx"], a["tan_fovy"],
                        int(prefiltered), int(antialiasing), int(render_geo), _fp(color), _fp(invd), _fp(amap),
                        radii.ctypes.data_as(C.POINTER(C.c_int)))
    if not ctx:
        raise MemoryError("oracle forward allocation failed")
    return ForwardResult(ctx, P, H, W, color, invd, amap, radii[:P], a)


def backward(fw: ForwardResult, dL_dcolor, dL_dinvdepth, dL_dall_map):
    """Returns dict with the 9 reference gradients (+ dL_dconic, dL_dinvdepths scratch)."""
    L = _sigs()
    a = fw.inputs
    P, M = fw.P, a["M"]
    n = max(P, 1)
    g = dict(dL_dmeans2D=np.zeros((n, 3), np.float32), dL_dcolors=np.zeros((n, 1), np.float32),
             dL_dopacity=np.zeros((n, 1), np.float32), dL_dmeans3D=np.zeros((n, 3), np.float32),
             dL_dcov3D=np.zeros((n, 6), np.float32), dL_dsh=np.zeros((n, max(M, 1)), np.float32),
             dL_dscales=np.zeros((n, 3), np.float32), dL_drotations=np.zeros((n, 4), np.float32),
             dL_dall_map=np.zeros((n, 4), np.float32), dL_dconic=np.zeros((n, 4), np.float32))
    dcol = _arr(dL_dcolor)
    if dcol is None:
        dcol = np.zeros((1, fw.H, fw.W), np.float32)
    dinv = _arr(dL_dinvdepth)
    damap = _arr(dL_dall_map)
    if damap is None:
        damap = np.zeros((4, fw.H, fw.W), np.float32)
    dinvd_splat = np.zeros(n, np.float32) if dinv is not None else None
    L.ora_backward(fw._ctx, _fp(a["bg"]), _fp(a["means3D"]), _fp(a["sh"]), _fp(a["colors"]), _fp(a["all_map"]),
                   _fp(a["opac"]), _fp(a["scales"]), a["scale_modifier"], _fp(a["rots"]), _fp(a["cov3D"]),
                   _fp(a["view"]), _fp(a["proj"]), _fp(a["campos"]), a["tan_fovx"], a["tan_fovy"], _fp(dcol), _fp(dinv),
                   _fp(damap), int(a["antialiasing"]), int(a["render_geo"]), _fp(g["dL_dmeans2D"]), _fp(g["dL_dcolors"]),
                   _fp(g["dL_dopacity"]), _fp(g["dL_dmeans3D"]), _fp(g["dL_dcov3D"]), _fp(g["dL_dsh"]),
                   _fp(g["dL_dscales"]), _fp(g["dL_drotations"]), _fp(g["dL_dall_map"]), _fp(g["dL_dconic"]),
                   _fp(dinvd_splat))
    out = {k: v[:P] for k, v in g.items()}
    out["dL_dsh"] = out["dL_dsh"][:, :M]
    out["dL_dinvdepths"] = None if dinvd_splat is None else dinvd_splat[:P]
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    L = _sigs()
    m = _arr(means3D)
    P = m.shape[0]
    out = np.zeros(P, np.uint8)
    L.ora_mark_visible(P, _fp(m), _fp(_arr(viewmatrix)), _fp(_arr(projmatrix)), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)


def set_num_threads(n: int):
    _sigs().ora_set_num_threads(int(n))


def num_threads() -> int:
    return int(_sigs().ora_num_threads())
