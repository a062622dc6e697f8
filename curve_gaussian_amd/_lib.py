"""ctypes binding of libcurvegs.so (C ABI declared in include/curvegs.h).

The product path has NO fallback: if the HIP library is missing or no GPU is present, calls raise.
``load()`` only dlopens the library (works on a CPU-only box, used by the "symbols exported" tests).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CGS_LIB", os.path.join(_HERE, "libcurvegs.so"))  # CGS_LIB: A/B experiment builds

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); kept in lock-step with include/curvegs.h (tests/test_abi.py parses the header)
SIGNATURES = {
    "cgs_last_error": (C.c_char_p, []),
    "cgs_version": (_i, []),
    "cgs_target_arch": (C.c_char_p, []),
    "cgs_geometry_bytes": (C.c_size_t, [_i]),
    "cgs_image_bytes": (C.c_size_t, [_i, _i]),
    "cgs_binning_bytes": (C.c_size_t, [_i64]),
    "cgs_rasterize_forward_static": (_i, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _i, _i, _i, _vp, _i, _i,
                                          _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f,
                                          _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "cgs_view_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_size_t, _vp, C.c_uint32,
                              _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_forward_checked": (_i64, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_size_t, _vp,
                                        C.c_uint32, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_forward_begin": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_size_t, _vp,
                                    C.c_uint32, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_forward_shared": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_size_t, _vp,
                                     C.c_uint32, _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_forward_render": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, C.c_size_t, _vp, C.c_uint32,
                                     _vp, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_backward_render": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp,
                                      _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cgs_view_forward_wait": (_i64, [_i, C.POINTER(_i64)]),
    "cgs_view_forward_abandon": (None, [_i]),
    "cgs_render_epilogue": (_i, [_i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "cgs_clamp_backward": (_i, [_i64, _vp, _vp, _vp, _vp]),
    "cgs_bucket_capacity_hint": (C.c_uint32, [_i, _i, _i]),
    "cgs_last_forward_visible": (_i64, []),
    "cgs_visible_indices": (_i, [_i, _vp, _vp, _i, _i, _vp, _vp]),
    "cgs_view_shared_begin": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_view_shared_end": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "cgs_view_backward_scratch_floats": (C.c_size_t, [_i, _i]),
    "cgs_view_norms_backward_range": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "cgs_view_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp,
                               _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "cgs_image_status_offset": (C.c_size_t, [_i, _i]),
    "cgs_status_words": (_i, []),
    "cgs_bucket_capacity_limit": (C.c_uint32, []),
    "cgs_adam_step_flat_dev": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _i, _vp, _vp]),
    "cgs_adam_step_flat_dev_report": (_i, [_i64, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _i, _vp, _vp, _vp, _i, _vp]),
    "cgs_adam_state_bytes": (C.c_size_t, []),
    "cgs_reset_binning_hints": (None, []),
    "cgs_last_forward_stats": (None, [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i)]),
    "cgs_prof_enable": (None, [_i]),
    "cgs_prof_reset": (None, []),
    "cgs_prof_collect": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64), _i]),
    "cgs_rasterize_forward": (_i64, [ALLOC_FN, _vp, ALLOC_FN, _vp, ALLOC_FN, _vp, _i, _i, _i, _vp, _i, _i,
                                     _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i,
                                     _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "cgs_rasterize_backward": (_i, [_i, _i, _i, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp,
                                    _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "cgs_mark_visible": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "cgs_ssim_forward": (_i, [_i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_ssim_backward": (_i, [_i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_edge_aware_loss": (_i, [_i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "cgs_photometric_workspace_bytes": (C.c_size_t, [_i, _i]),
    "cgs_edge_count": (_i, [_i, _i, _i, _vp, _f, _vp, _vp]),
    "cgs_photometric_loss": (_i, [_i, _i, _vp, _vp, _f, _vp, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "cgs_endpoint_connection_workspace_bytes": (C.c_size_t, [_i]),
    "cgs_endpoint_connection_loss": (_i, [_i, _vp, _f, _f, _vp, _vp, _vp, _i, _vp]),
    "cgs_photometric_loss_indexed": (_i, [_i, _i, _vp, _vp, _vp, _f, _vp, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "cgs_curve_regularizers_workspace_bytes": (C.c_size_t, []),
    "cgs_curve_regularizers": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_adam_step_flat": (_i, [_i64, _vp, _vp, _vp, _vp, C.c_char_p, _i, _f, _f, _f, _i, _i, _vp]),
    "cgs_knn_workspace_bytes": (C.c_size_t, [_i]),
    "cgs_knn_mean_dist2": (_i, [_i, _vp, _vp, _vp, _vp]),
    "cgs_sample_curves_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "cgs_sample_curves_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_splat_attrs_forward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cgs_splat_attrs_backward": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _vp]),
}

_lib = None


class CurveGSError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen libcurvegs.so and attach signatures.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CurveGSError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C curve_gaussian_amd/csrc`). There is no CPU fallback.")
    import torch  # noqa: F401  -- make sure torch's bundled libamdhip64.so.7 is the HIP runtime in this process
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_shim = None
SHIM_PATH = os.path.join(_HERE, "_cgs_torch.so")
# _cgs_torch.so is linked against $ORIGIN/libcurvegs.so: with CGS_LIB pointing at an experiment build it would run the
# DEFAULT kernels (and hand out handles of another library instance) while ctypes drives the experiment -- an A/B run must
# use one library, so an overridden CGS_LIB selects the ctypes bindings.  Resolved once: use_shim() sits on the
# per-iteration path of the eager routes (two realpath() calls there cost 70 us per training iteration).
_LIB_IS_DEFAULT = os.path.realpath(LIB_PATH) == os.path.realpath(os.path.join(_HERE, "libcurvegs.so"))


def use_shim() -> bool:
    """The compiled torch <-> C-ABI host shim (csrc/torch_shim.cpp) is the default binding of the operator API and of the fused
    view route; CGS_TORCH_SHIM=0 selects the ctypes bindings instead (A/B measurements of the host floor).  Same library, same
    kernels either way."""
    if os.environ.get("CGS_TORCH_SHIM", "1") == "0":
        return False
    return _LIB_IS_DEFAULT


def shim():
    """import curve_gaussian_amd._cgs_torch (built by `make -C curve_gaussian_amd/csrc`).  Raises if it has not been built."""
    global _shim
    if _shim is not None:
        return _shim
    load()   # libcurvegs.so first (use_shim() is False when CGS_LIB names another build: the shim binds $ORIGIN/libcurvegs.so)
    if not os.path.exists(SHIM_PATH):
        raise CurveGSError(
            f"{SHIM_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C curve_gaussian_amd/csrc`); CGS_TORCH_SHIM=0 selects the ctypes bindings.")
    import importlib
    _shim = importlib.import_module("curve_gaussian_amd._cgs_torch")
    return _shim


def last_error() -> str:
    return load().cgs_last_error().decode("utf-8", "replace")


def check(rc, what: str):
    if rc < 0:
        raise CurveGSError(f"{what} failed (status {rc}): {last_error()}")
    return rc


def require_gpu_tensor(t, name: str):
    if not t.is_cuda:
        raise CurveGSError(f"{name} must be a GPU tensor (got device {t.device}); libcurvegs has no CPU path")


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_GUARD = _NoGuard()


def device_guard(dev):
    """``torch.cuda.device(dev)`` only when `dev` is not already the current device (the context manager costs ~8 us of
    host time per op; the reference's shim has no device guard at all)."""
    import torch
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)


def raw_stream(dev):
    """hipStream_t of torch's current stream on `dev` as an int (one C call; torch.cuda.current_stream() builds a
    Stream object and is ~10x slower)."""
    import torch
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the reference's 'empty tensor' convention)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def prof_collect():
    lib = load()
    cap = 64
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    n_l = (_i64 * cap)()
    n = lib.cgs_prof_collect(names, ms, n_l, cap)
    return {names[i].decode(): (ms[i], int(n_l[i])) for i in range(min(n, cap))}
