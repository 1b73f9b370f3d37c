"""ctypes binding of libobia_hip.so (C ABI: include/obia_hip.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is present the operators raise.
"""
import ctypes
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# OBIA_HIP_LIB: developer override used by kernel experiments (tools/build_variant.sh); still a HIP build of this library
LIB_PATH = os.environ.get("OBIA_HIP_LIB") or os.path.join(_CSRC, "libobia_hip.so")

OBIA_OK = 0
E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED, E_EMPTY, E_NONFINITE = -1, -2, -3, -4, -5, -6


class SlicParams(ctypes.Structure):
    """obia_slic_params (include/obia_hip.h)."""
    _fields_ = [("compactness", ctypes.c_double), ("min_size_factor", ctypes.c_double),
                ("max_size_factor", ctypes.c_double), ("n_segments", ctypes.c_int32),
                ("max_num_iter", ctypes.c_int32), ("convert2lab", ctypes.c_int32),
                ("enforce_connectivity", ctypes.c_int32), ("slic_zero", ctypes.c_int32),
                ("start_label", ctypes.c_int32), ("normalize_bands", ctypes.c_int32),
                ("exit_on_fixed_point", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("sigma_zyx", ctypes.c_double * 3), ("spacing_zyx", ctypes.c_double * 3)]


class SlicSeeds(ctypes.Structure):
    """obia_slic_seeds (include/obia_hip.h): caller-supplied initial centroids."""
    _fields_ = [("yx", ctypes.c_void_p), ("steps_zyx", ctypes.c_double * 3), ("n", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class TilingParams(ctypes.Structure):
    """obia_tiling_params (include/obia_hip.h)."""
    _fields_ = [("crown_radius", ctypes.c_double), ("pixel_width", ctypes.c_double),
                ("pixel_height", ctypes.c_double), ("tile_size", ctypes.c_int32), ("buffer", ctypes.c_int32),
                ("white_order", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_SIGNATURES = {
    "obia_abi_version": (ctypes.c_int, []),
    "obia_last_error": (ctypes.c_char_p, []),
    "obia_create": (_P, [_I]),
    "obia_create_on_stream": (_P, [_I, _P]),
    "obia_destroy": (None, [_P]),
    "obia_synchronize": (_I, [_P]),
    "obia_workspace_bytes": (ctypes.c_int64, [_P]),
    "obia_slic_default_params": (None, [ctypes.POINTER(SlicParams)]),
    "obia_slic_f32": (_I, [_P, _P, _I, _I, _I, _P, ctypes.POINTER(SlicParams), _P, ctypes.POINTER(_I)]),
    "obia_slic_f32_dev": (_I, [_P, _P, _I, _I, _I, _P, ctypes.POINTER(SlicParams), _P, ctypes.POINTER(_I)]),
    "obia_slic_assign_only_f32_dev": (_I, [_P, _P, _I, _I, _I, _P, ctypes.POINTER(SlicParams), _P, ctypes.POINTER(_I)]),
    "obia_slic_seeded_f32_dev": (_I, [_P, _P, _I, _I, _I, _P, ctypes.POINTER(SlicParams), ctypes.POINTER(SlicSeeds), _I, _P,
                                      ctypes.POINTER(_I)]),
    "obia_enforce_connectivity_i32_dev": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, ctypes.POINTER(_I)]),
    "obia_zonal_stats_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "obia_zonal_stats_f32_dev": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "obia_zonal_moments_f32_dev": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P]),
    "obia_zonal_moments_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "obia_texture_stats_f32_dev": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P]),
    "obia_label_edges_u8_dev": (_I, [_P, _P, _I, _I, _P, _P]),
    "obia_sample_labels_i32_dev": (_I, [_P, _P, _I, _I, _P, _P, ctypes.c_int64, _I, _P]),
    "obia_polygon_count_i32_dev": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "obia_polygon_rings_i32_dev": (_I, [_P, _P, _I, _I, _I, ctypes.c_int64, ctypes.c_int64, _P, _P, _P, _P, _P, _P]),
    "obia_quickshift_f32": (_I, [_P, _P, _I, _I, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _I, _P, ctypes.POINTER(_I)]),
    "obia_quickshift_f32_dev": (_I, [_P, _P, _I, _I, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P, _I, _P, ctypes.POINTER(_I)]),
    "obia_tiled_slic_f32_dev": (_I, [_P, _P, _P, _I, _I, _I, ctypes.POINTER(TilingParams), ctypes.POINTER(SlicParams), _P,
                                     ctypes.POINTER(ctypes.c_int64)]),
    "obia_tiled_slic_f32": (_I, [_P, _P, _P, _I, _I, _I, ctypes.POINTER(TilingParams), ctypes.POINTER(SlicParams), _P,
                                 ctypes.POINTER(ctypes.c_int64)]),
    "obia_tiler_create": (_P, [_P, _P, _P, _I, _I, _I, _I, _I, ctypes.POINTER(TilingParams), ctypes.POINTER(SlicParams), _P, _I]),
    "obia_tiler_destroy": (None, [_P]),
    "obia_tiler_run": (_I, [_P, _I, _I, _I, _I]),
    "obia_tiler_next_id": (_I, [_P]),
    "obia_tiler_set_segments": (_I, [_P, _I, _I, _P]),
    "obia_tiler_get_alive": (_I, [_P, _P, _I]),
    "obia_tiler_set_alive": (_I, [_P, _P, _I]),
    "obia_tiler_finalize": (_I, [_P, ctypes.POINTER(ctypes.c_int64)]),
    "obia_tiler_import_seam": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int)]),
    "obia_set_profiling": (_I, [_P, _I]),
    "obia_last_timing": (ctypes.c_double, [_P, _I]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def build(force=False):
    """Compile libobia_hip.so for gfx950 with hipcc (obia_amd/csrc/Makefile)."""
    args = ["make", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    """Load the HIP library; raise ImportError loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc, gfx950). obia_amd has no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                # an OLDER build of this library shipped for an A/B timing (OBIA_HIP_LIB, tools/ab.sh) may lack the newest entry
                # points; the regular library must export every one of them (tests/test_abi_and_host.py checks the list)
                if os.environ.get("OBIA_HIP_LIB"):
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().obia_last_error().decode("utf-8", "replace")


def check(rc):
    """Map ABI status codes to the exceptions the reference's callers see."""
    if rc == OBIA_OK:
        return
    msg = last_error()
    if rc in (E_INVALID, E_EMPTY, E_NONFINITE):
        raise ValueError(msg)          # the tiler catches ValueError per tile (tiling.py:149-150)
    if rc == E_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"obia_hip error {rc}: {msg}")


class Context:
    """One obia_ctx: a (device, stream) pair plus its reusable device workspace."""

    def __init__(self, device=0, stream=None):
        lib = load()
        if stream is None:
            self._h = lib.obia_create(int(device))
        else:
            self._h = lib.obia_create_on_stream(int(device), ctypes.c_void_p(int(stream)))
        if not self._h:
            raise RuntimeError(f"obia_create failed: {last_error()} (obia_amd needs an AMD GPU; there is no CPU path)")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            load().obia_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, on):
        """False / 0: off; True / 1: event pairs around every kernel class; 2: around the SLIC colour sweeps only."""
        check(load().obia_set_profiling(self._h, int(on)))

    def timing(self):
        lib = load()
        names = ["assign_ms", "sweeps", "features_ms", "connectivity_ms", "zonal_ms", "total_ms", "prepass_ms", "assign_px",
                 "prepass_px", "assign_store_px", "assign_busy_ms", "prepass_busy_ms", "batch_repeats"]
        return {n: lib.obia_last_timing(self._h, i) for i, n in enumerate(names)}

    def workspace_bytes(self):
        return int(load().obia_workspace_bytes(self._h))


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


def np_ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def mask_bytes(mask, device=None):
    """Device mask as one byte per pixel, any non-zero byte = valid (what every kernel tests).  uint8 and bool tensors
    are passed through without a copy; other dtypes are compared with zero once."""
    import torch
    m = torch.as_tensor(mask, device=device)
    if m.dtype == torch.bool:
        m = m.contiguous().view(torch.uint8)
    elif m.dtype != torch.uint8:
        m = (m != 0).view(torch.uint8)
    return m.contiguous()
