"""ctypes front-end of ORACLE O2 (``gsr_oracle.c``): fp32 tiled CPU restatement.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED (no reference tests /
golden vectors exist for this path; anchored on call sites
/root/reference/src/tracking/train_utils.py:174-192, /root/reference/src/render/renderer.py:18-23).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")
_lib = None


def build_oracle_lib(force: bool = False, f64: bool = False) -> str:
    """Compile ``gsr_oracle.c`` with gcc (recipe: oracle/Makefile).  ``f64``: the same file with every float a double."""
    src = os.path.join(_HERE, "gsr_oracle.c")
    name = "libgsr_oracle64.so" if f64 else "libgsr_oracle.so"
    path = os.path.join(_HERE, name)
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return path


def _cam_struct(real):
    class _Cam(C.Structure):
        _fields_ = [
            ("H", C.c_int), ("W", C.c_int),
            ("tanfovx", real), ("tanfovy", real),
            ("scale_modifier", real),
            ("sh_degree", C.c_int), ("M", C.c_int), ("prefiltered", C.c_int),
            ("bg", real * 3), ("view", real * 16), ("proj", real * 16),
            ("campos", real * 3),
        ]
    return _Cam


_Cam = _cam_struct(C.c_float)
_Cam64 = _cam_struct(C.c_double)
_libs = {}


def _load(f64: bool = False):
    global _lib
    if f64 not in _libs:
        lib = C.CDLL(build_oracle_lib(f64=f64))
        real = C.c_double if f64 else C.c_float
        fp = C.POINTER(real)
        lib.gsro_forward.restype = C.c_void_p
        lib.gsro_forward.argtypes = [C.POINTER(_Cam64 if f64 else _Cam), C.c_int, fp, fp, fp, fp, fp, fp, fp, C.c_int]
        lib.gsro_backward.restype = None
        lib.gsro_backward.argtypes = [C.c_void_p] + [fp] * 9 + [C.c_int]
        lib.gsro_free.argtypes = [C.c_void_p]
        lib.gsro_num_rendered.restype = C.c_uint32
        lib.gsro_num_rendered.argtypes = [C.c_void_p]
        lib.gsro_num_tiles.restype = C.c_int
        lib.gsro_num_tiles.argtypes = [C.c_void_p]
        for name in ("out_color", "out_depth", "radii", "means2D", "depths", "conic_opacity", "cov3D", "rgb",
                     "rect", "tiles_touched", "offsets", "keys", "point_list", "ranges", "final_T",
                     "n_contrib", "ambiguous"):
            fn = getattr(lib, "gsro_" + name)
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_void_p]
        lib.gsro_mark_visible.argtypes = [fp, C.c_int, fp, C.POINTER(C.c_uint8)]
        lib.gsro_set_overrides.restype = None
        lib.gsro_set_overrides.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _libs[f64] = lib
    if not f64:
        _lib = _libs[False]
    return _libs[f64]


@dataclass
class OracleCamera:
    """Plain-data mirror of the 11-field settings record (fields as in
    /root/reference/src/tracking/helpers.py:20-32), numpy instead of device tensors."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray  # 16 floats as stored by the reference (w2c transposed)
    projmatrix: np.ndarray
    sh_degree: int = 0
    campos: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    prefiltered: bool = False


def _f32(a, dtype=np.float32):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_double if a.dtype == np.float64 else C.c_float))


class TiledOracle:
    """One forward pass (+ optional backward) of oracle O2.  Holds intermediate state for tests."""

    def __init__(self, cam: OracleCamera, means3D, opacities, colors_precomp=None, scales=None,
                 rotations=None, shs=None, cov3D_precomp=None, nthreads: int = 1, f64: bool = False, decisions_of=None):
        """``f64``: the fp64 build of the same C file (the inputs are the fp32 values, widened).  ``decisions_of``: a finished fp32
        ``TiledOracle`` of the same inputs whose discrete decisions (radii, tile rects, binary32 depth sort keys) this run takes over,
        so that both blend the same tile lists (fp64 runs only)."""
        lib = _load(f64)
        self._lib = lib
        self._real = np.float64 if f64 else np.float32
        self.cam = cam
        self.P = int(np.asarray(means3D).shape[0])
        r_ = self._real
        self._in = dict(
            means3D=_f32(means3D, r_), scales=_f32(scales, r_), rot=_f32(rotations, r_),
            opac=_f32(np.asarray(opacities).reshape(-1), r_), colors=_f32(colors_precomp, r_), shs=_f32(shs, r_),
            cov3D=_f32(cov3D_precomp, r_))
        assert (self._in["colors"] is None) != (self._in["shs"] is None)
        assert (self._in["cov3D"] is None) != (self._in["scales"] is None)
        M = 0 if self._in["shs"] is None else int(self._in["shs"].shape[1])
        self.M = M
        c = (_Cam64 if f64 else _Cam)()
        c.H, c.W = int(cam.image_height), int(cam.image_width)
        c.tanfovx, c.tanfovy = float(cam.tanfovx), float(cam.tanfovy)
        c.scale_modifier = float(cam.scale_modifier)
        c.sh_degree, c.M, c.prefiltered = int(cam.sh_degree), M, int(bool(cam.prefiltered))
        c.bg[:] = [float(x) for x in np.asarray(cam.bg).reshape(-1)[:3]]
        c.view[:] = [float(x) for x in np.asarray(cam.viewmatrix, dtype=np.float32).reshape(-1)[:16]]
        c.proj[:] = [float(x) for x in np.asarray(cam.projmatrix, dtype=np.float32).reshape(-1)[:16]]
        c.campos[:] = [float(x) for x in np.asarray(cam.campos).reshape(-1)[:3]]
        self._c = c
        self.nthreads = nthreads
        i = self._in
        keep = None
        if decisions_of is not None:
            assert f64 and decisions_of._real == np.float32
            keep = (np.ascontiguousarray(decisions_of.radii), np.ascontiguousarray(decisions_of.rect), np.ascontiguousarray(decisions_of.depths))
            lib.gsro_set_overrides(keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data)
        try:
            self._ctx = self._forward(lib, c, i, nthreads)
        finally:
            if keep is not None:
                lib.gsro_set_overrides(None, None, None)
        self.H, self.W = c.H, c.W
        self.num_rendered = int(lib.gsro_num_rendered(self._ctx))
        self.num_tiles = int(lib.gsro_num_tiles(self._ctx))

    def _forward(self, lib, c, i, nthreads):
        return lib.gsro_forward(C.byref(c), self.P, _ptr(i["means3D"]), _ptr(i["scales"]), _ptr(i["rot"]),
                                     _ptr(i["opac"]), _ptr(i["colors"]), _ptr(i["shs"]), _ptr(i["cov3D"]),
                                     nthreads)

    def __del__(self):
        if getattr(self, "_ctx", None):
            self._lib.gsro_free(self._ctx)
            self._ctx = None

    def _get(self, name, dtype, shape):
        if dtype == np.float32:
            dtype = self._real
        p = getattr(self._lib, "gsro_" + name)(self._ctx)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    # ---- outputs
    @property
    def color(self): return self._get("out_color", np.float32, (3, self.H, self.W))
    @property
    def depth(self): return self._get("out_depth", np.float32, (1, self.H, self.W))
    @property
    def radii(self): return self._get("radii", np.int32, (self.P,))
    # ---- intermediates
    @property
    def means2D(self): return self._get("means2D", np.float32, (self.P, 2))
    @property
    def depths(self): return self._get("depths", np.float32, (self.P,))
    @property
    def conic_opacity(self): return self._get("conic_opacity", np.float32, (self.P, 4))
    @property
    def cov3D(self): return self._get("cov3D", np.float32, (self.P, 6))
    @property
    def rgb(self): return self._get("rgb", np.float32, (self.P, 3))
    @property
    def rect(self): return self._get("rect", np.int32, (self.P, 4))
    @property
    def tiles_touched(self): return self._get("tiles_touched", np.uint32, (self.P,))
    @property
    def offsets(self): return self._get("offsets", np.uint32, (self.P + 1,))
    @property
    def keys(self): return self._get("keys", np.uint64, (self.num_rendered,))
    @property
    def point_list(self): return self._get("point_list", np.uint32, (self.num_rendered,))
    @property
    def ranges(self): return self._get("ranges", np.uint32, (self.num_tiles, 2))
    @property
    def final_T(self): return self._get("final_T", np.float32, (self.H, self.W))
    @property
    def n_contrib(self): return self._get("n_contrib", np.uint32, (self.H, self.W))
    @property
    def ambiguous(self): return self._get("ambiguous", np.uint8, (self.H, self.W)).astype(bool)

    def backward(self, dL_dcolor, nthreads: Optional[int] = None):
        """Returns dict of gradients (numpy fp32)."""
        r_ = self._real
        g = _f32(dL_dcolor, r_).reshape(3, self.H, self.W)
        P, M = self.P, self.M
        out = dict(
            means3D=np.zeros((P, 3), r_), means2D=np.zeros((P, 3), r_),
            colors_precomp=np.zeros((P, 3), r_), opacities=np.zeros((P, 1), r_),
            scales=np.zeros((P, 3), r_), rotations=np.zeros((P, 4), r_),
            cov3D_precomp=np.zeros((P, 6), r_),
            shs=np.zeros((P, max(M, 1), 3), r_))
        self._lib.gsro_backward(
            self._ctx, _ptr(g), _ptr(out["means3D"]), _ptr(out["means2D"]), _ptr(out["colors_precomp"]),
            _ptr(out["opacities"]), _ptr(out["scales"]), _ptr(out["rotations"]), _ptr(out["cov3D_precomp"]),
            _ptr(out["shs"]) if M > 0 else None, self.nthreads if nthreads is None else nthreads)
        if M == 0:
            out["shs"] = None
        return out


def mark_visible(viewmatrix, means3D) -> np.ndarray:
    lib = _load()
    v = _f32(np.asarray(viewmatrix).reshape(-1)[:16])
    m = _f32(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    lib.gsro_mark_visible(_ptr(v), m.shape[0], _ptr(m), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)
