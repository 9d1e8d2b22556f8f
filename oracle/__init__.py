"""CPU oracle of the 2DGS surfel rasteriser -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported CPU baseline.  The product path
(``lara_amd``, ``diff_surfel_rasterization``) never does; it fails loudly without its HIP library.

PARITY UNPINNED: see the header of ``surfel_oracle.c`` -- the reference's rasteriser sources are
absent (empty un-pinned submodule, /root/reference/.gitmodules:1-3) and the reference ships no
golden vectors for this boundary, so this oracle restates the published 2DGS algorithm and is
anchored on the reference's call site (lightning/renderer_2dgs.py:119-139,209-242), on
self-made known-answer tests and on an fp64 autograd restatement (``autograd_ref.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsurfel_oracle.so")
_lib = None


class _Cfg(ctypes.Structure):
    _fields_ = [
        ("P", ctypes.c_int32),
        ("sh_degree", ctypes.c_int32),
        ("sh_coeffs", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("tan_fovx", ctypes.c_float),
        ("tan_fovy", ctypes.c_float),
        ("scale_modifier", ctypes.c_float),
        ("bg", ctypes.c_float * 3),
        ("viewmatrix", ctypes.c_float * 16),
        ("projmatrix", ctypes.c_float * 16),
        ("campos", ctypes.c_float * 3),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Returns the library path."""
    src = os.path.join(_HERE, "surfel_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_forward.restype = ctypes.c_void_p
        lib.oracle_forward.argtypes = [ctypes.c_void_p] * 11
        lib.oracle_backward.restype = None
        lib.oracle_backward.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_int] + [ctypes.c_void_p] * 8
        lib.oracle_free.argtypes = [ctypes.c_void_p]
        lib.oracle_num_rendered.restype = ctypes.c_int64
        lib.oracle_num_rendered.argtypes = [ctypes.c_void_p]
        lib.oracle_blended_pairs.restype = ctypes.c_int64
        lib.oracle_blended_pairs.argtypes = [ctypes.c_void_p]
        lib.oracle_state_ptr.restype = ctypes.c_void_p
        lib.oracle_state_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.oracle_mark_visible.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3
        _lib = lib
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@dataclass
class View:
    """The 12 fields of GaussianRasterizationSettings (renderer_2dgs.py:124-137), as numpy."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    sh_degree: int
    campos: np.ndarray
    prefiltered: bool = False
    debug: bool = False


@dataclass
class Result:
    color: np.ndarray
    allmap: np.ndarray
    radii: np.ndarray
    num_rendered: int
    blended_pairs: int      # (pixel, splat) pairs the composite blended (alpha >= 1/255, before the T < 1e-4 stop)
    # saved state, for bit-exact comparison of the integer stages and for backward
    transMats: np.ndarray
    normal_opacity: np.ndarray
    rgb: np.ndarray
    means2D: np.ndarray
    depths: np.ndarray
    tiles_touched: np.ndarray
    rect: np.ndarray
    clamped: np.ndarray
    point_offsets: np.ndarray
    keys_sorted: np.ndarray
    point_list: np.ndarray
    ranges: np.ndarray
    final_T: np.ndarray
    n_contrib: np.ndarray
    _handle: int = field(default=0, repr=False)
    _cfg: object = field(default=None, repr=False)
    _inputs: tuple = field(default=(), repr=False)

    def __del__(self):
        if self._handle and _lib is not None:
            _lib.oracle_free(self._handle)
            self._handle = 0


def _make_cfg(view: View, P: int, M: int) -> _Cfg:
    c = _Cfg()
    c.P, c.sh_degree, c.sh_coeffs = int(P), int(view.sh_degree), int(M)
    c.H, c.W = int(view.image_height), int(view.image_width)
    c.tan_fovx, c.tan_fovy = float(view.tanfovx), float(view.tanfovy)
    c.scale_modifier = float(view.scale_modifier)
    c.bg[:] = [float(x) for x in np.asarray(view.bg, dtype=np.float32).reshape(3)]
    c.viewmatrix[:] = [float(x) for x in np.asarray(view.viewmatrix, dtype=np.float32).reshape(16)]
    c.projmatrix[:] = [float(x) for x in np.asarray(view.projmatrix, dtype=np.float32).reshape(16)]
    c.campos[:] = [float(x) for x in np.asarray(view.campos, dtype=np.float32).reshape(3)]
    return c


def _view_np(ptr, dtype, shape):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype=dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


def forward(view: View, means3D, opacities, shs=None, colors_precomp=None, scales=None,
            rotations=None, transmat_precomp=None) -> Result:
    """CPU restatement of GaussianRasterizer.forward (renderer_2dgs.py:209-218)."""
    lib = _load()
    means3D = _f32(means3D).reshape(-1, 3)
    P = means3D.shape[0]
    if (shs is None) == (colors_precomp is None):
        raise ValueError("provide exactly one of shs / colors_precomp")
    if ((scales is None or rotations is None) and transmat_precomp is None) or \
            ((scales is not None or rotations is not None) and transmat_precomp is not None):
        raise ValueError("provide exactly one of scale/rotation pair or precomputed transMat")
    shs = _f32(shs)
    M = 0
    if shs is not None:
        shs = shs.reshape(P, -1, 3)
        M = shs.shape[1]
    colors_precomp = _f32(colors_precomp, (P, 3)) if colors_precomp is not None else None
    opacities = _f32(opacities).reshape(P)
    scales = _f32(scales, (P, 2)) if scales is not None else None
    rotations = _f32(rotations, (P, 4)) if rotations is not None else None
    transmat_precomp = _f32(transmat_precomp, (P, 9)) if transmat_precomp is not None else None
    H, W = int(view.image_height), int(view.image_width)
    color = np.zeros((3, H, W), np.float32)
    allmap = np.zeros((7, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    cfg = _make_cfg(view, P, M)
    h = lib.oracle_forward(ctypes.addressof(cfg), _ptr(means3D), _ptr(shs), _ptr(colors_precomp),
                           _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(transmat_precomp),
                           _ptr(color), _ptr(allmap), _ptr(radii))
    D = int(lib.oracle_num_rendered(h))
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    sp = lambda k: lib.oracle_state_ptr(h, k)
    return Result(
        color=color, allmap=allmap, radii=radii, num_rendered=D, blended_pairs=int(lib.oracle_blended_pairs(h)),
        transMats=_view_np(sp(0), np.float32, (P, 9)),
        normal_opacity=_view_np(sp(1), np.float32, (P, 4)),
        rgb=_view_np(sp(2), np.float32, (P, 3)),
        means2D=_view_np(sp(3), np.float32, (P, 2)),
        depths=_view_np(sp(4), np.float32, (P,)),
        tiles_touched=_view_np(sp(6), np.uint32, (P,)),
        rect=_view_np(sp(7), np.uint32, (P, 4)),
        clamped=_view_np(sp(8), np.uint8, (P, 3)),
        point_offsets=_view_np(sp(9), np.uint32, (P,)),
        keys_sorted=_view_np(sp(10), np.uint64, (D,)),
        point_list=_view_np(sp(11), np.uint32, (D,)),
        ranges=_view_np(sp(12), np.uint32, (tiles, 2)),
        final_T=_view_np(sp(13), np.float32, (3, H, W)),
        n_contrib=_view_np(sp(14), np.uint32, (2, H, W)),
        _handle=h, _cfg=cfg,
        _inputs=(means3D, shs, colors_precomp, scales, rotations, transmat_precomp, M),
    )


def backward(res: Result, dL_dcolor, dL_dallmap, lowpass_depth_quirk: bool = True) -> dict:
    """CPU restatement of the rasteriser backward (per-pixel reverse traversal + preprocess VJP)."""
    lib = _load()
    means3D, shs, colors_precomp, scales, rotations, transmat_precomp, M = res._inputs
    P = means3D.shape[0]
    H, W = res.color.shape[1:]
    dL_dcolor = _f32(dL_dcolor, (3, H, W))
    dL_dallmap = _f32(dL_dallmap, (7, H, W))
    n = max(P, 1)
    g = {
        "means3D": np.zeros((n, 3), np.float32),
        "means2D": np.zeros((n, 3), np.float32),
        "shs": np.zeros((n, max(M, 1), 3), np.float32),
        "colors_precomp": np.zeros((n, 3), np.float32),
        "opacities": np.zeros((n,), np.float32),
        "scales": np.zeros((n, 2), np.float32),
        "rotations": np.zeros((n, 4), np.float32),
        "transmat_precomp": np.zeros((n, 9), np.float32),
    }
    lib.oracle_backward(ctypes.addressof(res._cfg), res._handle, _ptr(means3D), _ptr(shs),
                        _ptr(colors_precomp), _ptr(scales), _ptr(rotations), _ptr(transmat_precomp),
                        _ptr(dL_dcolor), _ptr(dL_dallmap), int(bool(lowpass_depth_quirk)),
                        _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]),
                        _ptr(g["colors_precomp"]), _ptr(g["opacities"]), _ptr(g["scales"]),
                        _ptr(g["rotations"]), _ptr(g["transmat_precomp"]))
    out = {k: v[:P] for k, v in g.items()}
    out["shs"] = out["shs"][:, :M]
    return out


def mark_visible(means3D, viewmatrix) -> np.ndarray:
    lib = _load()
    means3D = _f32(means3D).reshape(-1, 3)
    vm = _f32(viewmatrix).reshape(16)
    out = np.zeros((means3D.shape[0],), np.uint8)
    lib.oracle_mark_visible(means3D.shape[0], _ptr(means3D), _ptr(vm), _ptr(out))
    return out.astype(bool)
