"""TEST INFRASTRUCTURE -- ctypes front-end of ``oracle/raycast_oracle.c`` (see the header of that file)."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    out = os.path.join(_HERE, "_build", "liboracle.so")
    src = os.path.join(_HERE, "raycast_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def raycast_f64(verts, tris, starts, dirs, max_dist=1e6):
    """Brute-force fp64 Moller-Trumbore closest hit.  Returns (hits f32[R,3], t f64[R], face i32[R])."""
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.uint32)
    starts = np.ascontiguousarray(starts, np.float32).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    R = starts.shape[0]
    hits = np.empty((R, 3), np.float32)
    t = np.empty(R, np.float64)
    face = np.empty(R, np.int32)
    _lib().imxo_raycast_f64(_p(verts), _p(tris), ctypes.c_int64(len(tris)), _p(starts), _p(dirs), ctypes.c_int64(R),
                            ctypes.c_double(max_dist), _p(hits), _p(t), _p(face))
    return hits, t, face


def raycast_woop_f32(verts, tris, starts, dirs, max_dist=1e6):
    """Brute-force fp32 Woop watertight closest hit.  Returns (hits f32[R,3], t f32[R], face i32[R])."""
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.ascontiguousarray(tris, np.uint32)
    starts = np.ascontiguousarray(starts, np.float32).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    R = starts.shape[0]
    hits = np.empty((R, 3), np.float32)
    t = np.empty(R, np.float32)
    face = np.empty(R, np.int32)
    _lib().imxo_raycast_woop_f32(_p(verts), _p(tris), ctypes.c_int64(len(tris)), _p(starts), _p(dirs),
                                 ctypes.c_int64(R), ctypes.c_float(max_dist), _p(hits), _p(t), _p(face))
    return hits, t, face
