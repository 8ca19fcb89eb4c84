"""ctypes binding of oracle/libmrt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path never does.  PARITY UNPINNED: see mrt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import mrt_pkg  # noqa: E402

mrt = mrt_pkg.load()
SO = HERE / "libmrt_oracle.so"
SO_FMA = HERE / "libmrt_oracle_fma.so"      # same source, multiply-adds contracted (see oracle/Makefile)
SO_ULP = HERE / "libmrt_oracle_ulp.so"      # same source, built-ins moved by whole ulps (see oracle/Makefile)
_lib = None
_lib_fma = None
_lib_ulp = None


def build():
    subprocess.run(["make", "-C", str(HERE), "-s", "all"], check=True)


def _load(so):
    if True:
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        vp = C.c_void_p
        L.mrt_oracle_render.argtypes = [C.POINTER(mrt.SceneStruct), C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp]
        L.mrt_oracle_camera_basis.argtypes = [vp] * 5
        L.mrt_oracle_camera_basis.restype = None
        L.mrt_oracle_to_byte.argtypes = [C.c_float]
        L.mrt_oracle_to_byte.restype = C.c_uint8
        for name in ("uv", "ortho_viewrays", "persp_viewrays", "intersect", "flat_shader", "phong_shader"):
            getattr(L, "mrt_oracle_" + name).restype = None
        L.mrt_oracle_uv.argtypes = [vp, vp, C.c_int, vp]
        L.mrt_oracle_ortho_viewrays.argtypes = [vp, vp, C.c_int, vp]
        L.mrt_oracle_persp_viewrays.argtypes = [vp, vp, C.c_int, C.c_float, vp]
        L.mrt_oracle_intersect.argtypes = [vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, vp, vp]

        class Rgb(C.Structure):
            _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("a", C.c_float)]
        L.Rgb = Rgb
        L.mrt_oracle_flat_shader.argtypes = [vp, C.c_int, Rgb, vp]
        L.mrt_oracle_phong_shader.argtypes = [vp, vp, C.c_int, Rgb, vp, vp, C.c_int, vp]
    return L


def lib():
    """The oracle: one rounding per operator."""
    global _lib
    if _lib is None:
        _lib = _load(SO)
    return _lib


def lib_fma():
    """The contracted build: NOT the oracle, only the measure of the reference's own latitude."""
    global _lib_fma
    if _lib_fma is None:
        _lib_fma = _load(SO_FMA)
    return _lib_fma


def lib_ulp(mode=0):
    """The build whose built-ins (`/`, sqrt, normalize, dot, powr) are off by whole ulps inside
    OpenCL 1.2's limits: NOT the oracle, only the measure of the reference's own latitude.
    mode 0: pseudo-random offsets, 1: all up, 2: all down."""
    global _lib_ulp
    if _lib_ulp is None:
        _lib_ulp = _load(SO_ULP)
        _lib_ulp.mrt_oracle_set_ulp_mode.argtypes = [C.c_int]
        _lib_ulp.mrt_oracle_set_ulp_mode.restype = None
    _lib_ulp.mrt_oracle_set_ulp_mode(int(mode))
    return _lib_ulp


def render(scene, rows=None, threads=1, want_rgbf=False, contracted=False, ulp_mode=None):
    """Returns dict(rgba[y,x,4], ids[y,x], t[y,x], rgbf[y,x,3] or None); rows outside
    the requested range are left zero / -1.  contracted: through the build whose multiply-adds
    are fused (lib_fma); ulp_mode 0 / 1 / 2: through the build whose built-ins are moved by whole
    ulps (lib_ulp)."""
    st = scene.struct
    h, w = st.yres, st.xres
    r0, r1 = (0, h) if rows is None else rows
    rgba = np.zeros((h, w, 4), np.uint8)
    ids = np.full((h, w), -1, np.int32)
    t = np.full((h, w), -1.0, np.float32)
    rgbf = np.zeros((h, w, 3), np.float32) if want_rgbf else None
    assert not (contracted and ulp_mode is not None)
    L = lib_ulp(ulp_mode) if ulp_mode is not None else (lib_fma() if contracted else lib())
    rc = L.mrt_oracle_render(C.byref(st), r0, r1, threads, rgba.ctypes.data, ids.ctypes.data,
                                 t.ctypes.data, rgbf.ctypes.data if want_rgbf else None)
    if rc != 0:
        raise RuntimeError("mrt_oracle_render rejected the scene")
    return {"rgba": rgba, "ids": ids, "t": t, "rgbf": rgbf}


def camera_basis(view, up):
    view = np.asarray(view, np.float64)
    up = np.asarray(up, np.float64)
    u, v, w = (np.zeros(3) for _ in range(3))
    lib().mrt_oracle_camera_basis(view.ctypes.data, up.ctypes.data, u.ctypes.data, v.ctypes.data, w.ctypes.data)
    return u, v, w


def to_byte(x):
    return int(lib().mrt_oracle_to_byte(float(x)))


def stage_uv(pos, vp):
    out = np.zeros(len(pos), mrt.CL_DTYPES["float2"])
    lib().mrt_oracle_uv(pos.ctypes.data, vp.ctypes.data, len(pos), out.ctypes.data)
    return out


def stage_viewrays(cam, uv, focal=None, out=None):
    rays = np.zeros(len(uv), mrt.CL_DTYPES["ray"]) if out is None else out
    if focal is None:
        lib().mrt_oracle_ortho_viewrays(cam.ctypes.data, uv.ctypes.data, len(uv), rays.ctypes.data)
    else:
        lib().mrt_oracle_persp_viewrays(cam.ctypes.data, uv.ctypes.data, len(uv), float(focal), rays.ctypes.data)
    return rays


def stage_intersect(objects, rays, t_upper, t_lower, out=None):
    xs = np.zeros(len(rays), mrt.CL_DTYPES["intersect"]) if out is None else out
    ids = np.zeros(len(rays), np.int32)
    lib().mrt_oracle_intersect(objects.ctypes.data if len(objects) else None, len(objects), rays.ctypes.data,
                               len(rays), float(t_upper), float(t_lower), xs.ctypes.data, ids.ctypes.data)
    return xs, ids


def stage_flat(xs, background, out=None):
    res = np.zeros(len(xs), mrt.CL_DTYPES["rgb"]) if out is None else out
    lib().mrt_oracle_flat_shader(xs.ctypes.data, len(xs), lib().Rgb(*background), res.ctypes.data)
    return res


def stage_phong(xs, phong, background, cam, lights, out=None):
    res = np.zeros(len(xs), mrt.CL_DTYPES["rgb"]) if out is None else out
    lib().mrt_oracle_phong_shader(xs.ctypes.data, phong.ctypes.data, len(xs), lib().Rgb(*background),
                                  cam.ctypes.data, lights.ctypes.data if len(lights) else None, len(lights),
                                  res.ctypes.data)
    return res
