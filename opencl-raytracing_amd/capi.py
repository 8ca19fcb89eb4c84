"""ctypes binding of libmrt_hip.so (include/mrt.h).

Loading is strict: if the shared library is missing or cannot be loaded the
import of :func:`lib` raises -- there is no Python or CPU fallback for any
entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent

FLAG_NONE = 0
FLAG_NO_PREFILTER = 1 << 0
FLAG_COUNT_EXACT = 1 << 1
FLAG_NO_CIRCLE = 1 << 3
FLAG_FULL_TEST = 1 << 5
FLAG_TRACE_NARROW = 1 << 6          # trace_kernel with 4x4 pixels per lane whatever the frame size
FLAG_TRACE_WIDE = 1 << 7            # ... with 8x4
FLAG_DEFER_NOALLOC_SETUP = 1 << 8   # mrt_create leaves the fixed array of the no-alloc entry point off the GPU
RESERVED_FLAG_BITS = (1 << 2) | (1 << 4) | (1 << 30)          # mrt_create refuses them
KERNEL_TRACE, KERNEL_TRACE_BLOCK, KERNEL_CULL, KERNEL_FULL, KERNEL_GRID, KERNEL_TRACE_WIDE = 0, 1, 2, 3, 4, 5    # mrt_stats.kernel
ACCEL_GRID, ACCEL_TILE_CULL, ACCEL_BLOCK_REJECT = 0, 1, 2                                 # mrt_accel_config.method
ACCEL_METHODS = {"grid": ACCEL_GRID, "cull": ACCEL_TILE_CULL, "block": ACCEL_BLOCK_REJECT}
ABI_VERSION = 8
PROJ_ORTHO, PROJ_PERSP = 0, 1
SHADER_FLAT, SHADER_PHONG = 0, 1

# Every symbol include/mrt.h declares.
EXPORTED_SYMBOLS = [
    "mrt_create", "mrt_destroy", "mrt_last_error", "mrt_abi_version", "mrt_set_scene", "mrt_set_camera",
    "mrt_render_rgba8", "mrt_render_debug", "mrt_band_rows", "mrt_balanced_band_rows", "mrt_render_bands_device",
    "mrt_bands_to_image", "mrt_scene_load_file", "mrt_scene_load_text", "mrt_scene_view",
    "mrt_scene_info_view", "mrt_scene_free", "mrt_write_png_rgb", "mrt_png_stream_begin", "mrt_png_stream_rows",
    "mrt_png_stream_end", "mrt_render_rgba8_rows", "mrt_multi_render_rgba8_rows", "mrt_stage_uv",
    "mrt_stage_ortho_viewrays", "mrt_stage_persp_viewrays", "mrt_stage_intersect",
    "mrt_stage_flat_shader", "mrt_stage_phong_shader", "mrt_stage_last_kernel_ms", "mrt_synchronize", "mrt_host_alloc", "mrt_host_free",
    "mrt_render_bands_device_noalloc",
    "mrt_multi_create", "mrt_multi_destroy", "mrt_multi_last_error", "mrt_multi_world", "mrt_multi_gather",
    "mrt_multi_note", "mrt_multi_set_scene", "mrt_multi_set_camera", "mrt_multi_render_rgba8",
    "mrt_accel_create", "mrt_accel_destroy", "mrt_accel_last_error", "mrt_accel_method", "mrt_accel_set_scene",
    "mrt_accel_set_camera", "mrt_accel_render_rgba8", "mrt_accel_render_debug", "mrt_accel_render_bands_device",
    "mrt_sequence_create", "mrt_sequence_destroy", "mrt_sequence_last_error", "mrt_sequence_in_flight",
    "mrt_sequence_set_scene", "mrt_sequence_submit", "mrt_sequence_submit_host", "mrt_sequence_wait",
]
GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_HOST = 0, 1, 2, 3
NOALLOC_MAX_KEYS = 8 * 1024 * 1024

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)


class SceneStruct(C.Structure):
    """mrt_scene."""
    _fields_ = [
        ("eye", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
        ("xres", C.c_uint32), ("yres", C.c_uint32),
        ("left", C.c_float), ("right", C.c_float), ("top", C.c_float), ("bottom", C.c_float),
        ("projection", C.c_int32), ("focal", C.c_float), ("shader", C.c_int32),
        ("ka", C.c_float), ("kd", C.c_float), ("ks", C.c_float),
        ("ambient", C.c_float * 3), ("background", C.c_float * 3),
        ("n_spheres", C.c_uint32), ("sph", _fp * 4), ("sph_mat", _up),
        ("n_tris", C.c_uint32), ("tri", _fp * 9), ("tri_mat", _up),
        ("n_materials", C.c_uint32), ("mat", _fp * 4),
        ("n_lights", C.c_uint32), ("light", _fp * 6),
    ]


class CameraStruct(C.Structure):
    """mrt_camera."""
    _fields_ = [
        ("eye", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
        ("xres", C.c_uint32), ("yres", C.c_uint32),
        ("left", C.c_float), ("right", C.c_float), ("top", C.c_float), ("bottom", C.c_float),
        ("projection", C.c_int32), ("focal", C.c_float),
    ]


class ConfigStruct(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    """mrt_stats."""
    _fields_ = [
        ("upload_ms", C.c_double), ("kernel_ms", C.c_double), ("trace_ms", C.c_double),
        ("shade_ms", C.c_double), ("download_ms", C.c_double),
        ("rays", C.c_uint64), ("prim_tests", C.c_uint64), ("exact_tests", C.c_uint64),
        ("kernel", C.c_uint32), ("prim_slices", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class AccelConfigStruct(C.Structure):
    """mrt_accel_config."""
    _fields_ = [("device", C.c_int32), ("method", C.c_uint32), ("flags", C.c_uint32)]


class AccelStats(C.Structure):
    """mrt_accel_stats; the fields of the embedded mrt_stats read through (stats.kernel, stats.trace_ms, ...)."""
    _fields_ = [("frame", Stats), ("build_ms", C.c_double), ("cells", C.c_uint64), ("refs", C.c_uint64)]

    def __getattr__(self, name):
        frame = C.Structure.__getattribute__(self, "frame")
        if name in {f for f, _ in Stats._fields_}:
            return getattr(frame, name)
        raise AttributeError(name)

    def as_dict(self):
        d = self.frame.as_dict()
        d.update(build_ms=self.build_ms, cells=self.cells, refs=self.refs)
        return d


class SequenceConfigStruct(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_uint32), ("in_flight", C.c_uint32)]


class MultiConfigStruct(C.Structure):
    """mrt_multi_config."""
    _fields_ = [("n_gpus", C.c_uint32), ("devices", C.POINTER(C.c_int32)), ("flags", C.c_uint32),
                ("band_rows", C.c_uint32), ("gather", C.c_uint32)]


class MultiStats(C.Structure):
    """mrt_multi_stats."""
    _fields_ = [("render_ms_max", C.c_double), ("render_ms_sum", C.c_double), ("total_ms", C.c_double),
                ("rays", C.c_uint64), ("prim_tests", C.c_uint64), ("n_gpus", C.c_uint32), ("gather", C.c_uint32)]


class SceneInfoStruct(C.Structure):
    _fields_ = [
        ("eye", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3), ("w", C.c_double * 3),
        ("left", C.c_double), ("top", C.c_double), ("right", C.c_double), ("bottom", C.c_double),
        ("focal", C.c_double), ("ka", C.c_double), ("kd", C.c_double), ("ks", C.c_double),
        ("n_lights", C.c_uint32), ("light_pos", C.POINTER(C.c_double)),
        ("light_intensity", C.POINTER(C.c_double)),
    ]


# numpy views of the packed AoS structs (mrt_cl_*), byte offsets as in the header.
_f3 = (np.float32, (4,))
CL_DTYPES = {
    "float3": np.dtype([("s", np.float32, (4,))]),
    "float2": np.dtype([("s", np.float32, (2,))]),
    "rgb": np.dtype([("r", np.float32), ("g", np.float32), ("b", np.float32), ("a", np.float32)]),
    "light": np.dtype({"names": ["location", "color", "intensity"], "formats": [_f3, _f3, np.float32],
                       "offsets": [0, 16, 32], "itemsize": 48}),
    "cam": np.dtype({"names": ["eye", "u", "v", "w"], "formats": [_f3] * 4, "offsets": [0, 16, 32, 48],
                     "itemsize": 64}),
    "vp": np.dtype({"names": ["x_resolution", "y_resolution", "left", "right", "top", "bottom"],
                    "formats": [np.float32] * 6, "offsets": [0, 4, 8, 12, 16, 20], "itemsize": 32}),
    "ray": np.dtype({"names": ["origin", "direction"], "formats": [_f3, _f3], "offsets": [0, 16],
                     "itemsize": 32}),
    "phong": np.dtype({"names": ["ambient_coef", "diffuse_coef", "specular_coef", "ambient_color"],
                       "formats": [np.float32, np.float32, np.float32, _f3], "offsets": [0, 4, 8, 12],
                       "itemsize": 32}),
    "object": np.dtype({
        "names": ["tri_color", "tri_shininess", "a", "b", "c", "sph_color", "sph_shininess", "center",
                  "radius", "is_triangle", "is_circle"],
        "formats": [_f3, np.float32, _f3, _f3, _f3, _f3, np.float32, _f3, np.float32, np.uint8, np.uint8],
        "offsets": [0, 16, 20, 36, 52, 80, 96, 100, 116, 128, 129], "itemsize": 144}),
    "intersect": np.dtype({
        "names": ["location", "normal", "intersects", "t_", "dummy2", "is_triangle", "is_circle",
                  "shininess", "color"],
        "formats": [_f3, _f3, np.uint8, np.float32, np.float32, np.uint8, np.uint8, np.float32, _f3],
        "offsets": [0, 16, 32, 33, 37, 57, 58, 60, 64], "itemsize": 80}),
}


class MrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"mrt error {code}: {message}")
        self.code = code
        self.message = message


ROWS_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)      # mrt_rows_ready_fn


def lib_path() -> Path:
    return Path(os.environ.get("MRT_HIP_LIB", PKG_DIR / "libmrt_hip.so"))


_lib = None


def lib():
    """The loaded libmrt_hip.so with argument types declared.  Raises OSError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP runtime (same SONAME as the system
    # one).  Two HIP runtimes in one process cannot both open the GPU, so when
    # torch is installed it is imported first and this library binds to the
    # runtime torch has already loaded.
    if "torch" not in sys.modules and os.environ.get("MRT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = lib_path()
    if not path.exists():
        raise OSError(f"{path} not found: build it with `make -C {PKG_DIR}` (hipcc, gfx950)")
    L = C.CDLL(str(path))
    vp = C.c_void_p
    L.mrt_create.argtypes = [C.POINTER(ConfigStruct), C.POINTER(vp)]
    L.mrt_destroy.argtypes = [vp]
    L.mrt_destroy.restype = None
    L.mrt_last_error.argtypes = [vp]
    L.mrt_last_error.restype = C.c_char_p
    L.mrt_set_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    L.mrt_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    L.mrt_synchronize.argtypes = [vp]
    L.mrt_set_slices_override_.argtypes = [vp, C.c_uint32]
    L.mrt_set_slices_override_.restype = None
    L.mrt_stage_last_t_switches_.argtypes = [vp]
    L.mrt_stage_last_t_switches_.restype = C.c_uint32
    L.mrt_render_rgba8.argtypes = [vp, vp, C.POINTER(Stats)]
    L.mrt_render_debug.argtypes = [vp, vp, vp, vp, C.POINTER(Stats)]
    L.mrt_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.mrt_host_free.argtypes = [vp]
    L.mrt_host_free.restype = None
    L.mrt_band_rows.argtypes = [C.c_uint32] * 4
    L.mrt_band_rows.restype = C.c_uint32
    L.mrt_balanced_band_rows.argtypes = [C.c_uint32] * 2
    L.mrt_balanced_band_rows.restype = C.c_uint32
    L.mrt_render_bands_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp,
                                          C.POINTER(Stats)]
    L.mrt_render_bands_device_noalloc.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp]
    L.mrt_multi_create.argtypes = [C.POINTER(MultiConfigStruct), C.POINTER(vp)]
    L.mrt_multi_destroy.argtypes = [vp]
    L.mrt_multi_destroy.restype = None
    L.mrt_multi_last_error.argtypes = [vp]
    L.mrt_multi_last_error.restype = C.c_char_p
    L.mrt_multi_note.argtypes = [vp]
    L.mrt_multi_note.restype = C.c_char_p
    L.mrt_multi_world.argtypes = [vp]
    L.mrt_multi_world.restype = C.c_uint32
    L.mrt_multi_gather.argtypes = [vp]
    L.mrt_multi_gather.restype = C.c_uint32
    L.mrt_multi_set_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    L.mrt_multi_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    L.mrt_multi_render_rgba8.argtypes = [vp, vp, C.POINTER(MultiStats)]
    L.mrt_sequence_create.argtypes = [C.POINTER(SequenceConfigStruct), C.POINTER(vp)]
    L.mrt_sequence_destroy.argtypes = [vp]
    L.mrt_sequence_destroy.restype = None
    L.mrt_sequence_last_error.argtypes = [vp]
    L.mrt_sequence_last_error.restype = C.c_char_p
    L.mrt_sequence_in_flight.argtypes = [vp]
    L.mrt_sequence_in_flight.restype = C.c_uint32
    L.mrt_sequence_set_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    L.mrt_sequence_submit.argtypes = [vp, C.POINTER(CameraStruct), C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                      C.c_size_t, C.POINTER(vp)]
    L.mrt_sequence_submit_host.argtypes = [vp, C.POINTER(CameraStruct), vp]
    L.mrt_sequence_wait.argtypes = [vp]
    L.mrt_accel_create.argtypes = [C.POINTER(AccelConfigStruct), C.POINTER(vp)]
    L.mrt_accel_destroy.argtypes = [vp]
    L.mrt_accel_destroy.restype = None
    L.mrt_accel_last_error.argtypes = [vp]
    L.mrt_accel_last_error.restype = C.c_char_p
    L.mrt_accel_method.argtypes = [vp]
    L.mrt_accel_method.restype = C.c_uint32
    L.mrt_accel_set_scene.argtypes = [vp, C.POINTER(SceneStruct)]
    L.mrt_accel_set_camera.argtypes = [vp, C.POINTER(CameraStruct)]
    L.mrt_accel_render_rgba8.argtypes = [vp, vp, C.POINTER(AccelStats)]
    L.mrt_accel_render_debug.argtypes = [vp, vp, vp, vp, C.POINTER(AccelStats)]
    L.mrt_accel_render_bands_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp,
                                                C.POINTER(AccelStats)]
    L.mrt_bands_to_image.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.mrt_scene_load_file.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.mrt_scene_load_text.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.mrt_scene_view.argtypes = [vp]
    L.mrt_scene_view.restype = C.POINTER(SceneStruct)
    L.mrt_scene_info_view.argtypes = [vp]
    L.mrt_scene_info_view.restype = C.POINTER(SceneInfoStruct)
    L.mrt_scene_free.argtypes = [vp]
    L.mrt_scene_free.restype = None
    L.mrt_write_png_rgb.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.mrt_png_stream_begin.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.mrt_png_stream_rows.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mrt_png_stream_end.argtypes = [vp]
    L.mrt_render_rgba8_rows.argtypes = [vp, vp, C.c_uint32, ROWS_READY_FN, vp, C.POINTER(Stats)]
    L.mrt_multi_render_rgba8_rows.argtypes = [vp, vp, ROWS_READY_FN, vp, C.POINTER(MultiStats)]
    L.mrt_stage_uv.argtypes = [vp, vp, vp, C.c_int, vp]
    L.mrt_stage_ortho_viewrays.argtypes = [vp, vp, vp, C.c_int, vp]
    L.mrt_stage_persp_viewrays.argtypes = [vp, vp, vp, C.c_int, C.c_float, vp]
    L.mrt_stage_intersect.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, vp]

    class _Rgb(C.Structure):
        _fields_ = [("r", C.c_float), ("g", C.c_float), ("b", C.c_float), ("a", C.c_float)]
    L._Rgb = _Rgb
    L.mrt_stage_flat_shader.argtypes = [vp, vp, C.c_int, _Rgb, vp]
    L.mrt_stage_phong_shader.argtypes = [vp, vp, vp, C.c_int, _Rgb, vp, vp, C.c_int, vp]
    L.mrt_stage_last_kernel_ms.argtypes = [vp]
    L.mrt_stage_last_kernel_ms.restype = C.c_double
    _lib = L
    return L


class Scene:
    """Owns numpy planes and exposes them as an mrt_scene struct."""

    def __init__(self, *, eye, u, v, w, xres, yres, left, right, top, bottom, projection, focal=0.0,
                 shader=SHADER_FLAT, ka=0.0, kd=0.0, ks=0.0, ambient=(0, 0, 0), background=(0, 0, 0),
                 sph=None, sph_mat=None, tri=None, tri_mat=None, mat=None, light=None):
        f32 = np.float32
        self.sph = np.ascontiguousarray(sph if sph is not None else np.zeros((4, 0)), dtype=f32)
        self.tri = np.ascontiguousarray(tri if tri is not None else np.zeros((9, 0)), dtype=f32)
        self.mat = np.ascontiguousarray(mat if mat is not None else np.zeros((4, 0)), dtype=f32)
        self.light = np.ascontiguousarray(light if light is not None else np.zeros((6, 0)), dtype=f32)
        self.sph_mat = np.ascontiguousarray(sph_mat if sph_mat is not None else np.zeros(0), dtype=np.uint32)
        self.tri_mat = np.ascontiguousarray(tri_mat if tri_mat is not None else np.zeros(0), dtype=np.uint32)
        assert self.sph.shape[0] == 4 and self.tri.shape[0] == 9 and self.mat.shape[0] == 4
        assert self.light.shape[0] == 6
        assert self.sph_mat.shape[0] == self.sph.shape[1] and self.tri_mat.shape[0] == self.tri.shape[1]
        s = SceneStruct()
        for name, val in (("eye", eye), ("u", u), ("v", v), ("w", w), ("ambient", ambient),
                          ("background", background)):
            arr = np.asarray(val, dtype=f32)
            setattr(s, name, (C.c_float * 3)(*[float(x) for x in arr]))
        s.xres, s.yres = int(xres), int(yres)
        s.left, s.right, s.top, s.bottom = (float(f32(x)) for x in (left, right, top, bottom))
        s.projection, s.focal, s.shader = int(projection), float(f32(focal)), int(shader)
        s.ka, s.kd, s.ks = (float(f32(x)) for x in (ka, kd, ks))
        s.n_spheres, s.n_tris = self.sph.shape[1], self.tri.shape[1]
        s.n_materials, s.n_lights = self.mat.shape[1], self.light.shape[1]

        def plane(a, k):
            return a[k].ctypes.data_as(_fp) if a.shape[1] else C.cast(None, _fp)
        for k in range(4):
            s.sph[k] = plane(self.sph, k)
            s.mat[k] = plane(self.mat, k)
        for k in range(9):
            s.tri[k] = plane(self.tri, k)
        for k in range(6):
            s.light[k] = plane(self.light, k)
        s.sph_mat = self.sph_mat.ctypes.data_as(_up)
        s.tri_mat = self.tri_mat.ctypes.data_as(_up)
        self.struct = s

    @classmethod
    def from_struct(cls, st: SceneStruct) -> "Scene":
        """Deep copy of an mrt_scene the library owns (e.g. a loaded scene's view)."""
        def planes(ptrs, k, n):
            return np.stack([np.ctypeslib.as_array(ptrs[i], shape=(n,)).copy() if n else np.zeros(0, np.float32)
                             for i in range(k)]) if n else np.zeros((k, 0), np.float32)

        def ids(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        return cls(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=st.xres, yres=st.yres,
                   left=st.left, right=st.right, top=st.top, bottom=st.bottom, projection=st.projection,
                   focal=st.focal, shader=st.shader, ka=st.ka, kd=st.kd, ks=st.ks, ambient=list(st.ambient),
                   background=list(st.background), sph=planes(st.sph, 4, st.n_spheres),
                   sph_mat=ids(st.sph_mat, st.n_spheres), tri=planes(st.tri, 9, st.n_tris),
                   tri_mat=ids(st.tri_mat, st.n_tris), mat=planes(st.mat, 4, st.n_materials),
                   light=planes(st.light, 6, st.n_lights))

    def with_resolution(self, xres, yres, left=None, right=None, top=None, bottom=None) -> "Scene":
        st = self.struct
        return Scene(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=xres, yres=yres,
                     left=st.left if left is None else left, right=st.right if right is None else right,
                     top=st.top if top is None else top, bottom=st.bottom if bottom is None else bottom,
                     projection=st.projection, focal=st.focal, shader=st.shader, ka=st.ka, kd=st.kd,
                     ks=st.ks, ambient=list(st.ambient), background=list(st.background), sph=self.sph,
                     sph_mat=self.sph_mat, tri=self.tri, tri_mat=self.tri_mat, mat=self.mat, light=self.light)

    @property
    def n_objects(self):
        return self.struct.n_spheres + self.struct.n_tris


class LoadedScene:
    """A scene parsed by the library's own loader (mrt_scene_load_*)."""

    def __init__(self, path=None, text=None):
        L = lib()
        handle = C.c_void_p()
        err = C.create_string_buffer(1024)
        if path is not None:
            rc = L.mrt_scene_load_file(str(path).encode(), C.byref(handle), err, len(err))
        else:
            data = text.encode() if isinstance(text, str) else bytes(text)
            rc = L.mrt_scene_load_text(data, len(data), C.byref(handle), err, len(err))
        if rc != 0:
            raise MrtError(rc, err.value.decode(errors="replace"))
        self._h = handle
        self.struct = L.mrt_scene_view(handle).contents
        self.info = L.mrt_scene_info_view(handle).contents

    def copy(self) -> Scene:
        return Scene.from_struct(self.struct)

    def close(self):
        if self._h:
            lib().mrt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """mrt_ctx bound to one GPU."""

    def __init__(self, device=-1, flags=FLAG_NONE):
        L = lib()
        self._h = C.c_void_p()
        cfg = ConfigStruct(device, flags)
        rc = L.mrt_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MrtError(rc, L.mrt_last_error(None).decode())
        self.xres = self.yres = 0

    def _check(self, rc):
        if rc != 0:
            raise MrtError(rc, lib().mrt_last_error(self._h).decode())

    def set_scene(self, scene):
        st = scene.struct
        self._check(lib().mrt_set_scene(self._h, C.byref(st)))
        self.xres, self.yres = st.xres, st.yres

    def set_camera(self, scene):
        """Takes the view (camera, viewport, projection) of `scene`, keeps everything else."""
        st = scene.struct
        cam = CameraStruct()
        for name in ("eye", "u", "v", "w"):
            setattr(cam, name, getattr(st, name))
        for name in ("xres", "yres", "left", "right", "top", "bottom", "projection", "focal"):
            setattr(cam, name, getattr(st, name))
        self._check(lib().mrt_set_camera(self._h, C.byref(cam)))
        self.xres, self.yres = st.xres, st.yres

    def synchronize(self):
        """Waits for everything the context has enqueued (record kernels, frames)."""
        self._check(lib().mrt_synchronize(self._h))

    def force_slices(self, n):
        """Tuning / test aid (not part of the public ABI): cut the primitive list of this context's
        frames into n slices or layers; 0 = the library's automatic choice."""
        lib().mrt_set_slices_override_(self._h, int(n))

    def render(self, want_ids=False, out=None):
        """Returns (rgba[y, x, 4] uint8, stats) or (rgba, ids, t, stats).  out: an RGBA8 array to
        render into (for example one from host_array(), which is page-locked)."""
        stats = Stats()
        img = np.empty((self.yres, self.xres, 4), np.uint8) if out is None else out
        if img.shape != (self.yres, self.xres, 4) or img.dtype != np.uint8 or not img.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        if not want_ids:
            self._check(lib().mrt_render_rgba8(self._h, img.ctypes.data, C.byref(stats)))
            return img, stats
        ids = np.empty((self.yres, self.xres), np.int32)
        t = np.empty((self.yres, self.xres), np.float32)
        self._check(lib().mrt_render_debug(self._h, img.ctypes.data, ids.ctypes.data, t.ctypes.data,
                                           C.byref(stats)))
        return img, ids, t, stats

    def render_rows(self, out, on_rows=None, strip_rows=0):
        """mrt_render_rgba8_rows: the frame into `out` strip by strip; on_rows(row_begin, row_end) is
        called as rows become final.  Returns stats."""
        stats = Stats()
        if out.shape != (self.yres, self.xres, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        cb = ROWS_READY_FN((lambda user, a, b: on_rows(a, b)) if on_rows else (lambda user, a, b: None))
        self._check(lib().mrt_render_rgba8_rows(self._h, out.ctypes.data, strip_rows, cb, None, C.byref(stats)))
        return stats

    def render_bands_device(self, band_rows_, rank, world, dev_ptr, nbytes, stream=0, timed=False):
        stats = Stats() if timed else None
        self._check(lib().mrt_render_bands_device(self._h, band_rows_, rank, world, dev_ptr, nbytes,
                                                  stream or None, C.byref(stats) if timed else None))
        return stats

    def render_bands_device_noalloc(self, band_rows_, rank, world, dev_ptr, nbytes, stream=0):
        self._check(lib().mrt_render_bands_device_noalloc(self._h, band_rows_, rank, world, dev_ptr, nbytes,
                                                          stream or None))

    # stage-level entry points on numpy structured arrays (CL_DTYPES)
    def stage_uv(self, pos, vp):
        out = np.zeros(len(pos), CL_DTYPES["float2"])
        self._check(lib().mrt_stage_uv(self._h, pos.ctypes.data, vp.ctypes.data, len(pos), out.ctypes.data))
        return out

    def stage_viewrays(self, cam, uv, focal=None, out=None):
        rays = np.zeros(len(uv), CL_DTYPES["ray"]) if out is None else out
        if focal is None:
            self._check(lib().mrt_stage_ortho_viewrays(self._h, cam.ctypes.data, uv.ctypes.data, len(uv),
                                                       rays.ctypes.data))
        else:
            self._check(lib().mrt_stage_persp_viewrays(self._h, cam.ctypes.data, uv.ctypes.data, len(uv),
                                                       float(focal), rays.ctypes.data))
        return rays

    def stage_intersect(self, objects, rays, t_upper, t_lower, out=None):
        xs = np.zeros(len(rays), CL_DTYPES["intersect"]) if out is None else out
        self._check(lib().mrt_stage_intersect(self._h, objects.ctypes.data if len(objects) else None,
                                              len(objects), rays.ctypes.data, len(rays), float(t_upper),
                                              float(t_lower), xs.ctypes.data))
        return xs

    def stage_t_switches(self):
        """Tuning / test aid (not part of the public ABI): waves of the last stage_intersect call that took
        up the filter with the fourth determinant."""
        return int(lib().mrt_stage_last_t_switches_(self._h))

    def stage_kernel_ms(self):
        """Device time of the kernels of the last stage_intersect call (HIP events)."""
        return float(lib().mrt_stage_last_kernel_ms(self._h))

    def stage_flat(self, xs, background, out=None):
        res = np.zeros(len(xs), CL_DTYPES["rgb"]) if out is None else out
        L = lib()
        self._check(L.mrt_stage_flat_shader(self._h, xs.ctypes.data, len(xs), L._Rgb(*background),
                                            res.ctypes.data))
        return res

    def stage_phong(self, xs, phong, background, cam, lights, out=None):
        res = np.zeros(len(xs), CL_DTYPES["rgb"]) if out is None else out
        L = lib()
        self._check(L.mrt_stage_phong_shader(self._h, xs.ctypes.data, phong.ctypes.data, len(xs),
                                             L._Rgb(*background), cam.ctypes.data,
                                             lights.ctypes.data if len(lights) else None, len(lights),
                                             res.ctypes.data))
        return res

    def close(self):
        if self._h:
            lib().mrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _camera_of(st):
    cam = CameraStruct()
    for name in ("eye", "u", "v", "w", "xres", "yres", "left", "right", "top", "bottom", "projection", "focal"):
        setattr(cam, name, getattr(st, name))
    return cam


class Accel:
    """mrt_accel: the same frames as a Context without testing every pair.  Same methods as Context
    (set_scene, set_camera, render, render_bands_device); stats are AccelStats."""

    def __init__(self, device=-1, method=ACCEL_GRID, flags=FLAG_NONE):
        L = lib()
        self._h = C.c_void_p()
        cfg = AccelConfigStruct(device, ACCEL_METHODS.get(method, method), flags)
        rc = L.mrt_accel_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MrtError(rc, L.mrt_accel_last_error(None).decode())
        self.method = L.mrt_accel_method(self._h)
        self.xres = self.yres = 0

    def _check(self, rc):
        if rc != 0:
            raise MrtError(rc, lib().mrt_accel_last_error(self._h).decode())

    def set_scene(self, scene):
        st = scene.struct
        self._check(lib().mrt_accel_set_scene(self._h, C.byref(st)))
        self.xres, self.yres = st.xres, st.yres

    def set_camera(self, scene):
        st = scene.struct
        cam = _camera_of(st)
        self._check(lib().mrt_accel_set_camera(self._h, C.byref(cam)))
        self.xres, self.yres = st.xres, st.yres

    def render(self, want_ids=False, out=None):
        stats = AccelStats()
        img = np.empty((self.yres, self.xres, 4), np.uint8) if out is None else out
        if img.shape != (self.yres, self.xres, 4) or img.dtype != np.uint8 or not img.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        if not want_ids:
            self._check(lib().mrt_accel_render_rgba8(self._h, img.ctypes.data, C.byref(stats)))
            return img, stats
        ids = np.empty((self.yres, self.xres), np.int32)
        t = np.empty((self.yres, self.xres), np.float32)
        self._check(lib().mrt_accel_render_debug(self._h, img.ctypes.data, ids.ctypes.data, t.ctypes.data,
                                                 C.byref(stats)))
        return img, ids, t, stats

    def render_bands_device(self, band_rows_, rank, world, dev_ptr, nbytes, stream=0, timed=False):
        stats = AccelStats() if timed else None
        self._check(lib().mrt_accel_render_bands_device(self._h, band_rows_, rank, world, dev_ptr, nbytes,
                                                        stream or None, C.byref(stats) if timed else None))
        return stats

    def close(self):
        if self._h:
            lib().mrt_accel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def renderer(device=-1, kind=0, flags=FLAG_NONE):
    """A brute-force Context for an integer flag set, an Accel for a method name ("grid", "cull",
    "block"): the two families render the same frames and share their Python methods."""
    if isinstance(kind, str):
        return Accel(device=device, method=kind, flags=flags)
    return Context(device=device, flags=int(kind) | flags)


class MultiContext:
    """mrt_multi: one frame over several GPUs (or several ranks sharing GPUs) from this process."""

    def __init__(self, devices=None, n_gpus=0, flags=FLAG_NONE, band_rows_=0, gather=GATHER_AUTO):
        L = lib()
        self._h = C.c_void_p()
        cfg = MultiConfigStruct()
        if devices is not None:
            self._devs = (C.c_int32 * len(devices))(*devices)
            cfg.n_gpus, cfg.devices = len(devices), self._devs
        else:
            cfg.n_gpus, cfg.devices = n_gpus, None
        cfg.flags, cfg.band_rows, cfg.gather = flags, band_rows_, gather
        rc = L.mrt_multi_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MrtError(rc, L.mrt_multi_last_error(None).decode())
        self.world = L.mrt_multi_world(self._h)
        self.gather = L.mrt_multi_gather(self._h)
        self.note = L.mrt_multi_note(self._h).decode()
        self.xres = self.yres = 0

    def _check(self, rc):
        if rc != 0:
            raise MrtError(rc, lib().mrt_multi_last_error(self._h).decode())

    def set_scene(self, scene):
        st = scene.struct
        self._check(lib().mrt_multi_set_scene(self._h, C.byref(st)))
        self.xres, self.yres = st.xres, st.yres

    def set_camera(self, scene):
        st = scene.struct
        cam = CameraStruct()
        for name in ("eye", "u", "v", "w", "xres", "yres", "left", "right", "top", "bottom", "projection", "focal"):
            setattr(cam, name, getattr(st, name))
        self._check(lib().mrt_multi_set_camera(self._h, C.byref(cam)))
        self.xres, self.yres = st.xres, st.yres

    def render(self, out=None):
        stats = MultiStats()
        img = np.empty((self.yres, self.xres, 4), np.uint8) if out is None else out
        if img.shape != (self.yres, self.xres, 4) or img.dtype != np.uint8 or not img.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        self._check(lib().mrt_multi_render_rgba8(self._h, img.ctypes.data, C.byref(stats)))
        return img, stats

    def render_rows(self, out, on_rows=None):
        """mrt_multi_render_rgba8_rows: on_rows(row_begin, row_end) per band as it lands in `out`."""
        stats = MultiStats()
        if out.shape != (self.yres, self.xres, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        cb = ROWS_READY_FN((lambda user, a, b: on_rows(a, b)) if on_rows else (lambda user, a, b: None))
        self._check(lib().mrt_multi_render_rgba8_rows(self._h, out.ctypes.data, cb, None, C.byref(stats)))
        return stats

    def close(self):
        if self._h:
            lib().mrt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Sequence:
    """mrt_sequence: many frames of one scene on one GPU, in_flight of them enqueued side by side."""

    def __init__(self, device=-1, flags=FLAG_NONE, in_flight=0):
        L = lib()
        self._h = C.c_void_p()
        cfg = SequenceConfigStruct(device, flags, in_flight)
        rc = L.mrt_sequence_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MrtError(rc, L.mrt_sequence_last_error(None).decode())
        self.in_flight = L.mrt_sequence_in_flight(self._h)
        self.xres = self.yres = 0

    def _check(self, rc):
        if rc != 0:
            raise MrtError(rc, lib().mrt_sequence_last_error(self._h).decode())

    def set_scene(self, scene):
        st = scene.struct
        self._check(lib().mrt_sequence_set_scene(self._h, C.byref(st)))
        self.xres, self.yres = st.xres, st.yres

    def submit(self, band_rows_, rank, world, dev_ptr, nbytes, view=None):
        """Enqueues the next frame (view: a Scene whose camera / viewport to take, None = as the
        frame before) and returns the hipStream_t it runs on, as an integer."""
        cam = None
        if view is not None:
            cam = _camera_of(view.struct)
        stream = C.c_void_p()
        self._check(lib().mrt_sequence_submit(self._h, C.byref(cam) if cam is not None else None, band_rows_, rank,
                                              world, dev_ptr, nbytes, C.byref(stream)))
        if view is not None:
            self.xres, self.yres = view.struct.xres, view.struct.yres
        return stream.value or 0

    def submit_host(self, out, view=None):
        """Enqueues the next frame into `out`, a C-contiguous uint8 array of shape (yres, xres, 4) of the
        frame's view (best from host_array()); complete after wait()."""
        cam = _camera_of(view.struct) if view is not None else None
        xres, yres = (view.struct.xres, view.struct.yres) if view is not None else (self.xres, self.yres)
        if out.shape != (yres, xres, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        self._check(lib().mrt_sequence_submit_host(self._h, C.byref(cam) if cam is not None else None,
                                                   out.ctypes.data))
        self.xres, self.yres = xres, yres

    def wait(self):
        self._check(lib().mrt_sequence_wait(self._h))

    def close(self):
        if self._h:
            lib().mrt_sequence_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """Owns one mrt_host_alloc block."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        rc = lib().mrt_host_alloc(nbytes, C.byref(p))
        if rc != 0:
            raise MrtError(rc, "mrt_host_alloc failed")
        self.ptr = p.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                lib().mrt_host_free(self.ptr)
                self.ptr = None
        except Exception:      # interpreter shutdown: the module's globals may be gone already
            pass


def host_array(shape, dtype=np.uint8):
    """A numpy array in page-locked host memory (mrt_host_alloc).  The block is freed when the
    last array viewing it is gone: numpy holds the ctypes buffer, the buffer holds the block."""
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    block = _PinnedBlock(nbytes)
    buf = (C.c_uint8 * nbytes).from_address(block.ptr)
    buf._owner = block
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def band_rows(yres, rows_per_band, rank, world):
    return int(lib().mrt_band_rows(yres, rows_per_band, rank, world))


def balanced_band_rows(yres, world):
    """mrt_balanced_band_rows: the band height (64 or 32 rows) that balances the ranks best."""
    return int(lib().mrt_balanced_band_rows(yres, world))


def bands_to_image(gathered: np.ndarray, stride_bytes, xres, yres, rows_per_band, world):
    out = np.empty((yres, xres, 4), np.uint8)
    g = np.ascontiguousarray(gathered, dtype=np.uint8)
    rc = lib().mrt_bands_to_image(g.ctypes.data, stride_bytes, xres, yres, rows_per_band, world,
                                  out.ctypes.data)
    if rc != 0:
        raise MrtError(rc, "mrt_bands_to_image")
    return out


def write_png_rgb(path, rgba: np.ndarray):
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    rc = lib().mrt_write_png_rgb(str(path).encode(), a.ctypes.data, a.shape[1], a.shape[0])
    if rc != 0:
        raise MrtError(rc, "mrt_write_png_rgb")


class PngStream:
    """mrt_png_stream: a PNG encoded while its rows arrive in `rgba` (kept alive by this object)."""

    def __init__(self, path, rgba: np.ndarray):
        if rgba.dtype != np.uint8 or rgba.ndim != 3 or rgba.shape[2] != 4 or not rgba.flags.c_contiguous:
            raise ValueError("rgba must be a C-contiguous uint8 array of shape (yres, xres, 4)")
        self._rgba = rgba
        self._h = C.c_void_p()
        rc = lib().mrt_png_stream_begin(str(path).encode(), rgba.ctypes.data, rgba.shape[1], rgba.shape[0], C.byref(self._h))
        if rc != 0:
            raise MrtError(rc, "mrt_png_stream_begin")

    def rows(self, row_begin, row_end):
        rc = lib().mrt_png_stream_rows(self._h, row_begin, row_end)
        if rc != 0:
            raise MrtError(rc, "mrt_png_stream_rows")

    def end(self):
        h, self._h = self._h, None
        rc = lib().mrt_png_stream_end(h)
        if rc != 0:
            raise MrtError(rc, "mrt_png_stream_end")
