"""MI355X-native render hot path: Python view of the C ABI in include/mrt.h.

The product is the shared library ``libmrt_hip.so`` (HIP kernels + C ABI) and
the ``mrraytracer`` command-line driver built from ``csrc/`` and ``host/``;
this package only binds the ABI with ctypes for tests and benchmarks.
"""
from .capi import (  # noqa: F401
    MrtError, Context, MultiContext, MultiStats, Sequence, LoadedScene, Scene, Stats, lib, lib_path, band_rows, balanced_band_rows, bands_to_image,
    write_png_rgb, PngStream, host_array, FLAG_NONE, FLAG_NO_PREFILTER, FLAG_COUNT_EXACT, FLAG_NO_CIRCLE,
    FLAG_FULL_TEST, FLAG_TRACE_NARROW, FLAG_TRACE_WIDE, FLAG_DEFER_NOALLOC_SETUP, RESERVED_FLAG_BITS, PROJ_ORTHO, PROJ_PERSP,
    SHADER_FLAT, SHADER_PHONG, CL_DTYPES, SceneStruct, EXPORTED_SYMBOLS, ABI_VERSION, NOALLOC_MAX_KEYS,
    KERNEL_TRACE, KERNEL_TRACE_BLOCK, KERNEL_CULL, KERNEL_FULL, KERNEL_GRID, KERNEL_TRACE_WIDE,
    Accel, AccelStats, renderer, ACCEL_GRID, ACCEL_TILE_CULL, ACCEL_BLOCK_REJECT, ACCEL_METHODS,
    GATHER_AUTO, GATHER_RCCL, GATHER_PEER, GATHER_HOST,
)
