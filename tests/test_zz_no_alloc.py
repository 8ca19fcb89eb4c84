"""mrt_render_bands_device_noalloc: the entry point that makes no device allocation.

Called through the C ABI and checked against mrt_render_bands_device (the entry point it is a
variant of) byte for byte -- the bar for byte work -- on two calls in a row, the second with a
smaller frame, so that whatever the first call left in the fixed key array must not leak into the
second.  Also: the size limit and the kernels it refuses, and free device memory before and
after a call.
"""
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, str(ROOT / "tools"))
import mkscene  # noqa: E402

pytestmark = pytest.mark.gpu


def _both(mrt, ctx, sc, band, rank, world):
    """This rank's bands through the allocating entry point and through the no-alloc one."""
    import torch
    st = sc.struct
    ctx.set_scene(sc)
    rows = mrt.band_rows(st.yres, band, rank, world)
    nbytes = rows * st.xres * 4
    ref = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
    out = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    ctx.render_bands_device(band, rank, world, ref.data_ptr(), nbytes, stream=stream)
    ctx.render_bands_device_noalloc(band, rank, world, out.data_ptr(), nbytes, stream=stream)
    torch.cuda.synchronize()
    return ref.cpu().numpy(), out.cpu().numpy()


def test_noalloc_matches_allocating_entry_point_twice(mrt):
    ctx = mrt.Context(device=0)
    try:
        # first call: the 4K benchmark frame, the largest the fixed array takes in one piece
        big = mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy()
        ref, out = _both(mrt, ctx, big, 2160, 0, 1)
        assert (ref.reshape(-1, 4)[:, 3] == 255).all() and len(np.unique(ref.reshape(-1, 4), axis=0)) > 100
        assert np.array_equal(out, ref)
        # second call, smaller input, other pitch, other camera: the array still holds the keys of
        # the first frame wherever this frame does not clear them
        small = mrt.LoadedScene(text=mkscene.soup(333, 222, n_tris=700, n_spheres=60, seed=5,
                                                  projection="ortho")).copy()
        ref2, out2 = _both(mrt, ctx, small, 222, 0, 1)
        assert len(np.unique(ref2.reshape(-1, 4), axis=0)) > 20
        assert np.array_equal(out2, ref2)
        # interleaved bands of another rank
        for band, rank, world in ((64, 1, 2), (7, 2, 3)):
            ref3, out3 = _both(mrt, ctx, big.with_resolution(1280, 720), band, rank, world)
            assert np.array_equal(out3, ref3)
        # fewer than 256 triangles: one slice, the kernel's own prologue clears the keys it uses
        tiny = mrt.LoadedScene(text=mkscene.soup(300, 200, n_tris=200, n_spheres=20, seed=8)).copy()
        ref4, out4 = _both(mrt, ctx, tiny, 200, 0, 1)
        assert len(np.unique(ref4.reshape(-1, 4), axis=0)) > 20
        assert np.array_equal(out4, ref4)
    finally:
        ctx.close()


def test_noalloc_other_kernels_and_limits(mrt):
    import torch
    sc = mrt.LoadedScene(text=mkscene.teatime(640, 360)).copy()
    # the kernels that work in the key plane (and the one that needs no plane at all) are served
    for flags in (mrt.FLAG_NO_CIRCLE, mrt.FLAG_FULL_TEST, mrt.FLAG_COUNT_EXACT):
        ctx = mrt.Context(device=0, flags=flags)
        try:
            ref, out = _both(mrt, ctx, sc, 360, 0, 1)
            assert np.array_equal(out, ref), flags
        finally:
            ctx.close()
    # the full-test kernel merges its layers in the key plane when the frame is small; a frame
    # beyond the fixed array runs as one layer, which needs no keys at all
    ctx = mrt.Context(device=0, flags=mrt.FLAG_FULL_TEST)
    try:
        ref, out = _both(mrt, ctx, sc.with_resolution(4000, 2300), 2300, 0, 1)
        assert np.array_equal(out, ref)
    finally:
        ctx.close()
    # beyond the fixed array: 7680 x 4320 needs 33.4 M keys, the array holds 8 M
    ctx = mrt.Context(device=0)
    try:
        ctx.set_scene(sc.with_resolution(7680, 4320))
        buf = torch.zeros(7680 * 4320 * 4, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(mrt.MrtError) as e:
            ctx.render_bands_device_noalloc(4320, 0, 1, buf.data_ptr(), buf.numel())
        assert e.value.code == -5 and "keys" in e.value.message
        # ... but an eighth of it (one rank of eight) fits
        rows = mrt.band_rows(4320, 64, 3, 8)
        ctx.render_bands_device_noalloc(64, 3, 8, buf.data_ptr(), rows * 7680 * 4)
        torch.cuda.synchronize()
        # argument checks are those of the allocating entry point
        with pytest.raises(mrt.MrtError):
            ctx.render_bands_device_noalloc(64, 3, 8, buf.data_ptr(), 16)
        with pytest.raises(mrt.MrtError):
            ctx.render_bands_device_noalloc(0, 0, 1, buf.data_ptr(), buf.numel())
    finally:
        ctx.close()


_DEFERRED_SNIPPET = r"""
import sys
sys.path[:0] = [{root!r}, {root!r} + "/tools"]
import numpy as np
import torch
import mrt_pkg, mkscene
mrt = mrt_pkg.load()
ctx = mrt.Context(device=0, flags=mrt.FLAG_DEFER_NOALLOC_SETUP)      # the only context of this process
ok = True
for xres, yres, band, rank, world in ((1280, 720, 720, 0, 1), (640, 360, 64, 1, 2)):
    sc = mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy()
    ctx.set_scene(sc)
    n = mrt.band_rows(yres, band, rank, world) * xres * 4
    ref = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    out = torch.full((n,), 0x5A, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    ctx.render_bands_device(band, rank, world, ref.data_ptr(), n, stream=s)
    ctx.render_bands_device_noalloc(band, rank, world, out.data_ptr(), n, stream=s)
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(ref, out)) and len(np.unique(ref.cpu().numpy().reshape(-1, 4), axis=0)) > 100
ctx.close()
print("deferred set-up:", "same bytes" if ok else "DIFFERENT")
"""


def test_noalloc_on_a_context_that_deferred_the_set_up():
    """MRT_FLAG_DEFER_NOALLOC_SETUP: mrt_create leaves the fixed array alone (what the CLI asks for).
    In a process of its own, whose only context has the flag, the entry point's first call takes
    the array and writes the bytes of the allocating entry point; a second, smaller call too."""
    import subprocess
    r = subprocess.run([sys.executable, "-c", _DEFERRED_SNIPPET.format(root=str(ROOT))], capture_output=True, text=True,
                       timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "deferred set-up: same bytes" in r.stdout, r.stdout[-500:]


def test_noalloc_leaves_free_memory_alone(mrt):
    """Free device memory as the driver reports it does not shrink across calls (after one call
    of each entry point has loaded whatever the runtime loads lazily)."""
    import torch
    sc = mrt.LoadedScene(text=mkscene.teatime(1920, 1080)).copy()
    ctx = mrt.Context(device=0)
    try:
        _both(mrt, ctx, sc, 1080, 0, 1)
        buf = torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        for _ in range(5):
            ctx.render_bands_device_noalloc(1080, 0, 1, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info(0)
        assert free1 >= free0, (free0, free1)
    finally:
        ctx.close()


def _hip():
    """The HIP runtime the process already runs on (torch's copy, which libmrt_hip.so binds to)."""
    import ctypes as C
    import os
    import torch
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    vp = C.c_void_p
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    return hip


def test_capturing_streams_are_refused(mrt):
    """HIP graph capture is not supported: on a stream that is recording a graph every entry point
    that takes a stream fails with MRT_ERR_INVALID before it enqueues anything, the capture stays
    valid and comes out empty, and the context goes on rendering afterwards.  (Nothing recorded is
    ever launched here.)"""
    import ctypes as C
    import torch
    hip = _hip()
    sc = mrt.LoadedScene(text=mkscene.teatime(640, 360)).copy()
    nbytes = 640 * 360 * 4
    stream = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0            # non-blocking
    ctx = mrt.Context(device=0)
    grid = mrt.Accel(device=0, method="grid")
    try:
        ctx.set_scene(sc)
        grid.set_scene(sc)
        ref = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
        out = torch.full((nbytes,), 0x5A, dtype=torch.uint8, device="cuda:0")
        ctx.render_bands_device(360, 0, 1, ref.data_ptr(), nbytes, stream=stream.value)
        assert hip.hipStreamSynchronize(stream) == 0
        ctx.synchronize()
        graph = C.c_void_p()
        assert hip.hipStreamBeginCapture(stream, 1) == 0                    # thread-local mode
        for call in (lambda: ctx.render_bands_device_noalloc(360, 0, 1, out.data_ptr(), nbytes, stream=stream.value),
                     lambda: ctx.render_bands_device(360, 0, 1, out.data_ptr(), nbytes, stream=stream.value),
                     lambda: grid.render_bands_device(360, 0, 1, out.data_ptr(), nbytes, stream=stream.value)):
            with pytest.raises(mrt.MrtError) as e:
                call()
            assert e.value.code == -1 and "captur" in e.value.message
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0          # the refusals left it valid
        if graph.value:
            assert hip.hipGraphDestroy(graph) == 0
        torch.cuda.synchronize()
        assert (out.cpu().numpy() == 0x5A).all()                            # nothing ran, nothing was recorded
        ctx.render_bands_device_noalloc(360, 0, 1, out.data_ptr(), nbytes, stream=stream.value)
        assert hip.hipStreamSynchronize(stream) == 0
        assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy())
    finally:
        hip.hipStreamSynchronize(stream)
        ctx.close()
        grid.close()
        hip.hipStreamDestroy(stream)
