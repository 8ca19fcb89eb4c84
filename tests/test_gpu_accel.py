"""The accelerated entry points (mrt_accel_*) against the brute-force context and the golden
fixtures: a frame rendered through the screen-space grid, the tile cull or the block reject must
carry the same winners, the same t (bit for bit) and the same bytes as the frame that tested every
pixel against every object.  The reference has no such path (its loop is kernel.cl:320-360); the
bar is therefore identity with the brute-force path, which the other GPU tests hold to the oracle.
"""
import sys

import numpy as np
import pytest

from conftest import SCENES, GOLDEN, ALL_SCENES, ROOT
from parity import check

sys.path.insert(0, str(ROOT / "tools"))
import mkscene  # noqa: E402
import scene_py  # noqa: E402

pytestmark = pytest.mark.gpu

METHODS = ("grid", "cull", "block")


def frame(r, scene=None):
    if scene is not None:
        r.set_scene(scene)
    img, ids, t, st = r.render(want_ids=True)
    return img, ids, t, st


def assert_identical(a, b, what=""):
    assert np.array_equal(a[1], b[1]), (what, int((a[1] != b[1]).sum()))
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)), what
    assert np.array_equal(a[0], b[0]), what


@pytest.mark.parametrize("name", ALL_SCENES)
def test_accel_matches_golden(mrt, name):
    g = np.load(GOLDEN / f"{name}.npz")
    ls = mrt.LoadedScene(path=SCENES / f"{name}.json")
    for method in METHODS:
        a = mrt.Accel(device=0, method=method)
        try:
            img, ids, t, st = frame(a, ls)
        finally:
            a.close()
        assert np.array_equal(ids, g["ids"]) and np.array_equal(t, g["t"]), method
        check(img, g["rgba"], g["ids"])
        if method == "grid":
            assert st.kernel == mrt.KERNEL_GRID and st.cells == 13 * 13 and st.refs > 0 and st.build_ms > 0


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (31, 33), (33, 65), (100, 37), (257, 129), (640, 360)])
def test_grid_ragged_resolutions(gpu_ctx, mrt, w, h):
    """Image sizes that are not multiples of the 32x32 cell, of the 64x64 block or of 4 pixels."""
    grid = mrt.Accel(device=0, method="grid")
    try:
        for name in ("scene_2spheres_persp_phong", "scene_gtri_ortho_phong", "teatime"):
            sc = scene_py.load_scene(SCENES / f"{name}.json", xres=w, yres=h)
            assert_identical(frame(grid, sc), frame(gpu_ctx, sc), (name, w, h))
    finally:
        grid.close()


def _view_scene(mrt, base, view):
    st = view.struct
    return mrt.Scene(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=st.xres, yres=st.yres,
                     left=st.left, right=st.right, top=st.top, bottom=st.bottom, projection=st.projection,
                     focal=st.focal, shader=base.struct.shader, ka=base.struct.ka, kd=base.struct.kd,
                     ks=base.struct.ks, ambient=list(base.struct.ambient), background=list(base.struct.background),
                     sph=base.sph, sph_mat=base.sph_mat, tri=base.tri, tri_mat=base.tri_mat, mat=base.mat,
                     light=base.light)


def test_grid_follows_the_camera(gpu_ctx, mrt):
    """mrt_accel_set_camera rebuilds the grid: a sequence of views (both projections, changing
    resolution, so the number of cells and of references changes too) through one handle equals
    the brute-force frame of each view; the list buffer grows and is reused."""
    from test_gpu_parity import _random_scene
    rng = np.random.default_rng(2024)
    base = _random_scene(mrt, rng, mrt.PROJ_PERSP, 200, 150, 600, 60)
    grid = mrt.Accel(device=0, method="grid")
    try:
        grid.set_scene(base)
        refs = set()
        for k in range(8):
            view = _random_scene(mrt, rng, mrt.PROJ_PERSP if k % 2 else mrt.PROJ_ORTHO, 90 + 61 * k, 70 + 43 * k, 1, 1)
            sc = _view_scene(mrt, base, view)
            grid.set_camera(sc)
            a = frame(grid)
            assert_identical(a, frame(gpu_ctx, sc), k)
            assert a[3].cells == ((sc.struct.xres + 31) // 32) * ((sc.struct.yres + 31) // 32)
            refs.add(int(a[3].refs))
        assert len(refs) >= 3 and max(refs) > 1000
        # frames on a stream that is not the handle's own wait for the build
        import torch
        side = torch.cuda.Stream()
        view = _view_scene(mrt, base, _random_scene(mrt, rng, mrt.PROJ_PERSP, 160, 128, 1, 1))
        buf = torch.zeros(160 * 128 * 4, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        grid.set_camera(view)
        grid.render_bands_device(128, 0, 1, buf.data_ptr(), buf.numel(), stream=side.cuda_stream)
        torch.cuda.synchronize()
        gpu_ctx.set_scene(view)
        want, _ = gpu_ctx.render()
        assert np.array_equal(buf.cpu().numpy().reshape(128, 160, 4), want)
    finally:
        grid.close()


def test_grid_with_primitives_of_every_size(gpu_ctx, mrt):
    """Triangles that cover the whole image (in every cell's list), needles that cross it (a large
    bounding box, few cells), subpixel ones, geometry behind and around the eye (records every
    tile passes), NaN and infinite vertices, spheres around the eye: the lists may only ever hold
    too much, never too little."""
    nan, inf = float("nan"), float("inf")
    rng = np.random.default_rng(99)
    small = rng.uniform(-1, 1, (200, 9)) * 0.05 + np.repeat(rng.uniform(-1.5, 1.5, (200, 3)), 3, axis=0).reshape(200, 9)
    small[:, 2::3] = -rng.uniform(2, 6, (200, 1))
    special = np.array([
        [-100, -100, -50, 100, -100, -50, 0, 100, -50],      # covers everything, far away
        [-3, -3, -8, 3, 3, -8, 3.001, 3, -8],                # needle across the image
        [-3, 2.5, -7, 3, -2.5, -7, 3, -2.499, -7],           # another, the other way
        [0.1, 0.1, -3, 0.1001, 0.1, -3, 0.1, 0.1001, -3],    # far smaller than a pixel
        [-1, -1, 2, 1, -1, -3, 0, 1, -3],                    # crosses the eye plane
        [-5, -5, 1, 5, -5, 1, 0, 5, 1],                      # behind the camera
        [nan, 0, -2, 1, 0, -2, 0, 1, -2],
        [-1, -1, -2.5, inf, 0, -2.5, 0, 1, -2.5],
        [-1e19, -1e19, -4, 1e19, -1e19, -4, 0, 1e19, -4],
    ], dtype=np.float64)
    tri = np.concatenate([special, small]).T
    sph = np.array([[0, 0, 0, 2.0], [0.3, 0.3, -4, 0.5], [0, 0, 3, 1.0], [2, 0, -0.2, 1.0], [0, 0, -30, 1e-3],
                    [-0.8, 0.5, -2, -0.4], [1.5, -1.0, -5, 0.7]], dtype=np.float64).T
    n_tri, n_sph = tri.shape[1], sph.shape[1]
    grid = mrt.Accel(device=0, method="grid")
    cull = mrt.Accel(device=0, method="cull")
    try:
        for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
            for (w, h) in ((640, 400), (97, 61)):
                sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, 1, 0], w=[0, 0, 1], xres=w, yres=h, left=-1.6,
                               right=1.6, top=1.0, bottom=-1.0, projection=proj, focal=1.0, shader=mrt.SHADER_PHONG,
                               ka=0.2, kd=0.6, ks=0.3, ambient=(1, 1, 1), background=(0.1, 0.2, 0.3), sph=sph,
                               sph_mat=np.arange(n_sph) % 2, tri=tri, tri_mat=np.arange(n_tri) % 2,
                               mat=[[1.0, 0.2], [0.5, 0.9], [0.25, 0.4], [4.0, 16.0]],
                               light=[[2.0], [3.0], [1.0], [1.0], [1.0], [1.0]])
                want = frame(gpu_ctx, sc)
                assert len(np.unique(want[1])) > (20 if w > 100 else 5)
                got = frame(grid, sc)
                assert_identical(got, want, (proj, w, h))
                assert_identical(frame(cull, sc), want, (proj, w, h))
                cells = got[3].cells
                assert got[3].refs >= cells            # the covering triangle alone is in every list
    finally:
        grid.close()
        cull.close()


def test_grid_empty_scene_and_stats(mrt):
    sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=70, yres=50, left=-1, right=1,
                   top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0, shader=mrt.SHADER_PHONG, ka=0.3,
                   kd=0.5, ks=0.2, ambient=(1, 0.5, 0.25), background=(0.25, 0.5, 0.75))
    grid = mrt.Accel(device=0, method="grid", flags=mrt.FLAG_COUNT_EXACT)
    try:
        img, ids, t, st = frame(grid, sc)
        assert (ids == -1).all() and (t == -1).all() and (img == [63, 127, 191, 255]).all()
        assert st.refs == 0 and st.cells == 3 * 2 and st.exact_tests == 0 and st.rays == 70 * 50
        # one sphere in the middle: only the cells it touches list it
        one = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=256, yres=256, left=-1, right=1,
                        top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0, shader=mrt.SHADER_FLAT,
                        background=(0, 0, 0), sph=[[0], [0], [4], [0.5]], sph_mat=[0], mat=[[1], [1], [1], [4]])
        img, ids, t, st = frame(grid, one)
        assert st.cells == 64 and 4 <= st.refs <= 16 and (ids == 0).sum() > 500
        assert 0 < st.exact_tests < 256 * 256 // 4
    finally:
        grid.close()


def test_accel_api_errors(mrt):
    # the culling kernels are not flags of the brute-force context any more
    for bits in (1 << 2, 1 << 4, mrt.RESERVED_FLAG_BITS):
        with pytest.raises(mrt.MrtError) as e:
            mrt.Context(device=0, flags=bits)
        assert e.value.code == -1 and "mrt_accel_create" in e.value.message
    with pytest.raises(mrt.MrtError):
        mrt.Accel(device=0, method=7)
    with pytest.raises(mrt.MrtError):
        mrt.Accel(device=0, method="grid", flags=mrt.FLAG_FULL_TEST)
    with pytest.raises(mrt.MrtError):
        mrt.Accel(device=99)
    for method in METHODS:
        a = mrt.Accel(device=0, method=method)
        try:
            a.xres = a.yres = 4
            with pytest.raises(mrt.MrtError) as e:
                a.render()
            assert e.value.code == -4            # MRT_ERR_NO_SCENE
            with pytest.raises(mrt.MrtError) as e:
                a.set_camera(mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=8, yres=8,
                                       left=-1, right=1, top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0))
            assert e.value.code == -4
            ls = mrt.LoadedScene(path=SCENES / "scene_gtri_persp_flat.json")
            a.set_scene(ls)
            import torch
            buf = torch.zeros(400 * 400 * 4, dtype=torch.uint8, device="cuda:0")
            with pytest.raises(mrt.MrtError):
                a.render_bands_device(400, 0, 1, buf.data_ptr(), 16)           # too small
            with pytest.raises(mrt.MrtError):
                a.render_bands_device(0, 0, 1, buf.data_ptr(), buf.numel())     # bad partition
            a.render_bands_device(400, 0, 1, buf.data_ptr(), buf.numel())
            torch.cuda.synchronize()
            want, _ = a.render()
            assert np.array_equal(buf.cpu().numpy().reshape(400, 400, 4), want)
            # a camera that is refused outright leaves the handle as it was
            bad = ls.copy()
            bad.struct.projection = 9
            with pytest.raises(mrt.MrtError) as e:
                a.set_camera(bad)
            assert e.value.code == -1
            a.xres, a.yres = 400, 400
            again, _ = a.render()
            assert np.array_equal(again, want)
        finally:
            a.close()


def test_grid_teatime_8k_and_bands(mrt):
    """teatime at 7680 x 4320 through the grid: identical to the brute-force frame, whole and cut
    into interleaved 64-row bands for 2 and 8 ranks."""
    import torch
    xres, yres, band = 7680, 4320, 64
    sc = mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy()
    ctx = mrt.Context(device=0)
    grid = mrt.Accel(device=0, method="grid")
    try:
        want = frame(ctx, sc)
        got = frame(grid, sc)
        assert_identical(got, want)
        assert got[3].cells == 240 * 135
        full = want[0]
        for world in (2, 8):
            rows = [mrt.band_rows(yres, band, r, world) for r in range(world)]
            stride = max(rows) * xres * 4
            gathered = torch.zeros((world, stride), dtype=torch.uint8, device="cuda:0")
            for r in range(world):
                grid.render_bands_device(band, r, world, gathered[r].data_ptr(), stride,
                                         stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            img = mrt.bands_to_image(gathered.cpu().numpy(), stride, xres, yres, band, world)
            assert np.array_equal(img, full), world
            del gathered, img
    finally:
        ctx.close()
        grid.close()
