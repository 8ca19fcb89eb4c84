"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle and the committed golden fixtures.

Tolerance: <= 1 LSB per 8-bit channel on >= 99.9 % of
pixels, the rest confined to primitive / silhouette edges (tests/parity.py).
Hit/miss decisions, winners and t are integer / decision work and are held to
the stricter bar: identical to the oracle's, bit for bit, because the device
runs the same FP32 operations in the same order for every accepted pair.  The
only arithmetic allowed to differ is powf() inside Blinn-Phong (device libm vs
glibc), which is why colour bytes get the 1-LSB tolerance.
"""
import subprocess
import sys

import numpy as np
import pytest

from conftest import SCENES, GOLDEN, ALL_SCENES, ROOT, free_port
from parity import check

sys.path.insert(0, str(ROOT / "tools"))
import mkscene  # noqa: E402
import scene_py  # noqa: E402

pytestmark = pytest.mark.gpu


def gpu_render(ctx, scene):
    ctx.set_scene(scene)
    img, ids, t, st = ctx.render(want_ids=True)
    return {"rgba": img, "ids": ids, "t": t, "stats": st}


def gpu_render_flags(mrt, scene, flags):
    ctx = mrt.Context(device=0, flags=flags)
    try:
        return gpu_render(ctx, scene)
    finally:
        ctx.close()


def assert_same_hits(g, o):
    assert np.array_equal(g["ids"], o["ids"]), int((g["ids"] != o["ids"]).sum())
    assert np.array_equal(g["t"], o["t"])


@pytest.mark.parametrize("name", ALL_SCENES)
def test_native_scenes_match_golden(gpu_ctx, mrt, name):
    g = np.load(GOLDEN / f"{name}.npz")
    ls = mrt.LoadedScene(path=SCENES / f"{name}.json")
    r = gpu_render(gpu_ctx, ls)
    assert_same_hits(r, g)
    stats = check(r["rgba"], g["rgba"], g["ids"])
    if "flat" in name:
        assert stats["exact"] == stats["pixels"]


@pytest.mark.parametrize("name", ALL_SCENES)
def test_native_scenes_wide_lane_blocks_match_golden(mrt, name):
    """The 8x4-pixels-per-lane shape of the default kernel (what frames of 3 Mpixel and more get),
    forced onto the 400x400 scenes: 64x32 wave tiles, 128x64 block tiles, 64-bit pixel masks."""
    g = np.load(GOLDEN / f"{name}.npz")
    ctx = mrt.Context(device=0, flags=mrt.FLAG_TRACE_WIDE)
    try:
        r = gpu_render(ctx, mrt.LoadedScene(path=SCENES / f"{name}.json"))
    finally:
        ctx.close()
    assert r["stats"].kernel == mrt.KERNEL_TRACE_WIDE
    assert_same_hits(r, g)
    check(r["rgba"], g["rgba"], g["ids"])


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (129, 65), (257, 33), (640, 360)])
def test_ragged_resolutions_wide_lane_blocks(mrt, oracle, w, h):
    sc = scene_py.load_scene(SCENES / "teatime.json", xres=w, yres=h)
    o = oracle.render(sc, threads=8)
    ctx = mrt.Context(device=0, flags=mrt.FLAG_TRACE_WIDE)
    try:
        r = gpu_render(ctx, sc)
    finally:
        ctx.close()
    assert r["stats"].kernel == mrt.KERNEL_TRACE_WIDE
    assert_same_hits(r, o)
    check(r["rgba"], o["rgba"], o["ids"], frac=0.0 if w * h < 2000 else 0.999)


@pytest.mark.parametrize("name", ["scene_2spheres_persp_phong", "scene_gtri_persp_phong", "teatime"])
def test_native_scenes_match_oracle_run_here(gpu_ctx, mrt, oracle, name):
    sc = scene_py.load_scene(SCENES / f"{name}.json")
    o = oracle.render(sc, threads=8)
    r = gpu_render(gpu_ctx, sc)
    assert_same_hits(r, o)
    check(r["rgba"], o["rgba"], o["ids"])


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (33, 65), (100, 37), (257, 129), (640, 360)])
def test_ragged_resolutions(gpu_ctx, oracle, w, h):
    # sizes that are not multiples of the 32x32 wave tile or of 4 pixels
    for name in ("scene_2spheres_persp_phong", "teatime"):
        sc = scene_py.load_scene(SCENES / f"{name}.json", xres=w, yres=h)
        o = oracle.render(sc, threads=8)
        r = gpu_render(gpu_ctx, sc)
        assert_same_hits(r, o)
        check(r["rgba"], o["rgba"], o["ids"], frac=0.0 if w * h < 2000 else 0.999)


@pytest.mark.parametrize("proj", ["persp", "ortho"])
@pytest.mark.parametrize("seed", [1, 2])
def test_random_soup_matches_oracle(gpu_ctx, mrt, oracle, proj, seed):
    text = mkscene.soup(320, 200, n_tris=1500, n_spheres=120, seed=seed, projection=proj)
    sc = mrt.LoadedScene(text=text).copy()
    o = oracle.render(sc, threads=8)
    assert (o["ids"] >= 0).sum() > 500
    r = gpu_render(gpu_ctx, sc)
    assert_same_hits(r, o)
    check(r["rgba"], o["rgba"], o["ids"])


@pytest.mark.parametrize("shape", ["narrow", "wide"])
def test_prefilter_is_a_superset_of_exact(mrt, oracle, shape):
    """The conservative edge-function prefilter may only drop pairs the exact
    test rejects: with and without it the winners and t must be identical, in both lane shapes
    of the kernel (and both shapes count the same survivors: the tests are per pixel)."""
    shape_flag = mrt.FLAG_TRACE_WIDE if shape == "wide" else mrt.FLAG_TRACE_NARROW
    a = mrt.Context(device=0, flags=mrt.FLAG_COUNT_EXACT | shape_flag)
    b = mrt.Context(device=0, flags=mrt.FLAG_NO_PREFILTER | mrt.FLAG_COUNT_EXACT | shape_flag)
    try:
        cases = [scene_py.load_scene(SCENES / "teatime.json", xres=960, yres=540,
                                     viewport=(-16 / 9, 16 / 9, 1.0, -1.0)),
                 scene_py.load_scene(SCENES / "teatime.json", xres=301, yres=299),
                 mrt.LoadedScene(text=mkscene.soup(512, 512, 4096, 256, seed=7)).copy(),
                 mrt.LoadedScene(text=mkscene.soup(400, 300, 2000, 10, seed=9, projection="ortho")).copy()]
        for sc in cases:
            ra = gpu_render(a, sc)
            rb = gpu_render(b, sc)
            assert_same_hits(ra, rb)
            assert np.array_equal(ra["rgba"], rb["rgba"])
            n_pairs = sc.struct.xres * sc.struct.yres * sc.n_objects
            assert rb["stats"].exact_tests >= sc.struct.xres * sc.struct.yres * sc.struct.n_tris
            assert ra["stats"].exact_tests < 0.05 * n_pairs   # the filter does reject almost everything
            assert ra["stats"].kernel == rb["stats"].kernel == (mrt.KERNEL_TRACE_WIDE if shape == "wide" else mrt.KERNEL_TRACE)
            if shape == "wide":       # the per-pixel funnel passes the same pairs whatever the lane block
                # (up to the rounding of the incremental edge evaluation and the padding pixels of the last tiles)
                narrow = mrt.Context(device=0, flags=mrt.FLAG_COUNT_EXACT | mrt.FLAG_TRACE_NARROW)
                try:
                    rn = gpu_render(narrow, sc)
                finally:
                    narrow.close()
                assert abs(rn["stats"].exact_tests - ra["stats"].exact_tests) <= 64 + ra["stats"].exact_tests // 1000
                rc = gpu_render_flags(mrt, sc, mrt.FLAG_NO_CIRCLE | mrt.FLAG_COUNT_EXACT | shape_flag)
                assert_same_hits(ra, rc)
                assert rc["stats"].exact_tests == ra["stats"].exact_tests
    finally:
        a.close()
        b.close()


def test_block_reject_flag_changes_nothing(mrt, oracle):
    """The accelerated entry points (mrt_accel_*: 4x4 block reject, 32x32 tile cull, screen-space
    grid), MRT_FLAG_NO_CIRCLE (no bounding-circle level) and MRT_FLAG_FULL_TEST must give the same
    winners, t and bytes as the default path, including with band heights that are not multiples of
    4 or 32."""
    import torch
    a = mrt.Context(device=0, flags=mrt.FLAG_COUNT_EXACT)
    b = mrt.Accel(device=0, method="block", flags=mrt.FLAG_COUNT_EXACT)
    c = mrt.Context(device=0, flags=mrt.FLAG_NO_CIRCLE | mrt.FLAG_COUNT_EXACT)
    d = mrt.Accel(device=0, method="cull", flags=mrt.FLAG_COUNT_EXACT)
    e = mrt.Context(device=0, flags=mrt.FLAG_FULL_TEST | mrt.FLAG_COUNT_EXACT)
    f = mrt.Accel(device=0, method="grid", flags=mrt.FLAG_COUNT_EXACT)
    try:
        rng = np.random.default_rng(5)
        cases = [scene_py.load_scene(SCENES / "teatime.json", xres=1280, yres=720,
                                     viewport=(-16 / 9, 16 / 9, 1.0, -1.0)),
                 scene_py.load_scene(SCENES / "scene_gtri_ortho_phong.json"),
                 mrt.LoadedScene(text=mkscene.soup(640, 480, 8192, 512, seed=3)).copy(),
                 _random_scene(mrt, rng, mrt.PROJ_PERSP, 333, 222, 500, 50),
                 _random_scene(mrt, rng, mrt.PROJ_ORTHO, 333, 222, 500, 50, skew=True)]
        for sc in cases:
            ra = gpu_render(a, sc)
            rb = gpu_render(b, sc)
            assert_same_hits(ra, rb)
            assert np.array_equal(ra["rgba"], rb["rgba"])
            assert rb["stats"].exact_tests == ra["stats"].exact_tests     # same survivors reach the exact test
            # a whole frame is consecutive rows whatever its height: the flags get the kernel they ask for
            assert ra["stats"].kernel == mrt.KERNEL_TRACE and rb["stats"].kernel == mrt.KERNEL_TRACE_BLOCK
            rc = gpu_render(c, sc)                                        # without the bounding-circle level
            assert_same_hits(ra, rc)
            assert np.array_equal(ra["rgba"], rc["rgba"])
            assert rc["stats"].exact_tests == ra["stats"].exact_tests
            rd = gpu_render(d, sc)                                        # tile culling
            assert rd["stats"].kernel == mrt.KERNEL_CULL
            assert_same_hits(ra, rd)
            assert np.array_equal(ra["rgba"], rd["rgba"])
            assert rd["stats"].exact_tests <= ra["stats"].exact_tests * 1.01 + 64
            rf = gpu_render(f, sc)                                        # screen-space grid
            assert rf["stats"].kernel == mrt.KERNEL_GRID
            assert_same_hits(ra, rf)
            assert np.array_equal(ra["rgba"], rf["rgba"])
            # same tile tests, same survivors -- except that the grid does not list what only touches
            # the padding rows and columns of the last tiles
            assert 0.99 * rd["stats"].exact_tests <= rf["stats"].exact_tests <= rd["stats"].exact_tests
            n_cells = ((sc.struct.xres + 31) // 32) * ((sc.struct.yres + 31) // 32)
            assert rf["stats"].cells == n_cells and 0 < rf["stats"].refs < n_cells * sc.n_objects
            re_ = gpu_render(e, sc)                                       # determinants for every pair
            assert re_["stats"].kernel == mrt.KERNEL_FULL
            assert_same_hits(ra, re_)
            assert np.array_equal(ra["rgba"], re_["rgba"])
            assert re_["stats"].exact_tests >= sc.struct.xres * sc.struct.yres * sc.n_objects
        # bands of 6 rows: the block path must step aside (a lane's rows are not consecutive)
        sc = cases[0]
        b.set_scene(sc)
        full, _ = b.render()
        rows = [mrt.band_rows(720, 6, r, 2) for r in range(2)]
        stride = max(rows) * 1280 * 4
        g = torch.zeros((2, stride), dtype=torch.uint8, device="cuda:0")
        for r in range(2):
            st6 = b.render_bands_device(6, r, 2, g[r].data_ptr(), stride, timed=True)
            assert st6.kernel == mrt.KERNEL_TRACE          # stepped aside
        torch.cuda.synchronize()
        assert np.array_equal(mrt.bands_to_image(g.cpu().numpy(), stride, 1280, 720, 6, 2), full)
        # the tile cull needs 32 consecutive rows: bands of 32 use it, bands of 6 fall back
        d.set_scene(sc)
        for band in (32, 6):
            rows = [mrt.band_rows(720, band, r, 3) for r in range(3)]
            stride = max(rows) * 1280 * 4
            g = torch.zeros((3, stride), dtype=torch.uint8, device="cuda:0")
            for r in range(3):
                stb = d.render_bands_device(band, r, 3, g[r].data_ptr(), stride, timed=True)
                assert stb.kernel == (mrt.KERNEL_CULL if band == 32 else mrt.KERNEL_TRACE)
            torch.cuda.synchronize()
            assert np.array_equal(mrt.bands_to_image(g.cpu().numpy(), stride, 1280, 720, band, 3), full)
        # the grid needs the same of interleaved bands, and says so instead of falling back
        f.set_scene(sc)
        rows = [mrt.band_rows(720, 32, r, 3) for r in range(3)]
        stride = max(rows) * 1280 * 4
        g = torch.zeros((3, stride), dtype=torch.uint8, device="cuda:0")
        for r in range(3):
            stb = f.render_bands_device(32, r, 3, g[r].data_ptr(), stride, timed=True)
            assert stb.kernel == mrt.KERNEL_GRID
        torch.cuda.synchronize()
        assert np.array_equal(mrt.bands_to_image(g.cpu().numpy(), stride, 1280, 720, 32, 3), full)
        with pytest.raises(mrt.MrtError) as err:
            f.render_bands_device(6, 0, 3, g[0].data_ptr(), stride)
        assert err.value.code == -1 and "32" in err.value.message
        e.set_scene(sc)
        rows = [mrt.band_rows(720, 6, r, 2) for r in range(2)]
        stride = max(rows) * 1280 * 4
        g = torch.zeros((2, stride), dtype=torch.uint8, device="cuda:0")
        for r in range(2):
            e.render_bands_device(6, r, 2, g[r].data_ptr(), stride)
        torch.cuda.synchronize()
        assert np.array_equal(mrt.bands_to_image(g.cpu().numpy(), stride, 1280, 720, 6, 2), full)
    finally:
        a.close()
        b.close()
        c.close()
        d.close()
        e.close()
        f.close()


def test_degenerate_and_edge_on_triangles(gpu_ctx, mrt, oracle):
    """Triangles the prefilter cannot orient (eye in the triangle's plane, zero
    area after rounding, parallel to an orthographic ray) must still match."""
    tri = np.array([
        # a            b            c
        [0, 0, 2,     1, 0, 2,      0, 1, 2],        # ordinary
        [0, 0, 0,     1, 0, 3,      -1, 0, 3],       # plane contains the eye (y = 0)
        [0, 0, 2,     1e-30, 0, 2,  0, 1e-30, 2],    # vanishing area
        [-1, -1, 1,   1, 1, 5,      -1, -1, 5],      # contains the view axis direction partially
        [0.2, -0.5, 1, 0.2, 0.5, 1, 0.2, 0, 4],      # parallel to ortho rays (x = 0.2 plane)
    ], np.float32).T
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=160, yres=120, left=-1,
                       right=1, top=1, bottom=-1, projection=proj, focal=1.0, shader=mrt.SHADER_PHONG,
                       ka=0.1, kd=0.5, ks=0.3, ambient=(1, 1, 1), background=(0.1, 0.2, 0.3), tri=tri,
                       tri_mat=[0, 1, 0, 1, 0], mat=[[1, 0.2], [0.5, 0.9], [0.2, 0.1], [4, 16]],
                       light=[[2], [2], [-1], [1], [1], [1]])
        o = oracle.render(sc, threads=4)
        r = gpu_render(gpu_ctx, sc)
        assert_same_hits(r, o)
        check(r["rgba"], o["rgba"], o["ids"])


def _random_scene(mrt, rng, proj, xres, yres, n_tris, n_sph, skew=False):
    """Random camera (any orientation, eye away from the origin) looking at a cloud of primitives."""
    eye = rng.uniform(-50, 50, 3)
    view = rng.normal(size=3)
    view /= np.linalg.norm(view)
    up = rng.normal(size=3)
    u, v, w = scene_py.camera_basis(list(view), list(up))
    u, v, w = np.array(u), np.array(v), np.array(w)
    if skew:                       # not orthonormal: the API takes any basis
        u = u + 0.2 * v
        v = 1.3 * v + 0.1 * w
    centre = eye + view * 6.0 if proj == mrt.PROJ_PERSP else view * 6.0   # the ortho origin ignores the eye
    c = centre + rng.uniform(-3, 3, (n_tris, 3))
    tri = np.concatenate([(c + rng.uniform(-0.6, 0.6, (n_tris, 3))).T for _ in range(3)], axis=0)
    sc_ = centre + rng.uniform(-3, 3, (n_sph, 3))
    sph = np.concatenate([sc_.T, rng.uniform(0.05, 0.5, (1, n_sph))], axis=0)
    nm = 5
    mat = np.concatenate([rng.uniform(0, 1, (3, nm)), rng.choice([1.0, 2.0, 4.0, 32.0], (1, nm))], axis=0)
    light = np.concatenate([(centre + rng.uniform(-8, 8, (3, 3))).T, rng.uniform(0, 1, (3, 3))], axis=0)
    return mrt.Scene(eye=eye, u=u, v=v, w=w, xres=xres, yres=yres, left=-1.2, right=0.9, top=0.7, bottom=-1.1,
                     projection=proj, focal=rng.uniform(0.5, 2.0), shader=mrt.SHADER_PHONG, ka=0.1, kd=0.6,
                     ks=0.3, ambient=(1, 0.9, 0.8), background=(0.05, 0.1, 0.2), sph=sph,
                     sph_mat=rng.integers(0, nm, n_sph), tri=tri, tri_mat=rng.integers(0, nm, n_tris), mat=mat,
                     light=light)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_cameras_match_oracle(gpu_ctx, mrt, oracle, seed):
    """Arbitrary camera orientation and position, asymmetric viewport, non-square pixels, both
    projections; seeds 13 and 14 use a basis that is not orthonormal."""
    rng = np.random.default_rng(seed)
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        sc = _random_scene(mrt, rng, proj, 200, 150, 400, 40, skew=seed >= 13)
        o = oracle.render(sc, threads=8)
        assert (o["ids"] >= 0).sum() > 1000
        r = gpu_render(gpu_ctx, sc)
        assert_same_hits(r, o)
        check(r["rgba"], o["rgba"], o["ids"])


@pytest.mark.parametrize("offset", [3.0e3, 1.0e5])
def test_far_from_origin_cancellation(gpu_ctx, mrt, oracle, offset):
    """Scene translated far from the origin: a - o cancels catastrophically in FP32, the exact
    test's determinants are noisy, and the prefilter's error margins have to cover that noise
    (the filter must still be a superset of what the exact arithmetic accepts)."""
    rng = np.random.default_rng(21)
    shift = np.array([offset, -0.7 * offset, 0.4 * offset])
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        base = _random_scene(mrt, rng, proj, 160, 120, 300, 30)
        st = base.struct
        eye = np.array(list(st.eye)) + shift
        tri = base.tri.astype(np.float64).copy()
        sph = base.sph.astype(np.float64).copy()
        light = base.light.astype(np.float64).copy()
        if proj == mrt.PROJ_PERSP:          # move camera and geometry together
            for k in range(3):
                tri[k::3] += shift[k]
            sph[:3] += shift[:, None]
            light[:3] += shift[:, None]
        sc = mrt.Scene(eye=eye, u=list(st.u), v=list(st.v), w=list(st.w), xres=160, yres=120, left=st.left,
                       right=st.right, top=st.top, bottom=st.bottom, projection=proj, focal=st.focal,
                       shader=mrt.SHADER_FLAT, background=(0.05, 0.1, 0.2), sph=sph, sph_mat=base.sph_mat,
                       tri=tri, tri_mat=base.tri_mat, mat=base.mat, light=light)
        o = oracle.render(sc, threads=8)
        r = gpu_render(gpu_ctx, sc)
        assert_same_hits(r, o)
        assert np.array_equal(r["rgba"], o["rgba"])


def test_negative_zero_barycentric_is_a_hit(mrt, oracle):
    """beta = Nb/det rounds to -0 when Nb is a negative denormal and det is huge; the reference
    tests `beta < 0`, which -0 passes, so those pixels hit although they lie (by 1e-43) outside the
    triangle.  Every kernel must reproduce it: none of the rejection tests may drop a pair on
    the sign of a numerator alone.  With a viewport 2^130 times larger the same pixels miss."""
    flag_sets = (0, mrt.FLAG_FULL_TEST, "cull", "block", "grid", mrt.FLAG_NO_CIRCLE, mrt.FLAG_NO_PREFILTER)
    tri = np.array([[0, -0.3, -1, -1e30, -0.3, -1, 0, 0.7, -1]], dtype=np.float64).T
    for e, expect_all in ((2.0 ** -140, True), (2.0 ** -10, False)):
        sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, 1, 0], w=[0, 0, 1], xres=4, yres=2, left=-e, right=e,
                       top=e, bottom=-e, projection=mrt.PROJ_ORTHO, focal=1.0, shader=mrt.SHADER_FLAT,
                       background=(0, 0, 0), tri=tri, tri_mat=[0], mat=[[1.0], [0.5], [0.25], [4.0]])
        o = oracle.render(sc)
        assert bool((o["ids"] == 0).all()) == expect_all
        assert (o["ids"][:, 0] == 0).all()              # the first column is inside either way
        for flags in flag_sets:
            ctx = mrt.renderer(device=0, kind=flags)
            try:
                r = gpu_render(ctx, sc)
            finally:
                ctx.close()
            assert r["stats"].kernel == _expected_kernel(mrt, flags)       # also on a 4 x 2 frame
            assert_same_hits(r, o)
            assert np.array_equal(r["rgba"], o["rgba"])


def _expected_kernel(mrt, kind):
    """The kernel a whole-frame render runs: ctx flags (an integer) or an mrt_accel method (a name)."""
    if isinstance(kind, str):
        return {"cull": mrt.KERNEL_CULL, "block": mrt.KERNEL_TRACE_BLOCK, "grid": mrt.KERNEL_GRID}[kind]
    return mrt.KERNEL_FULL if kind & mrt.FLAG_FULL_TEST else mrt.KERNEL_TRACE


def test_non_finite_and_odd_inputs(mrt, oracle):
    """The C ABI takes any floats.  NaN and infinite coordinates, a NaN radius, a negative radius
    (same sphere, normals pointing inwards: kernel.cl:282-300 divides by the radius), a zero radius
    and triangles at 1e19 (products overflow to infinity) must come out as the oracle has them,
    from every kernel."""
    nan, inf = float("nan"), float("inf")
    tri = np.array([[0, 0, -3, 1, 0, -3, 0, 1, -3],            # ordinary
                    [nan, 0, -2, 1, 0, -2, 0, 1, -2],          # NaN vertex
                    [-1, -1, -2.5, inf, 0, -2.5, 0, 1, -2.5],  # infinite vertex
                    [-1e19, -1e19, -4, 1e19, -1e19, -4, 0, 1e19, -4],   # overflowing products
                    [-0.5, -0.5, -1.5, 0.5, -0.5, -1.5, 0, 0.5, -1.5]], dtype=np.float64).T
    sph = np.array([[-0.8, 0.5, -2, -0.4],      # negative radius
                    [0.8, 0.5, -2, nan],
                    [nan, 0, -2, 0.3],
                    [0.8, -0.5, -2, 0.0],
                    [0.5, 0.6, -2.2, 0.3]], dtype=np.float64).T
    flag_sets = (0, mrt.FLAG_FULL_TEST, "cull", "block", "grid", mrt.FLAG_NO_PREFILTER)
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, 1, 0], w=[0, 0, 1], xres=96, yres=64, left=-1.2,
                       right=1.2, top=0.8, bottom=-0.8, projection=proj, focal=1.0, shader=mrt.SHADER_PHONG,
                       ka=0.2, kd=0.6, ks=0.3, ambient=(1, 1, 1), background=(0.1, 0.2, 0.3), sph=sph,
                       sph_mat=[0, 1, 0, 1, 1], tri=tri, tri_mat=[0, 1, 0, 1, 0],
                       mat=[[1.0, 0.2], [0.5, 0.9], [0.25, 0.4], [4.0, 16.0]],
                       light=[[2.0], [3.0], [1.0], [1.0], [1.0], [1.0]])
        o = oracle.render(sc)
        assert len(np.unique(o["ids"])) >= 3          # the NaN-radius sphere "hits" at t = -b/a wherever that is > 0
        for flags in flag_sets:
            ctx = mrt.renderer(device=0, kind=flags)
            try:
                r = gpu_render(ctx, sc)
            finally:
                ctx.close()
            assert r["stats"].kernel == _expected_kernel(mrt, flags)
            assert_same_hits(r, o)
            check(r["rgba"], o["rgba"], o["ids"])


def test_equal_t_goes_to_the_lower_object_id(mrt, oracle):
    """Coincident primitives give equal t; the reference keeps the first in object order (strict
    '<', kernel.cl:326,344).  The full-test kernel evaluates triangles grouped by hit sign, in an
    order its atomics decide, so this must hold by its id tie-break -- on every run."""
    rng = np.random.default_rng(77)
    base = rng.uniform(-1, 1, (40, 9)) * 0.8 + np.tile([0, 0, -3.0], 3)
    flipped = base[:, [0, 1, 2, 6, 7, 8, 3, 4, 5]]                 # same triangles, opposite winding
    tri = np.concatenate([base, flipped, base[::-1], flipped[5:25]], axis=0).T
    n_tri = tri.shape[1]
    sph = np.array([[0.3, 0.2, -2.0, 0.5], [0.3, 0.2, -2.0, 0.5], [-0.6, -0.4, -2.5, 0.4],
                    [-0.6, -0.4, -2.5, 0.4]]).T
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, 1, 0], w=[0, 0, 1], xres=160, yres=120, left=-1.3,
                       right=1.3, top=1.0, bottom=-1.0, projection=proj, focal=1.0, shader=mrt.SHADER_FLAT,
                       background=(0, 0, 0), sph=sph, sph_mat=[0, 1, 2, 3], tri=tri,
                       tri_mat=np.arange(n_tri) % 7, mat=[np.linspace(0.1, 1, 7), np.linspace(1, 0.1, 7),
                                                          np.full(7, 0.5), np.full(7, 4.0)])
        o = oracle.render(sc)
        assert len(np.unique(o["ids"])) > 20
        for flags in (mrt.FLAG_FULL_TEST, 0, "cull", "grid", mrt.FLAG_NO_PREFILTER):
            for _ in range(4 if flags == mrt.FLAG_FULL_TEST else 1):
                ctx = mrt.renderer(device=0, kind=flags)
                try:
                    r = gpu_render(ctx, sc)
                finally:
                    ctx.close()
                assert r["stats"].kernel == _expected_kernel(mrt, flags)
                assert_same_hits(r, o)
                assert np.array_equal(r["rgba"], o["rgba"])


def test_camera_inside_and_behind_geometry(gpu_ctx, mrt, oracle):
    """Primitives that straddle or lie behind the eye plane, a sphere containing the eye, huge and
    tiny primitives: the prefilter must stay a superset wherever its projection is unbounded."""
    tri = np.array([
        [-1, -1, -2,   1, -1, 3,    0, 1, 3],          # crosses the eye plane z = 0
        [-5, -5, -1,   5, -5, -1,   0, 5, -1],         # entirely behind the camera
        [-100, -100, 50, 100, -100, 50, 0, 100, 50],   # covers the whole view, far away
        [0.1, 0.1, 1,  0.1001, 0.1, 1, 0.1, 0.1001, 1],  # far smaller than a pixel
        [-0.5, 0.2, 0.05, 0.5, 0.2, 0.05, 0, 0.3, 0.05],  # very close to the eye
    ], np.float32).T
    sph = np.array([[0, 0, 0, 2.0], [0.3, 0.3, 4, 0.5], [0, 0, -3, 1.0], [2, 0, 0.2, 1.0],
                    [0, 0, 30, 1e-3]], np.float32).T
    for proj in (mrt.PROJ_PERSP, mrt.PROJ_ORTHO):
        sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=160, yres=120, left=-1,
                       right=1, top=1, bottom=-1, projection=proj, focal=1.0, shader=mrt.SHADER_PHONG, ka=0.1,
                       kd=0.5, ks=0.3, ambient=(1, 1, 1), background=(0.1, 0.2, 0.3), tri=tri,
                       tri_mat=[0, 1, 0, 1, 0], sph=sph, sph_mat=[1, 0, 1, 0, 1],
                       mat=[[1, 0.2], [0.5, 0.9], [0.2, 0.1], [4, 16]], light=[[2], [2], [-1], [1], [1], [1]])
        o = oracle.render(sc, threads=4)
        r = gpu_render(gpu_ctx, sc)
        assert_same_hits(r, o)
        check(r["rgba"], o["rgba"], o["ids"])


def test_empty_scene_and_no_lights(gpu_ctx, mrt, oracle):
    sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=70, yres=50, left=-1, right=1,
                   top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0, shader=mrt.SHADER_PHONG, ka=0.3,
                   kd=0.5, ks=0.2, ambient=(1, 0.5, 0.25), background=(0.25, 0.5, 0.75))
    r = gpu_render(gpu_ctx, sc)
    assert (r["ids"] == -1).all() and (r["rgba"] == [63, 127, 191, 255]).all()
    # objects but no lights: ambient term only
    sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=70, yres=50, left=-1, right=1,
                   top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0, shader=mrt.SHADER_PHONG, ka=0.3,
                   kd=0.5, ks=0.2, ambient=(1, 0.5, 0.25), background=(0.25, 0.5, 0.75),
                   sph=[[0], [0], [3], [1]], sph_mat=[0], mat=[[1], [1], [1], [4]])
    o = oracle.render(sc)
    r = gpu_render(gpu_ctx, sc)
    assert_same_hits(r, o)
    assert np.array_equal(r["rgba"], o["rgba"])


def test_teatime_4k_full_frame_against_oracle(gpu_ctx, mrt, oracle):
    """The benchmark frame itself: teatime at 3840x2160, Blinn-Phong, every pixel against the oracle
    (3.3e10 ray-primitive tests on the host cores, about 10-20 s on 16 threads)."""
    ls = mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy()
    r = gpu_render(gpu_ctx, ls)
    assert r["stats"].kernel == mrt.KERNEL_TRACE_WIDE                      # 8.3 Mpixel: 8x4 pixels per lane
    o = oracle.render(ls, threads=16)
    assert (o["ids"] >= 0).sum() > 300000
    assert_same_hits(r, o)
    stats = check(r["rgba"], o["rgba"], o["ids"])
    assert stats["within_1lsb"] == stats["pixels"]
    # the other kernels on the same frame: the 4x4 shape of the default kernel, a 3x3 solve for
    # every pair, and the accelerated entry points
    for flags in (mrt.FLAG_TRACE_NARROW, mrt.FLAG_FULL_TEST, "cull", "block", "grid"):
        ctx = mrt.renderer(device=0, kind=flags)
        try:
            r2 = gpu_render(ctx, ls)
        finally:
            ctx.close()
        assert r2["stats"].kernel == _expected_kernel(mrt, flags)          # 2160 rows: no multiple of 32
        assert_same_hits(r2, o)
        assert np.array_equal(r2["rgba"], r["rgba"])


def test_frames_meet_parity_under_either_reading_of_contraction(gpu_ctx, mrt, oracle):
    """The HIP frames against the oracle's source built BOTH ways -- one rounding per operator (the
    oracle) and multiply-adds fused as an OpenCL compiler may fuse kernel.cl:149-172 -- on the nine
    shipped scenes and on the rows of the 4K frame that tests/golden/oracle_contraction.json
    covers: the stated tolerance (<= 1 LSB on >= 99.9 % of pixels, the rest on primitive edges)
    holds against either."""
    import json
    rec = json.loads((GOLDEN / "oracle_contraction.json").read_text())
    for name in ALL_SCENES:
        sc = scene_py.load_scene(SCENES / f"{name}.json")
        g = gpu_render(gpu_ctx, sc)
        for contracted in (False, True):
            o = oracle.render(sc, threads=8, contracted=contracted)
            stats = check(g["rgba"], o["rgba"], o["ids"])
            assert stats["within_1lsb"] == stats["pixels"], (name, contracted, stats)
            assert int((g["ids"] != o["ids"]).sum()) == 0, (name, contracted)
    sc = mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy()
    g = gpu_render(gpu_ctx, sc)
    for contracted in (False, True):
        for r0, r1 in rec["teatime_3840x2160_rows"]["rows"]:
            a0, a1 = r0 - 1, r1 + 1                               # a row of context for the edge mask
            o = oracle.render(sc, rows=(a0, a1), threads=16, contracted=contracted)
            stats = check(g["rgba"][a0:a1], o["rgba"][a0:a1], o["ids"][a0:a1])
            assert stats["within_1lsb"] == stats["pixels"], (contracted, r0, stats)
            assert np.array_equal(g["ids"][r0:r1], o["ids"][r0:r1]), (contracted, r0)


def test_frames_meet_parity_whatever_the_device_built_ins_round_to(gpu_ctx, mrt, oracle):
    """The HIP frames against the oracle's source built with its divisions, square roots,
    normalize(), dot() and powr() moved by whole ulps inside what OpenCL 1.2 lets a conforming device
    be off by (oracle/builtin_latitude.py: pseudo-randomly, all up, all down) -- on the nine shipped
    scenes and on the rows of the 4K frame that tests/golden/oracle_builtin_latitude.json covers:
    the stated tolerance (<= 1 LSB on >= 99.9 % of pixels, the rest on primitive edges) holds against
    every one of them, and so do the winners."""
    import json
    rec = json.loads((GOLDEN / "oracle_builtin_latitude.json").read_text())
    for name in ALL_SCENES:
        sc = scene_py.load_scene(SCENES / f"{name}.json")
        g = gpu_render(gpu_ctx, sc)
        for mode in (0, 1, 2):
            o = oracle.render(sc, threads=8, ulp_mode=mode)
            stats = check(g["rgba"], o["rgba"], o["ids"])
            assert stats["within_1lsb"] == stats["pixels"], (name, mode, stats)
            assert int((g["ids"] != o["ids"]).sum()) == 0, (name, mode)
    sc = mrt.LoadedScene(text=mkscene.teatime(3840, 2160)).copy()
    g = gpu_render(gpu_ctx, sc)
    for mode in (0, 1, 2):
        for r0, r1 in rec["teatime_3840x2160_rows"]["rows"][::4]:
            a0, a1 = r0 - 1, r1 + 1                               # a row of context for the edge mask
            o = oracle.render(sc, rows=(a0, a1), threads=16, ulp_mode=mode)
            stats = check(g["rgba"][a0:a1], o["rgba"][a0:a1], o["ids"][a0:a1])
            assert stats["within_1lsb"] == stats["pixels"], (mode, r0, stats)
            assert np.array_equal(g["ids"][r0:r1], o["ids"][r0:r1]), (mode, r0)


def test_soup_all_primitives_against_oracle(gpu_ctx, mrt, oracle):
    """The benchmark soup (16 384 triangles + 1024 spheres, 8 lights) at 384x384: whole frame vs oracle."""
    sc = mrt.LoadedScene(text=mkscene.soup(384, 384)).copy()
    assert sc.struct.n_tris == 16384 and sc.struct.n_spheres == 1024 and sc.struct.n_lights == 8
    o = oracle.render(sc, threads=16)
    r = gpu_render(gpu_ctx, sc)
    assert (o["ids"] >= 0).sum() > 10000
    assert_same_hits(r, o)
    check(r["rgba"], o["rgba"], o["ids"])


def test_soup_16k_rows_against_oracle(mrt, oracle):
    """The 16384 x 16384 soup benchmark frame at full size (one GPU renders the whole frame here);
    the oracle re-renders single rows."""
    sc = mrt.LoadedScene(text=mkscene.soup(16384, 16384)).copy()
    ctx = mrt.Context(device=0)
    try:
        ctx.set_scene(sc)
        img, ids, t, st = ctx.render(want_ids=True)
    finally:
        ctx.close()
    assert st.rays == 16384 * 16384
    for y in (0, 5000, 8191, 16383):
        o = oracle.render(sc, rows=(y, y + 1), threads=16)
        assert np.array_equal(ids[y], o["ids"][y])
        assert np.array_equal(t[y], o["t"][y])
        check(img[y:y + 1], o["rgba"][y:y + 1], o["ids"][y:y + 1])
    # size-independent property at the full size: the tile-cull and grid kernels must reproduce
    # all 268 435 456 winners and t values of the default kernel
    for method in ("cull", "grid"):
        ctx = mrt.Accel(device=0, method=method)
        try:
            ctx.set_scene(sc)
            img2, ids2, t2, st2 = ctx.render(want_ids=True)
        finally:
            ctx.close()
        assert np.array_equal(ids2, ids) and np.array_equal(t2.view(np.uint32), t.view(np.uint32)), method
        assert np.array_equal(img2, img), method
        del img2, ids2, t2
    assert st2.kernel == mrt.KERNEL_GRID and st2.cells == 512 * 512 and st2.refs < 40 * st2.cells
    del img, ids, t


def test_set_camera_equals_fresh_scene(mrt, oracle):
    """mrt_set_camera rebuilds only the per-view records: a sequence of views through one context
    must equal a fresh mrt_set_scene (and the oracle) for each view."""
    rng = np.random.default_rng(31)
    base = _random_scene(mrt, rng, mrt.PROJ_PERSP, 160, 120, 300, 30)
    ctx = mrt.Context(device=0)
    fresh = mrt.Context(device=0)
    try:
        ctx.set_scene(base)
        for k in range(4):
            view = _random_scene(mrt, rng, mrt.PROJ_PERSP if k % 2 else mrt.PROJ_ORTHO, 100 + 37 * k, 90 + 11 * k, 1, 1)
            st = view.struct
            sc = mrt.Scene(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=st.xres, yres=st.yres,
                           left=st.left, right=st.right, top=st.top, bottom=st.bottom, projection=st.projection,
                           focal=st.focal, shader=base.struct.shader, ka=base.struct.ka, kd=base.struct.kd,
                           ks=base.struct.ks, ambient=list(base.struct.ambient),
                           background=list(base.struct.background), sph=base.sph, sph_mat=base.sph_mat,
                           tri=base.tri, tri_mat=base.tri_mat, mat=base.mat, light=base.light)
            ctx.set_camera(sc)
            a = ctx.render(want_ids=True)
            fresh.set_scene(sc)
            b = fresh.render(want_ids=True)
            o = oracle.render(sc, threads=8)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(a[1], o["ids"]) and np.array_equal(a[2], o["t"])
        # mrt_set_camera does not wait: views and frames enqueued back to back, the frames on a
        # stream that is not the context's own, each into its own buffer; every frame must show its view
        import torch
        views = [_random_scene(mrt, rng, mrt.PROJ_PERSP if k % 2 else mrt.PROJ_ORTHO, 160, 120, 1, 1) for k in range(6)]
        side = torch.cuda.Stream()
        bufs = [torch.zeros(160 * 120 * 4, dtype=torch.uint8, device="cuda:0") for _ in views]
        torch.cuda.synchronize()
        for view, buf in zip(views, bufs):
            ctx.set_camera(view)
            ctx.render_bands_device(120, 0, 1, buf.data_ptr(), buf.numel(), stream=side.cuda_stream)
        torch.cuda.synchronize()
        for view, buf in zip(views, bufs):
            st = view.struct
            sc = mrt.Scene(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=160, yres=120,
                           left=st.left, right=st.right, top=st.top, bottom=st.bottom, projection=st.projection,
                           focal=st.focal, shader=base.struct.shader, ka=base.struct.ka, kd=base.struct.kd,
                           ks=base.struct.ks, ambient=list(base.struct.ambient),
                           background=list(base.struct.background), sph=base.sph, sph_mat=base.sph_mat,
                           tri=base.tri, tri_mat=base.tri_mat, mat=base.mat, light=base.light)
            fresh.set_scene(sc)
            want, _ = fresh.render()
            assert np.array_equal(buf.cpu().numpy().reshape(120, 160, 4), want)
    finally:
        ctx.close()
        fresh.close()


def _scene_with_view(mrt, base, view):
    """`base` (geometry, materials, lights, shader) seen through the camera and viewport of `view`."""
    st, b = view.struct, base.struct
    return mrt.Scene(eye=list(st.eye), u=list(st.u), v=list(st.v), w=list(st.w), xres=st.xres, yres=st.yres,
                     left=st.left, right=st.right, top=st.top, bottom=st.bottom, projection=st.projection,
                     focal=st.focal, shader=b.shader, ka=b.ka, kd=b.kd, ks=b.ks, ambient=list(b.ambient),
                     background=list(b.background), sph=base.sph, sph_mat=base.sph_mat, tri=base.tri,
                     tri_mat=base.tri_mat, mat=base.mat, light=base.light)


@pytest.mark.parametrize("in_flight", [1, 2, 3])
def test_sequence_of_views_equals_fresh_contexts(mrt, in_flight):
    """mrt_sequence: views submitted back to back, in_flight of them enqueued side by side, each
    into its own buffer; every frame equals what a fresh context renders for its view, also when
    the view stays (NULL) and when the resolution changes in mid-sequence."""
    import torch
    rng = np.random.default_rng(77)
    base = _random_scene(mrt, rng, mrt.PROJ_PERSP, 320, 200, 900, 40)
    b = base.struct
    eye0, w0, v0 = np.array(list(b.eye)), np.array(list(b.w)), np.array(list(b.v))

    def nearby_view(k):
        """The camera of `base` moved and turned a little, viewport and resolution varied."""
        u, v, w = scene_py.camera_basis(list(-w0 + rng.normal(scale=0.08, size=3)), list(v0 + rng.normal(scale=0.1, size=3)))
        xres, yres = (320, 200) if k < 5 else (256, 160)
        return mrt.Scene(eye=eye0 + rng.normal(scale=0.3, size=3), u=u, v=v, w=w, xres=xres, yres=yres,
                         left=-1.2 + 0.05 * k, right=0.9, top=0.7 + 0.03 * k, bottom=-1.1,
                         projection=mrt.PROJ_PERSP, focal=float(b.focal) * (1 + 0.05 * k), shader=b.shader, ka=b.ka,
                         kd=b.kd, ks=b.ks, ambient=list(b.ambient), background=list(b.background), sph=base.sph,
                         sph_mat=base.sph_mat, tri=base.tri, tri_mat=base.tri_mat, mat=base.mat, light=base.light)

    views = [None]                                                   # first frame: the scene's own view
    for k in range(7):
        views.append(nearby_view(k))
        if k in (2, 3, 6):
            views.append(None)                                       # the same view once more
    seq = mrt.Sequence(device=0, in_flight=in_flight)
    fresh = mrt.Context(device=0)
    try:
        assert seq.in_flight == in_flight
        with pytest.raises(mrt.MrtError) as e:
            seq.submit(200, 0, 1, 0, 0)
        assert e.value.code == -4
        seq.set_scene(base)
        bufs = [torch.zeros(320 * 200 * 4, dtype=torch.uint8, device="cuda:0") for _ in views]
        torch.cuda.synchronize()
        streams = set()
        current = base
        for view, buf in zip(views, bufs):
            current = view if view is not None else current
            streams.add(seq.submit(current.struct.yres, 0, 1, buf.data_ptr(), buf.numel(), view=view))
        assert len(streams) == in_flight and 0 not in streams
        seq.wait()
        current = base
        for k, (view, buf) in enumerate(zip(views, bufs)):
            current = view if view is not None else current
            fresh.set_scene(current)
            want, _ = fresh.render()
            assert len(np.unique(want.reshape(-1, 4), axis=0)) > 20, k     # the view does show the scene
            st = current.struct
            got = buf.cpu().numpy()[:st.xres * st.yres * 4].reshape(st.yres, st.xres, 4)
            assert np.array_equal(got, want), k
        # the same sequence into host memory (mrt_sequence_submit_host): the library's own device
        # buffers and copy stream, page-locked and pageable destinations
        current, outs = base, []
        seq.set_scene(base)
        for k, view in enumerate(views):
            current = view if view is not None else current
            st = current.struct
            out = mrt.host_array((st.yres, st.xres, 4)) if k % 3 else np.zeros((st.yres, st.xres, 4), np.uint8)
            seq.submit_host(out, view=view)
            outs.append((current, out))
        seq.wait()
        for k, (sc_k, out) in enumerate(outs):
            fresh.set_scene(sc_k)
            want, _ = fresh.render()
            assert np.array_equal(out, want), k
        with pytest.raises(ValueError):
            seq.submit_host(np.zeros((3, 3, 4), np.uint8))
        # a rejected view leaves the sequence where it was
        bad = nearby_view(0)
        bad.struct.projection = 7
        with pytest.raises(mrt.MrtError):
            seq.submit(200, 0, 1, bufs[0].data_ptr(), bufs[0].numel(), view=bad)
        st = current.struct
        seq.submit(st.yres, 0, 1, bufs[0].data_ptr(), bufs[0].numel())
        seq.wait()
        assert np.array_equal(bufs[0].cpu().numpy()[:st.xres * st.yres * 4].reshape(st.yres, st.xres, 4), want)
    finally:
        seq.close()
        fresh.close()
    with pytest.raises(mrt.MrtError) as e:
        mrt.Sequence(device=0, in_flight=5)
    assert e.value.code == -5


def test_sequence_slots_at_two_megapixels(gpu_ctx, mrt):
    """The slots of a sequence choose lane shape and slice count for a GPU that other frames keep
    full (8x4 lane blocks from 2 Mpixel, fewer slices): same bytes as a context's frame, at a size
    where the two choices differ."""
    sc = mrt.LoadedScene(text=mkscene.teatime(1920, 1080)).copy()
    gpu_ctx.set_scene(sc)
    want, st = gpu_ctx.render()
    assert st.kernel == mrt.KERNEL_TRACE
    seq = mrt.Sequence(device=0)
    try:
        seq.set_scene(sc)
        outs = [mrt.host_array((1080, 1920, 4)) for _ in range(5)]
        for o in outs:
            seq.submit_host(o)
        seq.wait()
        for o in outs:
            assert np.array_equal(o, want)
    finally:
        seq.close()


def test_slice_and_layer_count_invariance(mrt):
    """The triangle list cut into any number of slices (trace_kernel, both lane shapes) or layers
    (full_trace_kernel) gives the same winners, t and bytes as one piece: the pieces meet in the
    key plane, whose minimum does not depend on how the list was cut (the override is the
    library's tuning aid, also reachable as MRT_FORCE_SLICES when a context is created)."""
    scenes = [mrt.LoadedScene(text=mkscene.teatime(640, 360)).copy(),
              mrt.LoadedScene(text=mkscene.soup(333, 222, n_tris=1500, n_spheres=700, seed=3, projection="ortho")).copy()]
    for flags in (mrt.FLAG_TRACE_NARROW, mrt.FLAG_TRACE_WIDE, mrt.FLAG_FULL_TEST):
        ctx = mrt.Context(device=0, flags=flags)
        try:
            for sc in scenes:
                ctx.force_slices(1)
                ref = gpu_render(ctx, sc)
                assert ref["stats"].prim_slices == 1
                seen = {1}
                for n in (2, 3, 5, 16, 64, 1000):
                    ctx.force_slices(n)
                    r = gpu_render(ctx, sc)
                    seen.add(r["stats"].prim_slices)
                    assert_same_hits(r, ref)
                    assert np.array_equal(r["rgba"], ref["rgba"]), (flags, n)
                assert len(seen) >= 4, seen          # the override did cut the list in different ways
                ctx.force_slices(0)
                r = gpu_render(ctx, sc)
                assert_same_hits(r, ref)
                assert np.array_equal(r["rgba"], ref["rgba"])
        finally:
            ctx.close()


def test_render_is_deterministic(gpu_ctx, mrt):
    ls = mrt.LoadedScene(text=mkscene.teatime(1280, 720)).copy()
    a = gpu_render(gpu_ctx, ls)
    b = gpu_render(gpu_ctx, ls)
    assert np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["ids"], b["ids"])
    # into page-locked memory from mrt_host_alloc: same bytes; the block outlives the first view
    pinned = mrt.host_array((720, 1280, 4))
    img, _ = gpu_ctx.render(out=pinned)
    assert img is pinned and np.array_equal(pinned, a["rgba"])
    view = pinned[100:200]
    del pinned, img
    assert np.array_equal(view, a["rgba"][100:200])
    with pytest.raises(ValueError):
        gpu_ctx.render(out=np.empty((10, 10, 4), np.uint8))


def test_frames_in_flight_on_two_contexts(gpu_ctx, mrt):
    """Frames enqueued side by side, each on a context and a stream of its own (how bench.py keeps
    two frames in flight): every frame equals the frame one context renders alone, also when the
    two contexts hold different views."""
    import torch
    xres, yres = 1920, 1080
    views = [mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy(),
             mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy().with_resolution(
                 xres, yres, left=-1.0, right=1.5, top=0.7, bottom=-0.9)]
    want = []
    for sc in views:
        gpu_ctx.set_scene(sc)
        want.append(gpu_ctx.render()[0])
    ctxs = [mrt.Context(device=0) for _ in views]
    try:
        streams = [torch.cuda.Stream() for _ in views]
        bufs = [[torch.zeros(xres * yres * 4, dtype=torch.uint8, device="cuda:0") for _ in range(6)] for _ in views]
        for c, sc in zip(ctxs, views):
            c.set_scene(sc)
        torch.cuda.synchronize()
        for k in range(6):
            for f, c in enumerate(ctxs):
                c.render_bands_device(64, 0, 1, bufs[f][k].data_ptr(), bufs[f][k].numel(),
                                      stream=streams[f].cuda_stream)
        torch.cuda.synchronize()
        for f in range(len(views)):
            for k in range(6):
                assert np.array_equal(bufs[f][k].cpu().numpy().reshape(yres, xres, 4), want[f]), (f, k)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("shape", ["narrow", "wide"])
@pytest.mark.parametrize("world,band", [(2, 64), (3, 16), (8, 64), (4, 7)])
def test_band_partition_invariance(gpu_ctx, mrt, world, band, shape):
    """Interleaved row bands rendered rank by rank into device memory and put
    back together equal the single full-frame render, with both lane shapes of the default kernel."""
    import torch
    ls = mrt.LoadedScene(text=mkscene.teatime(640, 360)).copy()
    gpu_ctx.set_scene(ls)
    full, _ = gpu_ctx.render()
    xres, yres = 640, 360
    rows = [mrt.band_rows(yres, band, r, world) for r in range(world)]
    stride = max(rows) * xres * 4
    gathered = torch.zeros((world, stride), dtype=torch.uint8, device="cuda:0")
    ctx = mrt.Context(device=0, flags=mrt.FLAG_TRACE_WIDE if shape == "wide" else mrt.FLAG_TRACE_NARROW)
    try:
        ctx.set_scene(ls)
        for r in range(world):
            st = ctx.render_bands_device(band, r, world, gathered[r].data_ptr(), stride,
                                         stream=torch.cuda.current_stream().cuda_stream, timed=True)
            assert st.kernel == (mrt.KERNEL_TRACE_WIDE if shape == "wide" else mrt.KERNEL_TRACE)
        torch.cuda.synchronize()
    finally:
        ctx.close()
    img = mrt.bands_to_image(gathered.cpu().numpy(), stride, xres, yres, band, world)
    assert np.array_equal(img, full)


def test_teatime_8k_banded_against_oracle(mrt, oracle):
    """teatime at 7680 x 4320 cut into interleaved 64-row bands for 2, 4 and 8 ranks (each rank's
    bands rendered into its own device buffer, one rank after the other on this one GPU, then put
    back together on the host) must equal the single-launch frame pixel for pixel, and every pixel
    of that frame must carry the oracle's winner and t, and its colour within 1 LSB."""
    import torch
    xres, yres, band = 7680, 4320, 64
    sc = mrt.LoadedScene(text=mkscene.teatime(xres, yres)).copy()
    ctx = mrt.Context(device=0)
    try:
        ctx.set_scene(sc)
        full, ids, t, st = ctx.render(want_ids=True)
        assert st.rays == xres * yres
        # every pixel of the frame against the oracle (1.3e11 ray-primitive tests on the host
        # cores, about a minute on 16 threads), in four slabs to bound the host memory in use
        for y0 in range(0, yres, 1080):
            o = oracle.render(sc, rows=(y0, y0 + 1080), threads=16)
            assert np.array_equal(ids[y0:y0 + 1080], o["ids"][y0:y0 + 1080]), y0
            assert np.array_equal(t[y0:y0 + 1080], o["t"][y0:y0 + 1080]), y0
            stats = check(full[y0:y0 + 1080], o["rgba"][y0:y0 + 1080], o["ids"][y0:y0 + 1080])
            assert stats["within_1lsb"] == stats["pixels"], y0
            del o
        assert (ids >= 0).sum() > 1200000
        del ids, t
        for world in (2, 4, 8):
            rows = [mrt.band_rows(yres, band, r, world) for r in range(world)]
            assert sum(rows) == yres
            stride = max(rows) * xres * 4
            gathered = torch.zeros((world, stride), dtype=torch.uint8, device="cuda:0")
            for r in range(world):
                ctx.render_bands_device(band, r, world, gathered[r].data_ptr(), stride,
                                        stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            img = mrt.bands_to_image(gathered.cpu().numpy(), stride, xres, yres, band, world)
            assert np.array_equal(img, full), world
            del gathered, img
    finally:
        ctx.close()


def test_api_errors(mrt):
    ctx = mrt.Context(device=0)
    try:
        with pytest.raises(mrt.MrtError) as e:
            ctx.xres = ctx.yres = 4
            ctx.render()
        assert e.value.code == -4        # MRT_ERR_NO_SCENE
        with pytest.raises(mrt.MrtError):
            mrt.Context(device=99)
        with pytest.raises(mrt.MrtError) as e:
            ctx.set_camera(mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=8, yres=8,
                                     left=-1, right=1, top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0))
        assert e.value.code == -4        # no scene yet
        huge = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=70000, yres=8, left=-1,
                         right=1, top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0)
        with pytest.raises(mrt.MrtError) as e:
            ctx.set_scene(huge)
        assert e.value.code == -5        # MRT_ERR_TOO_LARGE
        bad = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=8, yres=8, left=-1,
                        right=1, top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0,
                        sph=[[0], [0], [3], [1]], sph_mat=[5], mat=[[1], [1], [1], [4]])
        with pytest.raises(mrt.MrtError) as e:
            ctx.set_scene(bad)
        assert "material index" in e.value.message
    finally:
        ctx.close()


def test_cli_end_to_end(tmp_path):
    from PIL import Image
    cli = ROOT / "opencl-raytracing_amd" / "mrraytracer"
    for name in ("scene_2spheres_ortho_flat", "scene_gtri_persp_phong", "teatime"):
        out = tmp_path / f"{name}.png"
        r = subprocess.run([str(cli), str(SCENES / f"{name}.json"), str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout.endswith(f"raytracing...done\nwriting {out}...done\n\n")
        g = np.load(GOLDEN / f"{name}.npz")
        with Image.open(out) as im:
            assert im.mode == "RGB"
            rgb = np.asarray(im)
        rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=-1)
        check(rgba, g["rgba"], g["ids"])


def test_turntable_example(tmp_path, mrt):
    """mrturntable: a C++ client of mrt_sequence_submit_host.  The middle frame of an odd number
    of frames is the scene's own view -- the bytes mrraytracer writes -- and every other frame is
    what a context renders for that frame's camera (the basis panned about v)."""
    import json
    from PIL import Image
    tt = ROOT / "opencl-raytracing_amd" / "mrturntable"
    cli = ROOT / "opencl-raytracing_amd" / "mrraytracer"
    scene = SCENES / "teatime.json"
    r = subprocess.run([str(tt), str(scene), str(tmp_path / "v"), "5", "40"], capture_output=True, text=True,
                       env=dict(__import__("os").environ, MRT_STATS="1", MRT_IN_FLIGHT="3"))
    assert r.returncode == 0, r.stderr
    st = json.loads(r.stderr.strip().splitlines()[-1])
    assert st["frames"] == 5 and st["in_flight"] == 3 and st["xres"] == 400
    one = tmp_path / "one.png"
    assert subprocess.run([str(cli), str(scene), str(one)], capture_output=True).returncode == 0
    assert (tmp_path / "v002.png").read_bytes() == one.read_bytes()
    ls = mrt.LoadedScene(path=scene)
    base = ls.copy()
    info_u, info_w = np.array(list(ls.info.u), np.float64), np.array(list(ls.info.w), np.float64)
    ctx = mrt.Context(device=0)
    try:
        ctx.set_scene(base)
        for k in (0, 1, 3, 4):
            a = (k / 4 - 0.5) * 40.0 * 3.14159265358979323846 / 180.0       # as the example forms it
            view = base.with_resolution(400, 400)
            for i in range(3):
                view.struct.u[i] = np.float32(np.cos(a) * info_u[i] - np.sin(a) * info_w[i])
                view.struct.w[i] = np.float32(np.sin(a) * info_u[i] + np.cos(a) * info_w[i])
            ctx.set_camera(view)
            want, _ = ctx.render()
            with Image.open(tmp_path / f"v{k:03d}.png") as im:
                assert np.array_equal(np.asarray(im), want[..., :3]), k
    finally:
        ctx.close()
    assert subprocess.run([str(tt), str(scene), str(tmp_path / "v"), "0"], capture_output=True).returncode == 255


def test_dist_cli_two_ranks_on_one_gpu(tmp_path):
    """mrraytracer_dist.py (one process per GPU, bands gathered to rank 0) with two ranks that
    share GPU 0 and gather over gloo: same stdout contract and the same pixels as the binary."""
    import os
    import sys
    from PIL import Image
    script = ROOT / "opencl-raytracing_amd" / "host" / "mrraytracer_dist.py"
    name = "teatime"
    out = tmp_path / "dist.png"
    env = dict(os.environ, MRT_DIST_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script),
                        str(SCENES / f"{name}.json"), str(out)], capture_output=True, text=True, env=env,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"raytracing...done\nwriting {out}...done\n\n" in r.stdout
    g = np.load(GOLDEN / f"{name}.npz")
    with Image.open(out) as im:
        assert im.mode == "RGB"
        rgb = np.asarray(im)
    rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=-1)
    check(rgba, g["rgba"], g["ids"])
    # a scene file that does not load: every rank exits 1, rank 0 reports it
    bad = tmp_path / "bad.json"
    bad.write_text("[1, 2]")
    r = subprocess.run([sys.executable, str(script), str(bad), str(out)], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode == 1 and "rayson must be comprised of one JSON object" in r.stderr


def test_frames_that_arrive_row_by_row(gpu_ctx, mrt):
    """mrt_render_rgba8_rows and mrt_multi_render_rgba8_rows: the bytes of mrt_render_rgba8, the
    callback covering every row exactly once, in order, and only rows that are already final in the
    caller's buffer; fed to mrt_png_stream_* they give the PNG mrt_write_png_rgb writes."""
    import tempfile
    from pathlib import Path
    for text, strips in ((mkscene.teatime(1920, 1080), (0, 64, 100, 5000)), (mkscene.soup(2048, 2048), (0, 256)),
                         (mkscene.teatime(640, 360), (0, 7))):
        sc = mrt.LoadedScene(text=text).copy()
        gpu_ctx.set_scene(sc)
        want, _ = gpu_ctx.render()
        for strip in strips:
            out = mrt.host_array(want.shape)
            out[:] = 0x5A
            seen = []

            def on_rows(a, b):
                assert np.array_equal(out[a:b], want[a:b])           # final when reported
                seen.append((a, b))

            st = gpu_ctx.render_rows(out, on_rows, strip_rows=strip)
            assert np.array_equal(out, want), strip
            assert seen[0][0] == 0 and seen[-1][1] == want.shape[0]
            assert all(x[1] == y[0] for x, y in zip(seen, seen[1:])) and st.rays == want.shape[0] * want.shape[1]
            if strip == 64:
                assert len(seen) == (want.shape[0] + 63) // 64
        for gather in (mrt.GATHER_PEER, mrt.GATHER_HOST):
            m = mrt.MultiContext(devices=[0, 0, 0], gather=gather)
            try:
                m.set_scene(sc)
                out = mrt.host_array(want.shape)
                out[:] = 0
                seen = []
                with tempfile.TemporaryDirectory() as d:
                    png = mrt.PngStream(Path(d) / "s.png", out)

                    def on_band(a, b):
                        assert np.array_equal(out[a:b], want[a:b])
                        seen.append((a, b))
                        png.rows(a, b)

                    m.render_rows(out, on_band)
                    png.end()
                    mrt.write_png_rgb(Path(d) / "p.png", want)
                    assert (Path(d) / "s.png").read_bytes() == (Path(d) / "p.png").read_bytes()
                assert np.array_equal(out, want)
                assert seen[0][0] == 0 and seen[-1][1] == want.shape[0] and all(x[1] == y[0] for x, y in zip(seen, seen[1:]))
            finally:
                m.close()
