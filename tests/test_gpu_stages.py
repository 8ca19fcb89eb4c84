"""Stage-level entry points (the six cl_* helpers of CLHelpers.hpp:76-322) on
the GPU against the oracle's restatement of the same kernels.  Everything but
the Blinn-Phong colour (powf) must be bit-identical."""
import numpy as np
import pytest

from conftest import SCENES
import scene_py

pytestmark = pytest.mark.gpu


def same_records(a, b):
    """Field-by-field equality of two structured arrays (padding bytes are not
    part of the numpy dtypes and are not compared)."""
    assert a.dtype == b.dtype and a.shape == b.shape
    for name in a.dtype.names:
        x, y = a[name], b[name]
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                              y.view(np.uint32) if y.dtype == np.float32 else y), name


def make_inputs(mrt, sc, n_side):
    st = sc.struct
    D = mrt.CL_DTYPES
    n = n_side * n_side
    pos = np.zeros(n, D["float3"])
    i = np.arange(n)
    pos["s"][:, 0] = (i % n_side) + 1.0
    pos["s"][:, 1] = (i // n_side) + 1.0
    vp = np.zeros(1, D["vp"])
    vp["x_resolution"] = vp["y_resolution"] = n_side
    vp["left"], vp["right"], vp["top"], vp["bottom"] = st.left, st.right, st.top, st.bottom
    cam = np.zeros(1, D["cam"])
    for k in ("eye", "u", "v", "w"):
        cam[k][0, :3] = list(getattr(st, k))
    objs = np.zeros(sc.n_objects, D["object"])
    ns = st.n_spheres
    for j in range(ns):
        m = sc.sph_mat[j]
        objs["center"][j, :3] = sc.sph[:3, j]
        objs["radius"][j] = sc.sph[3, j]
        objs["sph_color"][j, :3] = sc.mat[:3, m]
        objs["sph_shininess"][j] = sc.mat[3, m]
        objs["is_circle"][j] = 1
    for j in range(st.n_tris):
        m = sc.tri_mat[j]
        objs["a"][ns + j, :3] = sc.tri[0:3, j]
        objs["b"][ns + j, :3] = sc.tri[3:6, j]
        objs["c"][ns + j, :3] = sc.tri[6:9, j]
        objs["tri_color"][ns + j, :3] = sc.mat[:3, m]
        objs["tri_shininess"][ns + j] = sc.mat[3, m]
        objs["is_triangle"][ns + j] = 1
    lights = np.zeros(st.n_lights, D["light"])
    for l in range(st.n_lights):
        lights["location"][l, :3] = sc.light[:3, l]
        lights["color"][l, :3] = sc.light[3:, l]
        lights["intensity"][l] = 1.0
    phong = np.zeros(1, D["phong"])
    phong["ambient_coef"], phong["diffuse_coef"], phong["specular_coef"] = st.ka, st.kd, st.ks
    phong["ambient_color"][0, :3] = list(st.ambient)
    return pos, vp, cam, objs, lights, phong


def frame_rays(mrt, ctx, sc):
    """The primary rays of scene `sc`'s frame (xres x yres, its viewport and camera) as the array of
    rays the stage entry points pass around, made by the uv and view-ray stages on the GPU, and the
    scene's objects in the reference's records."""
    st = sc.struct
    D = mrt.CL_DTYPES
    n = st.xres * st.yres
    i = np.arange(n)
    pos = np.zeros(n, D["float3"])
    pos["s"][:, 0] = (i % st.xres) + 1.0
    pos["s"][:, 1] = (i // st.xres) + 1.0
    vp = np.zeros(1, D["vp"])
    vp["x_resolution"], vp["y_resolution"] = st.xres, st.yres
    vp["left"], vp["right"], vp["top"], vp["bottom"] = st.left, st.right, st.top, st.bottom
    _, _, cam, objs, _, _ = make_inputs(mrt, sc, 2)
    focal = None if st.projection == mrt.PROJ_ORTHO else st.focal
    return ctx.stage_viewrays(cam, ctx.stage_uv(pos, vp), focal), objs


# (the two 400x400 cases: 160 000 elements, enough for the host to scatter the results on several threads)
@pytest.mark.parametrize("name,n_side", [("scene_2spheres_persp_phong", 150), ("scene_gtri_ortho_phong", 150),
                                         ("teatime", 96), ("scene_2spheres_persp_phong", 400),
                                         ("scene_gtri_ortho_flat", 400)])
def test_stage_pipeline_matches_oracle(gpu_ctx, mrt, oracle, name, n_side):
    sc = scene_py.load_scene(SCENES / f"{name}.json")
    st = sc.struct
    pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, n_side)
    FLT_MAX = np.finfo(np.float32).max

    uv_g = gpu_ctx.stage_uv(pos, vp)
    uv_o = oracle.stage_uv(pos, vp)
    assert np.array_equal(uv_g["s"], uv_o["s"])

    focal = None if st.projection == mrt.PROJ_ORTHO else st.focal
    rays_g = gpu_ctx.stage_viewrays(cam, uv_g, focal)
    rays_o = oracle.stage_viewrays(cam, uv_o, focal)
    same_records(rays_g, rays_o)

    xs_g = gpu_ctx.stage_intersect(objs, rays_g, FLT_MAX, 0.0)
    xs_o, ids_o = oracle.stage_intersect(objs, rays_o, FLT_MAX, 0.0)
    assert (ids_o >= 0).any()
    same_records(xs_g, xs_o)
    # ... and into an array whose pages are resident (the call above got a fresh one: the library
    # fills an untouched array of this size on several host threads, a resident one on the caller's)
    same_records(gpu_ctx.stage_intersect(objs, rays_g, FLT_MAX, 0.0, out=xs_g.copy()), xs_o)
    same_records(gpu_ctx.stage_viewrays(cam, uv_g, focal, out=rays_g.copy()), rays_o)

    bg = tuple(st.background) + (0.5,)
    flat_g = gpu_ctx.stage_flat(xs_g, bg)
    flat_o = oracle.stage_flat(xs_o, bg)
    same_records(flat_g, flat_o)

    ph_g = gpu_ctx.stage_phong(xs_g, phong, bg, cam, lights)
    ph_o = oracle.stage_phong(xs_o, phong, bg, cam, lights)
    for c in "rgb":
        assert np.allclose(ph_g[c], ph_o[c], rtol=0, atol=2e-6)
    assert np.array_equal(ph_g["a"], ph_o["a"])     # background.a on a miss, untouched on a hit


def test_stage_viewrays_keep_the_padding_floats(gpu_ctx, mrt, oracle):
    """viewRay holds two cl_float3 (CLStructs.hpp): the kernels assign three components of each
    (kernel.cl:130-147), so the fourth float of origin and direction keeps what the caller's array
    held -- in the library, which does not upload that array, as in the oracle."""
    for name in ("scene_2spheres_persp_flat", "scene_2spheres_ortho_flat"):
        sc = scene_py.load_scene(SCENES / f"{name}.json")
        pos, vp, cam, _, _, _ = make_inputs(mrt, sc, 48)
        uv = gpu_ctx.stage_uv(pos, vp)
        focal = None if sc.struct.projection == mrt.PROJ_ORTHO else sc.struct.focal
        pre = np.zeros(len(uv), mrt.CL_DTYPES["ray"])
        raw = pre.view(np.uint8).reshape(len(pre), 32)
        raw[:, 12:16] = 0xC3
        raw[:, 28:32] = 0x3C
        g = gpu_ctx.stage_viewrays(cam, uv, focal, out=pre.view(np.uint8).copy().view(pre.dtype))
        o = oracle.stage_viewrays(cam, uv, focal, out=pre.view(np.uint8).copy().view(pre.dtype))
        graw, oraw = (a.view(np.uint8).reshape(len(a), 32) for a in (g, o))
        assert np.array_equal(graw, oraw)
        assert (graw[:, 12:16] == 0xC3).all() and (graw[:, 28:32] == 0x3C).all()
        assert graw[:, 16:28].any()                                     # ... and the directions were written


def test_stage_intersect_bounds_and_untouched_bytes(gpu_ctx, mrt, oracle):
    """t range is (lower, upper) for the loop, a miss writes three fields only."""
    sc = scene_py.load_scene(SCENES / "scene_2spheres_persp_flat.json")
    pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 64)
    uv = gpu_ctx.stage_uv(pos, vp)
    rays = gpu_ctx.stage_viewrays(cam, uv, sc.struct.focal)
    pre = np.zeros(len(rays), mrt.CL_DTYPES["intersect"])
    pre["location"][:] = 7.0
    pre["dummy2"][:] = 9.0
    # bytes no named field covers (dummy3 at 41..56, dummy1 at 59) and lane 3 of the float3 fields
    raw = pre.view(np.uint8).reshape(len(pre), 80)
    raw[:, 41:57] = 0xA5
    raw[:, 59] = 0x5A
    raw[:, 12:16] = 0xC3
    raw[:, 28:32] = 0x3C
    raw[:, 76:80] = 0x99
    def raw_copy(a):
        # ndarray.copy() of a structured array copies field by field and leaves the gaps between
        # fields undefined; a copy of the bytes keeps them
        return a.view(np.uint8).copy().view(a.dtype)

    for hi, lo in [(np.finfo(np.float32).max, 0.0), (3.0, 0.0), (100.0, 2.0), (1.0, 0.0)]:
        g = gpu_ctx.stage_intersect(objs, rays, hi, lo, out=raw_copy(pre))
        o, _ = oracle.stage_intersect(objs, rays, hi, lo, out=raw_copy(pre))
        same_records(g, o)
        # kernel.cl:361-377 assigns named fields only: everything else keeps the caller's bytes, on
        # hits and on misses, so the whole 80-byte records agree too
        graw, oraw = (a.view(np.uint8).reshape(len(a), 80) for a in (g, o))
        for lo_b, hi_b in ((41, 57), (59, 60), (12, 16), (28, 32), (76, 80)):
            assert np.array_equal(graw[:, lo_b:hi_b], raw[:, lo_b:hi_b]), (lo_b, "library")
            assert np.array_equal(oraw[:, lo_b:hi_b], raw[:, lo_b:hi_b]), (lo_b, "oracle")
        assert np.array_equal(graw, oraw)
    miss = g["intersects"] == 0
    assert miss.any() and (g["location"][miss][:, :3] == 7.0).all()


def test_stage_zero_elements_and_errors(gpu_ctx, mrt):
    D = mrt.CL_DTYPES
    assert len(gpu_ctx.stage_uv(np.zeros(0, D["float3"]), np.zeros(1, D["vp"]))) == 0
    with pytest.raises(mrt.MrtError):
        mrt.lib().mrt_stage_uv.argtypes  # keep flake quiet
        gpu_ctx._check(mrt.lib().mrt_stage_uv(gpu_ctx._h, None, None, 5, None))


def oracle_intersect_threads(oracle, objs, rays, hi, lo, out=None, threads=16):
    """oracle.stage_intersect over chunks of the rays on host threads (the C call drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(rays)
    res = np.zeros(n, oracle.mrt.CL_DTYPES["intersect"]) if out is None else out
    ids = np.zeros(n, np.int32)
    step = max(1, (n + threads - 1) // threads)

    def run(a):
        b = min(n, a + step)
        _, i = oracle.stage_intersect(objs, rays[a:b], hi, lo, out=res[a:b])
        ids[a:b] = i

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, range(0, n, step)))
    return res, ids


def random_objects(mrt, rng, n_tri, n_sph, span=3.0, tri_size=0.6):
    """Triangles and spheres interleaved in random order (the reference's object array may hold
    them in any order: kernel.cl:322 branches per object)."""
    D = mrt.CL_DTYPES
    n = n_tri + n_sph
    objs = np.zeros(n, D["object"])
    kind = np.zeros(n, bool)
    kind[rng.permutation(n)[:n_tri]] = True
    for j in range(n):
        c = rng.uniform(-span, span, 3).astype(np.float32)
        if kind[j]:
            for key in ("a", "b", "c"):
                objs[key][j, :3] = c + rng.uniform(-tri_size, tri_size, 3).astype(np.float32)
            objs["tri_color"][j, :3] = rng.uniform(0, 1, 3)
            objs["tri_shininess"][j] = float(2 ** rng.integers(1, 6))
            objs["is_triangle"][j] = 1
        else:
            objs["center"][j, :3] = c
            objs["radius"][j] = rng.uniform(0.05, 0.5)
            objs["sph_color"][j, :3] = rng.uniform(0, 1, 3)
            objs["sph_shininess"][j] = float(2 ** rng.integers(1, 6))
            objs["is_circle"][j] = 1
    return objs


def random_rays(mrt, rng, n, span=3.0):
    """Rays with origins all over the scene and directions all over the sphere, lengths from
    0.01 to 100: nothing a shared origin or direction could be precomputed from."""
    rays = np.zeros(n, mrt.CL_DTYPES["ray"])
    rays["origin"][:, :3] = rng.uniform(-span, span, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"][:, :3] = d * 10.0 ** rng.uniform(-2, 2, (n, 1))
    return rays


@pytest.mark.parametrize("seed,n_tri,n_sph,n_rays", [(1, 700, 60, 20000), (2, 33, 0, 5000), (3, 0, 40, 3000),
                                                    (4, 1500, 300, 4097), (5, 1, 1, 1)])
def test_stage_intersect_arbitrary_rays(gpu_ctx, mrt, oracle, seed, n_tri, n_sph, n_rays):
    """Every ray with its own origin and direction, objects in mixed order, every t range the
    loop of kernel.cl:305-379 can be given -- among them negative lower bounds, where hits behind
    the origin count -- bit for bit against the oracle."""
    rng = np.random.default_rng(seed)
    objs = random_objects(mrt, rng, n_tri, n_sph)
    rays = random_rays(mrt, rng, n_rays)
    # duplicates of an object further down the array: equal t must go to the lower index
    if len(objs) > 8:
        objs[-1] = objs[2]
        objs[-3] = objs[5]
    FLT_MAX = float(np.finfo(np.float32).max)
    for hi, lo in [(FLT_MAX, 0.0), (2.5, 0.0), (FLT_MAX, -4.0), (1.0, -1.0), (np.inf, 0.25), (0.5, 1.0)]:
        g = gpu_ctx.stage_intersect(objs, rays, hi, lo)
        o, ids = oracle.stage_intersect(objs, rays, hi, lo)
        same_records(g, o)
    assert gpu_ctx.stage_kernel_ms() > 0.0


def stacked_sheets(mrt, n_tri, z_lo=1.0, z_hi=4.0):
    """n_tri large triangles, one above the other, each covering the square |x|, |y| <= 2: every
    line through that square parallel to z meets every one of them."""
    objs = np.zeros(n_tri, mrt.CL_DTYPES["object"])
    z = np.linspace(z_lo, z_hi, n_tri).astype(np.float32)
    objs["a"][:, :3] = np.stack([np.full(n_tri, -9.0), np.full(n_tri, -6.0), z], axis=1)
    objs["b"][:, :3] = np.stack([np.full(n_tri, 9.0), np.full(n_tri, -6.0), z], axis=1)
    objs["c"][:, :3] = np.stack([np.full(n_tri, 0.0), np.full(n_tri, 12.0), z], axis=1)
    objs["tri_color"][:, :3] = 0.5
    objs["tri_shininess"][:] = 8.0
    objs["is_triangle"][:] = 1
    return objs


def test_stage_intersect_takes_up_the_fourth_determinant(gpu_ctx, mrt, oracle):
    """The kernel's filter starts without the determinant that carries the sign of t and takes it
    up, wave by wave, only where lines meet many triangles behind their rays.  Rays above a stack
    of sheets that look away from it (every sheet behind every ray: the waves must switch), rays
    that look at it (every sheet a hit: nothing to gain, no harm), both mixed, and the camera's
    primary rays on the teatime mesh (no wave switches): the records are the oracle's whichever
    filter ran, for t ranges on either side of zero."""
    rng = np.random.default_rng(11)
    objs = stacked_sheets(mrt, 600)
    n = 6000
    FLT_MAX = float(np.finfo(np.float32).max)

    def rays_from_above(sign):
        rays = np.zeros(n, mrt.CL_DTYPES["ray"])
        rays["origin"][:, 0:2] = rng.uniform(-2, 2, (n, 2))
        rays["origin"][:, 2] = 5.0
        rays["direction"][:, 0:2] = rng.uniform(-0.01, 0.01, (n, 2))
        rays["direction"][:, 2] = sign if np.isscalar(sign) else sign
        return rays

    away, toward = rays_from_above(1.0), rays_from_above(-1.0)
    mixed = rays_from_above(np.where(rng.random(n) < 0.5, 1.0, -1.0).astype(np.float32))
    for name, rays, must_switch in (("away", away, True), ("toward", toward, None), ("mixed", mixed, None)):
        for hi, lo in [(FLT_MAX, 0.0), (2.0, 0.5), (FLT_MAX, -10.0)]:
            g = gpu_ctx.stage_intersect(objs, rays, hi, lo)
            switched = gpu_ctx.stage_t_switches()
            o, ids = oracle.stage_intersect(objs, rays, hi, lo)
            same_records(g, o)
            if lo < 0:
                assert switched == 0, (name, hi, lo)          # t may be negative: the determinant says nothing
                assert (ids >= 0).all()
            elif must_switch:
                assert switched > 0, (name, hi, lo)
                assert (ids < 0).all()
    sc = scene_py.load_scene(SCENES / "teatime.json")
    pos, vp, cam, tobjs, _, _ = make_inputs(mrt, sc, 128)
    prim = gpu_ctx.stage_viewrays(cam, gpu_ctx.stage_uv(pos, vp), sc.struct.focal)
    g = gpu_ctx.stage_intersect(tobjs, prim, FLT_MAX, 0.0)
    assert gpu_ctx.stage_t_switches() == 0
    o, ids = oracle_intersect_threads(oracle, tobjs, prim, FLT_MAX, 0.0)
    same_records(g, o)


def test_stage_intersect_wild_rays_and_odd_objects(gpu_ctx, mrt, oracle):
    """Non-finite and huge ray components, rays of zero direction, degenerate and enormous
    triangles, spheres of zero / negative / NaN radius: whatever the reference arithmetic makes of
    them (kernel.cl:197-303), the library makes the same."""
    rng = np.random.default_rng(11)
    objs = random_objects(mrt, rng, 300, 40)
    objs["b"][3] = objs["a"][3]                        # two equal vertices
    objs["c"][4] = objs["a"][4]; objs["b"][4] = objs["a"][4]
    tri = np.flatnonzero(objs["is_triangle"])
    sph = np.flatnonzero(objs["is_circle"])
    objs["a"][tri[10], :3] *= 1e30                     # products overflow
    objs["b"][tri[11], 0] = np.inf
    objs["c"][tri[12], 1] = np.nan
    objs["a"][tri[13], :3] = 1e-30                     # products in the denormal range
    objs["b"][tri[13], :3] = (2e-30, 1e-30, 1e-30)
    objs["c"][tri[13], :3] = (1e-30, 2e-30, 1e-30)
    objs["radius"][sph[0]] = 0.0
    objs["radius"][sph[1]] = -0.3
    objs["radius"][sph[2]] = np.nan
    objs["center"][sph[3], 0] = np.inf
    rays = random_rays(mrt, rng, 6000)
    rays["origin"][5, 0] = np.nan
    rays["direction"][6, 1] = np.nan
    rays["direction"][7, 2] = np.inf
    rays["origin"][8, 1] = -np.inf
    rays["origin"][9, :3] = 1e30
    rays["direction"][10, :3] = (1e25, -1e25, 3e24)
    rays["direction"][11, :3] = 0.0
    rays["direction"][12, :3] = (1e-30, 2e-30, -1e-30)
    rays["origin"][13, :3] = (3e19, 0.0, 0.0)          # beyond the filter's range, finite
    rays["direction"][13, :3] = (-1.0, 0.0, 0.0)
    FLT_MAX = float(np.finfo(np.float32).max)
    for hi, lo in [(FLT_MAX, 0.0), (np.inf, -np.inf), (FLT_MAX, -1.0), (np.nan, 0.0)]:
        g = gpu_ctx.stage_intersect(objs, rays, hi, lo)
        o, _ = oracle.stage_intersect(objs, rays, hi, lo)
        same_records(g, o)


def test_stage_intersect_native_frame(gpu_ctx, mrt, oracle):
    """The reference's own workload: teatime at 400x400, 160 000 rays x 4034 objects, every record
    against the oracle; and the device time of the call's kernels."""
    sc = scene_py.load_scene(SCENES / "teatime.json")
    pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 400)
    uv = gpu_ctx.stage_uv(pos, vp)
    rays = gpu_ctx.stage_viewrays(cam, uv, sc.struct.focal)
    FLT_MAX = float(np.finfo(np.float32).max)
    g = gpu_ctx.stage_intersect(objs, rays, FLT_MAX, 0.0)
    ms = gpu_ctx.stage_kernel_ms()
    o, ids = oracle_intersect_threads(oracle, objs, rays, FLT_MAX, 0.0)
    assert (ids >= 0).sum() > 10000
    same_records(g, o)
    print(f"stage_intersect 160000 x {len(objs)}: kernels {ms:.3f} ms, {160000 * len(objs) / ms * 1e-9:.2f}e12 tests/s")
    assert ms < 5.0


@pytest.mark.parametrize("kind", ["generic", "needle", "edge_on", "subpixel", "huge", "behind", "far", "tiny",
                                  "micro", "denormal", "giant"])
def test_stage_intersect_fuzz_families(gpu_ctx, mrt, oracle, kind):
    """The scene families that stress the error margins of the fused kernels (needles, slivers
    seen edge-on, primitives around the eye, coordinates at 1e-13 ... 1e11, scenes 3e4 units from
    the origin), through mrt_stage_intersect: the camera's own rays, and rays between random points
    of the scene, against the oracle bit for bit."""
    import zlib
    from test_gpu_prefilter_fuzz import _scene
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 7)
    FLT_MAX = float(np.finfo(np.float32).max)
    hits = 0
    for it in range(12):
        proj = mrt.PROJ_PERSP if it % 2 == 0 else mrt.PROJ_ORTHO
        sc = _scene(mrt, rng, kind, proj)
        st = sc.struct
        side = 64
        pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, side)
        uv = oracle.stage_uv(pos, vp)
        rays = oracle.stage_viewrays(cam, uv, None if proj == mrt.PROJ_ORTHO else st.focal)
        # rays from one random vertex towards another, scaled by anything from 0.01 to 100
        tri = sc.tri.T.reshape(-1, 3, 3)
        n_extra = 3000
        a = tri[rng.integers(0, len(tri), n_extra), rng.integers(0, 3, n_extra)]
        b = tri[rng.integers(0, len(tri), n_extra), rng.integers(0, 3, n_extra)]
        extra = np.zeros(n_extra, rays.dtype)
        jitter = rng.normal(size=(n_extra, 3)) * np.abs(b - a).max() * 0.05
        extra["origin"][:, :3] = a + jitter
        extra["direction"][:, :3] = (b - a) * 10.0 ** rng.uniform(-2, 2, (n_extra, 1))
        allrays = np.concatenate([rays, extra])
        for hi, lo in ((FLT_MAX, 0.0), (FLT_MAX, -FLT_MAX)):
            g = gpu_ctx.stage_intersect(objs, allrays, hi, lo)
            o, ids = oracle.stage_intersect(objs, allrays, hi, lo)
            same_records(g, o)
            hits += int((ids >= 0).sum())
    assert hits > 0


def test_stage_intersect_equals_the_fused_path_on_a_frame(gpu_ctx, mrt):
    """Two kernels that share nothing but the reference arithmetic for their survivors: the fused
    frame kernel (rays made in registers from the camera, edge functions in pixel space) and the
    arbitrary-ray kernel (rays from memory, determinants as dot products).  On a 1280x720 frame of
    the teatime mesh -- 921 600 rays x 4034 objects -- winners and t agree bit for bit."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, str(ROOT / "tools"))
    import mkscene
    sc = mrt.LoadedScene(text=mkscene.teatime(1280, 720)).copy()
    st = sc.struct
    gpu_ctx.set_scene(sc)
    _, ids, t, _ = gpu_ctx.render(want_ids=True)
    n = st.xres * st.yres
    rays, objs = frame_rays(mrt, gpu_ctx, sc)
    xs = gpu_ctx.stage_intersect(objs, rays, float(np.finfo(np.float32).max), 0.0)
    ms = gpu_ctx.stage_kernel_ms()
    hit = xs["intersects"].reshape(st.yres, st.xres) != 0
    assert np.array_equal(hit, ids >= 0) and hit.sum() > 50000
    assert np.array_equal(xs["t_"].reshape(st.yres, st.xres).view(np.uint32), t.view(np.uint32))
    # the winner's material follows the winner: colour of the record against the scene's
    tri = ids >= st.n_spheres
    assert np.array_equal(xs["is_triangle"].reshape(st.yres, st.xres)[hit] != 0, tri[hit])
    print(f"stage_intersect {n} rays x {len(objs)} objects: kernels {ms:.3f} ms, {n * len(objs) / ms * 1e-9:.2f}e12 tests/s")
