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


@pytest.mark.parametrize("name,n_side", [("scene_2spheres_persp_phong", 150), ("scene_gtri_ortho_phong", 150),
                                         ("teatime", 96)])
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

    bg = tuple(st.background) + (0.5,)
    flat_g = gpu_ctx.stage_flat(xs_g, bg)
    flat_o = oracle.stage_flat(xs_o, bg)
    same_records(flat_g, flat_o)

    ph_g = gpu_ctx.stage_phong(xs_g, phong, bg, cam, lights)
    ph_o = oracle.stage_phong(xs_o, phong, bg, cam, lights)
    for c in "rgb":
        assert np.allclose(ph_g[c], ph_o[c], rtol=0, atol=2e-6)
    assert np.array_equal(ph_g["a"], ph_o["a"])     # background.a on a miss, untouched on a hit


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
