"""Adversarial check of the conservative prefilters: for many random scenes built to
stress the error margins (needles, slivers seen edge-on, sub-pixel and screen-filling
primitives, primitives straddling the eye plane, coordinates far from the origin,
extreme aspect ratios), rendering with the prefilters must give exactly the winners
and t of rendering with MRT_FLAG_NO_PREFILTER (exact test for every pair), which in
turn is held to the oracle by tests/test_gpu_parity.py."""
import zlib

import numpy as np
import pytest

import scene_py

pytestmark = pytest.mark.gpu


def _basis(rng):
    view = rng.normal(size=3)
    up = rng.normal(size=3)
    u, v, w = scene_py.camera_basis(list(view / np.linalg.norm(view)), list(up))
    return np.array(u), np.array(v), np.array(w)


def _scene(mrt, rng, kind, proj):
    xres, yres = int(rng.integers(33, 200)), int(rng.integers(33, 160))
    if kind == "grazing_hires":     # FP32 noise of the exact test, in pixels, grows with resolution
        xres, yres = int(rng.integers(3000, 8000)), int(rng.integers(24, 48))
    u, v, w = _basis(rng)
    scale = {"far": 1.0e4, "tiny": 1.0e-3, "micro": 1.0e-11, "denormal": 1.5e-13,
             "giant": 1.0e11}.get(kind, 1.0) * rng.uniform(0.5, 2.0)
    eye = rng.uniform(-3, 3, 3) * scale + (rng.uniform(-1, 1, 3) * 3.0e4 if kind == "far" else 0.0)
    focal = rng.uniform(0.3, 3.0)
    depth = rng.uniform(2.0, 8.0) * scale
    centre = (eye if proj == mrt.PROJ_PERSP else 0.0) - w * depth
    n = 160
    base = centre + (rng.uniform(-1, 1, (n, 3)) * depth * 0.6)
    if kind == "needle":            # two vertices almost coincident, third far away
        a = base
        b = a + rng.normal(size=(n, 3)) * 1e-4 * depth
        c = a + rng.normal(size=(n, 3)) * depth * 0.5
    elif kind == "edge_on":         # triangles in planes that (almost) contain the eye / ray direction
        a = base
        dirv = (a - eye) if proj == mrt.PROJ_PERSP else np.broadcast_to(-w, a.shape)
        b = a + dirv * rng.uniform(0.05, 0.5, (n, 1)) + rng.normal(size=(n, 3)) * 1e-5 * depth
        c = a + rng.normal(size=(n, 3)) * depth * 0.2
    elif kind == "grazing_hires":   # long triangles seen at 0.01 .. 1 degree from edge-on
        a = base
        dirv = (a - eye) if proj == mrt.PROJ_PERSP else np.broadcast_to(-w * depth, a.shape)
        side = np.cross(dirv, rng.normal(size=(n, 3)))
        side /= np.linalg.norm(side, axis=1, keepdims=True)
        tilt = np.cross(dirv, side)
        tilt /= np.linalg.norm(tilt, axis=1, keepdims=True)
        ang = 10.0 ** rng.uniform(-4, -1.7, (n, 1))
        b = a + dirv * rng.uniform(0.1, 0.4, (n, 1)) + tilt * ang * depth
        c = a + side * depth * rng.uniform(0.05, 0.4, (n, 1))
    elif kind == "subpixel":
        a = base
        b = a + rng.normal(size=(n, 3)) * depth * 2e-4
        c = a + rng.normal(size=(n, 3)) * depth * 2e-4
    elif kind == "huge":
        a = base + rng.normal(size=(n, 3)) * depth * 5
        b = base + rng.normal(size=(n, 3)) * depth * 5
        c = base + rng.normal(size=(n, 3)) * depth * 5
    elif kind == "behind":          # spread around the eye so many straddle the eye plane
        a = eye + rng.normal(size=(n, 3)) * depth
        b = eye + rng.normal(size=(n, 3)) * depth
        c = eye + rng.normal(size=(n, 3)) * depth
    else:                           # generic / far / tiny / micro / denormal / giant
        a = base
        b = a + rng.normal(size=(n, 3)) * depth * 0.15
        c = a + rng.normal(size=(n, 3)) * depth * 0.15
    tri = np.concatenate([a.T, b.T, c.T], axis=0)
    ns = 24
    sc_ = centre + rng.uniform(-1, 1, (ns, 3)) * depth * 0.6
    if kind == "behind":
        sc_ = eye + rng.normal(size=(ns, 3)) * depth * 0.5
    rad = np.abs(rng.normal(size=ns)) * depth * {"subpixel": 3e-4, "huge": 2.0}.get(kind, 0.08) + 1e-9 * depth
    sph = np.concatenate([sc_.T, rad[None, :]], axis=0)
    half_w = rng.uniform(0.2, 2.0) * (scale if proj == mrt.PROJ_ORTHO else 1.0) * (depth / scale if proj == mrt.PROJ_ORTHO else 1.0)
    half_h = half_w * rng.uniform(0.3, 3.0)
    return mrt.Scene(eye=eye, u=u, v=v, w=w, xres=xres, yres=yres, left=-half_w * rng.uniform(0.5, 1.5),
                     right=half_w, top=half_h, bottom=-half_h * rng.uniform(0.5, 1.5), projection=proj,
                     focal=focal, shader=mrt.SHADER_FLAT, background=(0, 0, 0), sph=sph,
                     sph_mat=np.zeros(ns, np.uint32), tri=tri, tri_mat=np.zeros(n, np.uint32),
                     mat=[[1.0], [0.5], [0.25], [4.0]])


@pytest.mark.parametrize("kind", ["generic", "needle", "edge_on", "subpixel", "huge", "behind", "far", "tiny",
                                  "grazing_hires", "micro", "denormal", "giant"])
def test_prefilters_never_drop_an_exact_hit(mrt, kind):
    fast = mrt.Context(device=0)
    wide = mrt.Context(device=0, flags=mrt.FLAG_TRACE_WIDE)
    blk = mrt.Accel(device=0, method="block")
    cull = mrt.Accel(device=0, method="cull")
    grid = mrt.Accel(device=0, method="grid")
    full = mrt.Context(device=0, flags=mrt.FLAG_FULL_TEST)
    exact = mrt.Context(device=0, flags=mrt.FLAG_NO_PREFILTER)
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    try:
        hits = 0
        for it in range(16 if kind == "grazing_hires" else 60):
            proj = mrt.PROJ_PERSP if it % 2 == 0 else mrt.PROJ_ORTHO
            sc = _scene(mrt, rng, kind, proj)
            exact.set_scene(sc)
            _, ids_e, t_e, _ = exact.render(want_ids=True)
            for ctx, kernel in ((fast, mrt.KERNEL_TRACE), (wide, mrt.KERNEL_TRACE_WIDE), (blk, mrt.KERNEL_TRACE_BLOCK), (cull, mrt.KERNEL_CULL),
                                (grid, mrt.KERNEL_GRID), (full, mrt.KERNEL_FULL)):
                ctx.set_scene(sc)
                _, ids_f, t_f, st_f = ctx.render(want_ids=True)
                assert st_f.kernel == kernel       # any frame height gets the kernel its flags ask for
                bad = int((ids_f != ids_e).sum())
                assert bad == 0, (kind, it, proj, bad)
                assert np.array_equal(t_f.view(np.uint32), t_e.view(np.uint32)), (kind, it, proj)
            hits += int((ids_e >= 0).sum())
        assert hits > 0, kind
    finally:
        fast.close()
        wide.close()
        blk.close()
        cull.close()
        grid.close()
        full.close()
        exact.close()
