"""Pins the CPU oracle with analytic known-answer values.

The reference has no tests or golden images for this path; every number below
follows from the shipped scene files and the formulas of kernel.cl alone
(pixel centre offset +1/+0.5, kernel.cl:126-127 with gfxraytrace.hpp:850-852;
truncating byte conversion, gfximage.hpp:46-51).
"""
import math

import numpy as np
import pytest

from conftest import SCENES, GOLDEN, ALL_SCENES, SMALL_SCENES
import scene_py


def load(name, **kw):
    return scene_py.load_scene(SCENES / f"{name}.json", **kw)


def test_byte_conversion_truncates(oracle):
    # 0.2 -> 51, 0.9 -> 229, 0.4 -> 102, 0.7 -> 178, 1.0 -> 255, 0 -> 0
    for x, b in [(0.2, 51), (0.9, 229), (0.4, 102), (0.7, 178), (1.0, 255), (0.0, 0), (0.999, 254)]:
        assert oracle.to_byte(np.float32(x)) == b
    assert oracle.to_byte(float("nan")) == 0


def test_camera_basis(oracle):
    # eye 0, up (0,-1,0), view (0,0,1)  =>  u=(1,0,0) v=(0,-1,0) w=(0,0,-1)  (gfxraytrace.hpp:881-889)
    u, v, w = oracle.camera_basis([0, 0, 1], [0, -1, 0])
    assert np.allclose(u, [1, 0, 0]) and np.allclose(v, [0, -1, 0]) and np.allclose(w, [0, 0, -1])
    u2, v2, w2 = scene_py.camera_basis([0.0, 0.0, 1.0], [0.0, -1.0, 0.0])
    assert np.array_equal(u, u2) and np.array_equal(v, v2) and np.array_equal(w, w2)
    # teatime: view (0,-1,0), up (0,0,-1)
    u, v, w = oracle.camera_basis([0, -1, 0], [0, 0, -1])
    assert np.allclose(w, [0, 1, 0]) and np.allclose(u, [1, 0, 0]) and np.allclose(v, [0, 0, -1])


def test_uv_kat(oracle, mrt):
    # uv = -1 + 0.005*(x + 1.5): pixel (0,0) -> -0.9925, (399,399) -> 1.0025, (99,200) -> (-0.4975, 0.0075)
    pos = np.zeros(3, mrt.CL_DTYPES["float3"])
    pos["s"][:, :2] = [[1, 1], [400, 400], [100, 201]]   # host grid is (x+1, y+1)
    vp = np.zeros(1, mrt.CL_DTYPES["vp"])
    vp["x_resolution"] = vp["y_resolution"] = 400
    vp["left"], vp["right"], vp["top"], vp["bottom"] = -1, 1, 1, -1
    uv = oracle.stage_uv(pos, vp)["s"]
    want = np.array([[-0.9925, -0.9925], [1.0025, 1.0025], [-0.4975, 0.0075]], np.float64).astype(np.float32)
    assert np.array_equal(uv, want)


def test_flat_colours_and_hit_counts(oracle):
    r = oracle.render(load("scene_gtri_ortho_flat"))
    hit = r["ids"] >= 0
    # triangle area 0.18 in a 2x2 viewport at 400x400 -> 0.18 / 0.005^2 = 7200 pixels
    assert abs(int(hit.sum()) - 7200) <= 72
    assert (r["rgba"][hit] == [51, 229, 51, 255]).all()
    assert (r["rgba"][~hit] == [102, 102, 102, 255]).all()
    assert np.allclose(r["t"][hit], 2.0)          # every orthographic hit is at t = 2 exactly

    r = oracle.render(load("scene_gtri_persp_flat"))
    assert abs(int((r["ids"] >= 0).sum()) - 1800) <= 36    # projected area / 4

    r = oracle.render(load("scene_2spheres_ortho_flat"))
    hit = r["ids"] >= 0
    # two half discs of radius 100 px -> pi * 100^2
    assert abs(int(hit.sum()) - math.pi * 1e4) <= 0.01 * math.pi * 1e4
    assert (r["rgba"][r["ids"] == 0] == [255, 0, 0, 255]).all()
    assert (r["rgba"][r["ids"] == 1] == [0, 0, 255, 255]).all()
    assert (r["rgba"][~hit] == [178, 178, 229, 255]).all()
    assert r["ids"][200, 99] == -1                 # |-0.4975 + 1| = 0.5025 > r


def test_phong_kat(oracle):
    # scene_gtri_ortho_phong pixel (199,199): hand-computed (163, 255, 163)
    r = oracle.render(load("scene_gtri_ortho_phong"), want_rgbf=True)
    assert tuple(r["rgba"][199, 199][:3]) == (163, 255, 163)
    assert abs(r["rgbf"][199, 199][0] - 0.6413) < 5e-4 and r["rgbf"][199, 199][1] == 1.0
    # scene_2spheres_persp_*: pixel (99,200): d = (-0.4975, -0.0075, 1), t = 1.55447 on sphere 0
    r = oracle.render(load("scene_2spheres_persp_phong"), want_rgbf=True)
    assert r["ids"][200, 99] == 0
    assert abs(r["t"][200, 99] - 1.55447) < 1e-5
    a = 0.4975 ** 2 + 0.0075 ** 2 + 1
    b = (-0.4975) * 1 + 1 * (-2)                   # d . (o - c), o = 0, c = (-1, 0, 2)
    c = 1 + 4 - 0.25
    t = (-b - math.sqrt(b * b - a * c)) / a
    assert abs(r["t"][200, 99] - t) < 1e-6


def test_informational_reference_values(oracle):
    """The only numbers that ever came out of the reference itself (seen in exploratory runs before
    executing its code was ruled out; informational, not authoritative): 31 428 hit pixels for the
    two orthographic half-discs, 9 924 under the perspective camera, and the float colour
    (1, 0.430362, 0.430362) at pixel x = 99, y = 200 of scene_2spheres_persp_phong."""
    assert int((oracle.render(load("scene_2spheres_ortho_flat"))["ids"] >= 0).sum()) == 31428
    assert int((oracle.render(load("scene_2spheres_ortho_phong"))["ids"] >= 0).sum()) == 31428
    assert int((oracle.render(load("scene_2spheres_persp_flat"))["ids"] >= 0).sum()) == 9924
    r = oracle.render(load("scene_2spheres_persp_phong"), want_rgbf=True)
    assert int((r["ids"] >= 0).sum()) == 9924
    assert np.allclose(r["rgbf"][200, 99], [1.0, 0.430362, 0.430362], rtol=0, atol=5e-7)


def test_sphere_near_root_only(oracle, mrt):
    # camera inside a sphere: the near root is negative, so it is a miss (kernel.cl:262-281)
    sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=8, yres=8, left=-1, right=1,
                   top=1, bottom=-1, projection=mrt.PROJ_PERSP, focal=1.0, background=(0.1, 0.2, 0.3),
                   sph=[[0], [0], [0], [5.0]], sph_mat=[0], mat=[[1], [1], [1], [4]])
    r = oracle.render(sc)
    assert (r["ids"] == -1).all()


def test_ortho_origin_drops_eye(oracle, mrt):
    # kernel.cl:133-136: the orthographic origin ignores the eye.  A triangle
    # at z=2 is hit at t=2 wherever the eye is.
    kw = dict(u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=16, yres=16, left=-1, right=1, top=1, bottom=-1,
              projection=mrt.PROJ_ORTHO, background=(0, 0, 0),
              tri=[[-5], [5], [2], [5], [5], [2], [0], [-5], [2]], tri_mat=[0], mat=[[1], [1], [1], [4]])
    r0 = oracle.render(mrt.Scene(eye=[0, 0, 0], **kw))
    r1 = oracle.render(mrt.Scene(eye=[3, 4, -50], **kw))
    assert (r0["ids"] == 0).any()
    assert np.array_equal(r0["t"], r1["t"]) and np.array_equal(r0["ids"], r1["ids"])


def test_tie_goes_to_earlier_object(oracle, mrt):
    # two identical triangles: strict t < best keeps the first (kernel.cl:326)
    tri = [[-5, -5], [5, 5], [2, 2], [5, 5], [5, 5], [2, 2], [0, 0], [-5, -5], [2, 2]]
    sc = mrt.Scene(eye=[0, 0, 0], u=[1, 0, 0], v=[0, -1, 0], w=[0, 0, -1], xres=8, yres=8, left=-1, right=1,
                   top=1, bottom=-1, projection=mrt.PROJ_ORTHO, background=(0, 0, 0), tri=tri, tri_mat=[0, 1],
                   mat=[[1, 0], [0, 1], [0, 0], [4, 4]])
    r = oracle.render(sc)
    assert set(np.unique(r["ids"])) <= {-1, 0} and (r["ids"] == 0).any()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_golden_fixtures_match_oracle(oracle, name):
    g = np.load(GOLDEN / f"{name}.npz")
    r = oracle.render(load(name), threads=4)
    assert np.array_equal(r["rgba"], g["rgba"])
    assert np.array_equal(r["ids"], g["ids"])
    assert np.array_equal(r["t"], g["t"])


@pytest.mark.parametrize("name", SMALL_SCENES)
def test_oracle_agrees_with_float64_model(oracle, name):
    import specmodel
    from parity import check
    sc = load(name)
    r = oracle.render(sc)
    s = specmodel.render(sc)
    check(s["rgba"], r["rgba"], r["ids"])
    mism = r["ids"] != s["ids"]
    assert mism.sum() <= 0.001 * mism.size


def test_oracle_agrees_with_float64_model_teatime_crop(oracle):
    # the full 400x400 teatime comparison takes minutes in numpy; a 200x200
    # render of the same view keeps the mesh, both spheres and all 4032 triangles.
    import specmodel
    from parity import check
    sc = load("teatime", xres=200, yres=200)
    r = oracle.render(sc, threads=4)
    s = specmodel.render(sc, chunk=32)
    check(s["rgba"], r["rgba"], r["ids"])
    assert (r["ids"] != s["ids"]).sum() <= 0.001 * r["ids"].size


def test_oracle_vs_float64_model_on_benchmark_workloads(oracle):
    """oracle/crosscheck.py compared the two restatements on the benchmark workloads (native
    teatime in full, rows of the 4K teatime frame, the soup and both scenes under the orthographic
    camera) and recorded the disagreements in tests/golden/oracle_vs_specmodel.json.  The record must
    show agreement (winner mismatches only at the FP32-vs-FP64 edge level, no pixel beyond 1 LSB
    off a primitive edge), and two of its sampled rows are recomputed here to hold it to account."""
    import json
    import crosscheck
    rec = json.loads((GOLDEN / "oracle_vs_specmodel.json").read_text())
    assert set(rec) >= {"teatime_400x400_all_rows", "teatime_3840x2160_rows", "teatime_ortho_640x320_rows",
                        "soup_1024x1024_rows", "soup_ortho_1024x1024_rows"}
    for name, r in rec.items():
        assert r["hit_pixels"] > 0.03 * r["pixels"], name
        assert r["id_mismatches"] <= 0.0005 * r["pixels"], name
        # FP32 vs FP64 colours part by more than 1 LSB only where FP32 is ill-conditioned: at primitive
        # edges and on the soup's small spheres (their limbs under the orthographic camera, the whole
        # disc at 16384^2, where a sphere of radius 0.02 at distance 4 is 80 pixels wide): b*b - a*c
        # (kernel.cl:254) cancels to 1e-2 .. 1e-3 relative and moves t by 1e-4, hence the normal
        # (loc - c) / r by half a percent and a specular term of exponent 32 by several bytes.  That is
        # the conditioning of the reference's own FP32 formula, which the oracle follows operation for
        # operation; on triangles the two restatements never part by more than 1 LSB off an edge
        assert r["beyond_1lsb"] <= 0.002 * r["pixels"], name
        assert r["beyond_1lsb_off_edge"] <= (3e-4 * r["pixels"] if "soup" in name else 0), name
        assert r.get("beyond_1lsb_off_edge_on_triangles", 0) == 0, name
        assert r["max_rel_t_diff_same_winner"] < 1e-3, name
    for name, sc, rows in crosscheck.cases():
        if name not in ("teatime_3840x2160_rows", "soup_1024x1024_rows"):
            continue
        again = crosscheck.compare(sc, rows[3:5], threads=4)
        assert again["beyond_1lsb_off_edge"] == 0
        assert again["id_mismatches"] <= rec[name]["id_mismatches"]
        assert again["beyond_1lsb"] <= rec[name]["beyond_1lsb"]


def test_threads_and_row_ranges_are_invariant(oracle):
    sc = load("scene_2spheres_persp_phong")
    full = oracle.render(sc, threads=1)
    par = oracle.render(sc, threads=7)
    assert np.array_equal(full["rgba"], par["rgba"])
    part = oracle.render(sc, rows=(100, 233), threads=3)
    assert np.array_equal(part["rgba"][100:233], full["rgba"][100:233])
    assert (part["rgba"][:100] == 0).all()


def test_contraction_latitude_of_the_reference(oracle):
    """An OpenCL compiler may fuse the a*b+c of kernel.cl:149-172 or not.  oracle/contraction.py
    rendered the shipped scenes and eighty rows of the 4K benchmark frame through the oracle's
    source built both ways and recorded what moves (tests/golden/oracle_contraction.json): t in the
    last bits on up to half of the hit pixels, a byte by 1 LSB on two pixels of a frame, never a
    winner and never hit against miss.  The record must say that the two readings meet the parity
    criterion against each other, and the shipped scenes are recomputed here to hold it to account."""
    import json
    import contraction
    from parity import check
    rec = json.loads((GOLDEN / "oracle_contraction.json").read_text())
    assert set(rec) == {f"{n}_400x400" for n in ALL_SCENES} | {"teatime_3840x2160_rows"}
    for name, r in rec.items():
        assert r["hit_pixels"] > 0.01 * r["pixels"], name
        assert r["winner_differs"] == 0 and r["hit_miss_differs"] == 0, name
        assert r["beyond_1lsb"] == 0 and r["max_byte_diff"] <= 1, name
        assert r["bytes_differ"] <= 1e-4 * r["pixels"], name
    assert rec["teatime_3840x2160_rows"]["pixels"] >= 8 * 3840
    assert sum(r["t_bits_differ"] for r in rec.values()) > 1000          # the two builds do differ
    for name in ALL_SCENES:
        sc = load(name)
        again = contraction.compare(sc, [(0, sc.struct.yres)], threads=4)
        for key in ("hit_pixels", "winner_differs", "hit_miss_differs", "t_bits_differ", "bytes_differ", "beyond_1lsb"):
            assert again[key] == rec[f"{name}_400x400"][key], (name, key)
        off = oracle.render(sc, threads=4)
        fma = oracle.render(sc, threads=4, contracted=True)
        check(fma["rgba"], off["rgba"], off["ids"])


def test_built_in_latitude_of_the_reference(oracle):
    """kernel.cl's divisions, sqrt, normalize(), dot() and powr() are not correctly rounded by rule:
    OpenCL 1.2 lets a device be off by 2.5 / 3 / 2 + n / a few / 16 ulp.  oracle/builtin_latitude.py
    rendered the shipped scenes and eighty rows of the 4K benchmark frame through the oracle's
    source with every such result moved by whole ulps inside those limits (pseudo-randomly, all up,
    all down) and recorded what moves (tests/golden/oracle_builtin_latitude.json): t in its last
    bits on most hit pixels, a byte by 1 LSB on at most ten pixels of a frame, never a winner and
    never hit against miss.  The record must say that every such device meets the parity criterion
    against the oracle, and shipped scenes are recomputed here to hold it to account."""
    import json
    import builtin_latitude
    from parity import check
    rec = json.loads((GOLDEN / "oracle_builtin_latitude.json").read_text())
    assert rec.pop("limits_ulp") == {"divide": 2, "sqrt": 3, "normalize_per_component": 5, "dot": 5, "powr": 16}
    assert set(rec) == {f"{n}_400x400" for n in ALL_SCENES} | {"teatime_3840x2160_rows"}
    for name, by_mode in rec.items():
        for mode in builtin_latitude.MODES:
            r = by_mode[mode]
            assert r["hit_pixels"] > 0.01 * r["pixels"], (name, mode)
            assert r["winner_differs"] == 0 and r["hit_miss_differs"] == 0 and r["differs_off_edge"] == 0, (name, mode)
            assert r["beyond_1lsb"] == 0 and r["max_byte_diff"] <= 1, (name, mode)
            assert r["bytes_differ"] <= 1e-4 * r["pixels"], (name, mode)
    assert rec["teatime_3840x2160_rows"]["random"]["pixels"] >= 8 * 3840
    assert sum(r[m]["t_bits_differ"] for r in rec.values() for m in builtin_latitude.MODES) > 10000   # the builds do differ
    for name in ALL_SCENES:
        sc = load(name)
        off = oracle.render(sc, threads=4)
        for mode in (("random",) if name == "teatime" else builtin_latitude.MODES):
            again = builtin_latitude.compare(sc, [(0, sc.struct.yres)], mode, threads=4)
            for key in builtin_latitude.KEYS:
                assert again[key] == rec[f"{name}_400x400"][mode][key], (name, mode, key)
            mov = oracle.render(sc, threads=4, ulp_mode=builtin_latitude.MODES[mode])
            check(mov["rgba"], off["rgba"], off["ids"])
