"""Independent scene reader for the tests -- TEST INFRASTRUCTURE ONLY.

Reads a (valid) scene file with Python's json module and applies the
conversions of gfx::scene::read_json (gfxraytrace.hpp:645-721, :727-808,
camera basis :881-889) so the C++ loader in the product can be checked against
a second implementation.  Validation / error strings are the C++ loader's job
and are tested there.
"""
from __future__ import annotations

import json
import math
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import mrt_pkg  # noqa: E402

mrt = mrt_pkg.load()


def camera_basis(view, up):
    vm = math.sqrt(view[0] * view[0] + view[1] * view[1] + view[2] * view[2])
    w = [-(view[i] / vm) for i in range(3)]
    c = [up[1] * w[2] - up[2] * w[1], up[2] * w[0] - up[0] * w[2], up[0] * w[1] - up[1] * w[0]]
    cm = math.sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])
    u = [c[i] / cm for i in range(3)]
    v = [w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]]
    return u, v, w


def scene_from_dict(j, xres=None, yres=None, viewport=None, projection=None):
    """xres / yres / viewport (left, right, top, bottom) / projection ("ortho" | "persp") override the file's."""
    eye = [float(x) for x in j["camera_eye"]]
    u, v, w = camera_basis([float(x) for x in j["camera_view"]], [float(x) for x in j["camera_up"]])
    mats = j["materials"]
    names = [m["name"] for m in mats]
    mat = np.array([[m["color"][0] for m in mats], [m["color"][1] for m in mats],
                    [m["color"][2] for m in mats], [m["shininess"] for m in mats]], np.float64)
    sph_l = j.get("spheres", [])
    tri_l = j.get("triangles", [])
    lig_l = j.get("point_lights", [])
    sph = np.array([[s["center"][0] for s in sph_l], [s["center"][1] for s in sph_l],
                    [s["center"][2] for s in sph_l], [s["radius"] for s in sph_l]], np.float64).reshape(4, -1)
    tri = np.array([[t[k][c] for t in tri_l] for k in "abc" for c in range(3)], np.float64).reshape(9, -1)
    light = np.array([[l["location"][c] for l in lig_l] for c in range(3)] +
                     [[l["color"][c] for l in lig_l] for c in range(3)], np.float64).reshape(6, -1)
    flat = "flat_shader" in j
    ph = j.get("phong_shader", {})
    left, right, top, bottom = (j["viewport_left"], j["viewport_right"], j["viewport_top"], j["viewport_bottom"])
    if viewport is not None:
        left, right, top, bottom = viewport
    return mrt.Scene(
        eye=eye, u=u, v=v, w=w,
        xres=j["x_resolution"] if xres is None else xres, yres=j["y_resolution"] if yres is None else yres,
        left=left, right=right, top=top, bottom=bottom,
        projection=((mrt.PROJ_ORTHO if "ortho_projection" in j else mrt.PROJ_PERSP) if projection is None
                    else {"ortho": mrt.PROJ_ORTHO, "persp": mrt.PROJ_PERSP}[projection]),
        focal=j.get("persp_focal_length", 1.0 if projection == "persp" else 0.0),
        shader=mrt.SHADER_FLAT if flat else mrt.SHADER_PHONG,
        ka=0.0 if flat else ph["ambient_coeff"], kd=0.0 if flat else ph["diffuse_coeff"],
        ks=0.0 if flat else ph["specular_coeff"],
        ambient=(0, 0, 0) if flat else ph["ambient_color"], background=j["background"],
        sph=sph, sph_mat=[names.index(s["material"]) for s in sph_l],
        tri=tri, tri_mat=[names.index(t["material"]) for t in tri_l], mat=mat, light=light)


def load_scene(path, **kw):
    with open(path) as f:
        return scene_from_dict(json.load(f), **kw)
