"""Scene-file generators for the benchmark configurations.

  teatime variants   the shipped teatime.json with another resolution and a
                     viewport widened to keep pixels square
  soup               synthetic triangle + sphere soup, fully determined by a seed

Both return JSON text in the scene schema (rayson.hpp:380-638): keys read with
get_double are always printed with a decimal point, resolutions as integers.

CLI:  python tools/mkscene.py teatime 3840 2160 > teatime_4k.json
      python tools/mkscene.py soup 16384 16384 [n_tris n_spheres seed] > soup16k.json
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
SOUP_SEED = 20260711


def _f(x) -> str:
    """A JSON float token that keeps its float type (always has '.' or an exponent)."""
    s = repr(float(x))
    return s if any(c in s for c in ".eEn") else s + ".0"


def teatime(xres: int, yres: int) -> str:
    """teatime.json at xres x yres; top/bottom stay +-1, left/right = -+ xres/yres."""
    text = (ROOT / "scenes" / "teatime.json").read_text()
    aspect = xres / yres
    rep = {
        '"x_resolution" : 400': f'"x_resolution" : {xres}',
        '"y_resolution" : 400': f'"y_resolution" : {yres}',
        '"viewport_left" : -1.0': f'"viewport_left" : {_f(-aspect)}',
        '"viewport_right" : 1.0': f'"viewport_right" : {_f(aspect)}',
    }
    for k, v in rep.items():
        assert k in text, k
        text = text.replace(k, v)
    return text


class _Mt:
    """mt19937 seeded like std::mt19937(seed); raw 32-bit draws mapped to [0,1) as (x >> 8) * 2^-24."""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(seed)

    def raw(self) -> int:
        return int(self.bg.random_raw())

    def unit(self) -> float:
        return (self.raw() >> 8) * (1.0 / 16777216.0)

    def uniform(self, lo, hi) -> float:
        return lo + (hi - lo) * self.unit()


def soup(xres=16384, yres=16384, n_tris=16384, n_spheres=1024, seed=SOUP_SEED, n_lights=8,
         n_materials=8, projection="persp") -> str:
    """Camera at the origin looking down +z, focal 1, viewport +-1; Blinn-Phong 0.05/0.5/0.25.
    Draw order: materials, lights, spheres, triangles.  Float32-exact values are not required;
    the loader casts."""
    g = _Mt(seed)
    mats = []
    for k in range(n_materials):
        col = [g.unit(), g.unit(), g.unit()]
        shin = float(2 ** (1 + g.raw() % 5))
        mats.append((f"m{k}", col, shin))
    lights = [([g.uniform(-4, 4), g.uniform(-4, 4), g.uniform(-2, 0)], [1.0, 1.0, 1.0], 1.0)
              for _ in range(n_lights)]
    spheres = []
    for k in range(n_spheres):
        c = [g.uniform(-1, 1), g.uniform(-1, 1), g.uniform(2, 6)]
        spheres.append((k % n_materials, c, g.uniform(0.01, 0.03)))
    tris = []
    for k in range(n_tris):
        c = [g.uniform(-1, 1), g.uniform(-1, 1), g.uniform(2, 6)]
        while True:
            v = [[c[i] + g.uniform(-0.04, 0.04) for i in range(3)] for _ in range(3)]
            if v[0] != v[1] and v[0] != v[2] and v[1] != v[2]:
                break
        tris.append((k % n_materials, v))

    def vec(v):
        return "[" + ", ".join(_f(x) for x in v) + "]"
    out = ["{"]
    out.append('  "camera_eye" : [0.0, 0.0, 0.0],')
    out.append('  "camera_up" : [0.0, -1.0, 0.0],')
    out.append('  "camera_view" : [0.0, 0.0, 1.0],')
    out.append(f'  "x_resolution" : {int(xres)},')
    out.append(f'  "y_resolution" : {int(yres)},')
    out.append('  "viewport_left" : -1.0,')
    out.append('  "viewport_top" : 1.0,')
    out.append('  "viewport_right" : 1.0,')
    out.append('  "viewport_bottom" : -1.0,')
    out.append('  "background" : [0.1, 0.1, 0.15],')
    out.append('  "ortho_projection" : true,' if projection == "ortho" else '  "persp_focal_length" : 1.0,')
    out.append('  "phong_shader" : { "ambient_coeff" : 0.05, "diffuse_coeff" : 0.5, "specular_coeff" : 0.25, '
               '"ambient_color" : [1.0, 1.0, 1.0] },')
    out.append('  "materials" : [')
    out.append(",\n".join(f'    {{ "name" : "{n}", "color" : {vec(c)}, "shininess" : {_f(s)} }}' for n, c, s in mats))
    out.append("  ],")
    out.append('  "point_lights" : [')
    out.append(",\n".join(f'    {{ "location" : {vec(p)}, "intensity" : {_f(i)}, "color" : {vec(c)} }}'
                          for p, c, i in lights))
    out.append("  ],")
    out.append('  "spheres" : [')
    out.append(",\n".join(f'    {{ "material" : "m{m}", "center" : {vec(c)}, "radius" : {_f(r)} }}'
                          for m, c, r in spheres))
    out.append("  ],")
    out.append('  "triangles" : [')
    out.append(",\n".join(f'    {{ "material" : "m{m}", "a" : {vec(v[0])}, "b" : {vec(v[1])}, "c" : {vec(v[2])} }}'
                          for m, v in tris))
    out.append("  ]")
    out.append("}")
    text = "\n".join(out) + "\n"
    json.loads(text)   # well-formed
    return text


if __name__ == "__main__":
    kind = sys.argv[1]
    a = [int(x) for x in sys.argv[2:]]
    if kind == "teatime":
        sys.stdout.write(teatime(a[0], a[1]))
    elif kind == "soup":
        sys.stdout.write(soup(*a))
    else:
        sys.exit("usage: mkscene.py teatime X Y | soup X Y [n_tris n_spheres seed]")
