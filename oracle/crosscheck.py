"""Cross-check of the two CPU restatements -- TEST INFRASTRUCTURE ONLY.

Runs the FP32 operator-for-operator oracle (mrt_oracle.c) and the independent float64
model (specmodel.py: Moller-Trumbore, textbook quadratic) on the benchmark workloads --
the shipped teatime scene at its native 400 x 400 (all pixels), sampled rows of the
3840 x 2160 teatime frame and of the 16 384-triangle / 1024-sphere soup at 1024 x 1024,
the soup and teatime again under the orthographic camera -- and records how far they
disagree in tests/golden/oracle_vs_specmodel.json: winner mismatches (expected only where
FP32 rounding decides a primitive edge), pixels beyond 1 LSB and whether every one of
those lies on a primitive edge.  tests/test_oracle_kat.py re-runs a bounded subset and
holds it to the recorded counts.

The reference ships nothing executable or recorded for this path, so this narrows what
"parity unpinned" leaves open; it does not close it.

Run (about half an hour on 8 cores):  python oracle/crosscheck.py [case ...]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))
import oracle_py  # noqa: E402
import scene_py  # noqa: E402
import specmodel  # noqa: E402
import mkscene  # noqa: E402
from parity import edge_mask  # noqa: E402

OUT = ROOT / "tests" / "golden" / "oracle_vs_specmodel.json"


def soup_scene(xres, yres, projection):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        f.write(mkscene.soup(xres, yres, projection=projection))
        path = f.name
    try:
        return scene_py.load_scene(path)
    finally:
        Path(path).unlink()


def ortho_teatime(xres, yres):
    """teatime under the orthographic camera.  Its rays start on the plane through the world
    origin spanned by u and v (the eye is dropped, kernel.cl:130-138), which cuts the mesh in
    two: the viewport below frames the half that lies in front of that plane."""
    return scene_py.load_scene(ROOT / "scenes" / "teatime.json", xres=xres, yres=yres, projection="ortho",
                               viewport=(-270.0, -140.0, -50.0, -115.0))


def compare(sc, rows_list, chunk=32, threads=8):
    """Oracle vs model on the given row ranges; counts summed over the ranges."""
    tot = {"pixels": 0, "hit_pixels": 0, "id_mismatches": 0, "beyond_1lsb": 0, "beyond_1lsb_off_edge": 0,
           "beyond_1lsb_off_edge_on_triangles": 0, "max_rel_t_diff_same_winner": 0.0}
    n_spheres = sc.struct.n_spheres
    for (y0, y1) in rows_list:
        # one row above and below so that the edge mask sees the neighbours
        a0, a1 = max(0, y0 - 1), min(sc.struct.yres, y1 + 1)
        o = oracle_py.render(sc, rows=(a0, a1), threads=threads)
        s = specmodel.render(sc, chunk=chunk, rows=(a0, a1))
        oi, orgb, ot = o["ids"][a0:a1], o["rgba"][a0:a1], o["t"][a0:a1]
        em = edge_mask(oi)
        sl = slice(y0 - a0, y1 - a0)
        d = np.abs(s["rgba"][sl, :, :3].astype(np.int16) - orgb[sl, :, :3].astype(np.int16)).max(axis=-1)
        bad = d > 1
        same = (s["ids"][sl] == oi[sl]) & (oi[sl] >= 0)
        if same.any():
            rel = np.abs(s["t"][sl][same] - ot[sl][same].astype(np.float64)) / np.abs(s["t"][sl][same])
            tot["max_rel_t_diff_same_winner"] = max(tot["max_rel_t_diff_same_winner"], float(rel.max()))
        tot["pixels"] += int(d.size)
        tot["hit_pixels"] += int((oi[sl] >= 0).sum())
        tot["id_mismatches"] += int((s["ids"][sl] != oi[sl]).sum())
        tot["beyond_1lsb"] += int(bad.sum())
        tot["beyond_1lsb_off_edge"] += int((bad & ~em[sl]).sum())
        # (off an edge the two disagree only on spheres: FP32 b*b - a*c, kernel.cl:254, cancels)
        tot["beyond_1lsb_off_edge_on_triangles"] += int((bad & ~em[sl] & (oi[sl] >= n_spheres)).sum())
    return tot


def cases():
    native = scene_py.load_scene(ROOT / "scenes" / "teatime.json")
    yield "teatime_400x400_all_rows", native, [(0, 400)]
    t4k = scene_py.load_scene(ROOT / "scenes" / "teatime.json", xres=3840, yres=2160,
                              viewport=(-3840 / 2160, 3840 / 2160, 1.0, -1.0))
    yield "teatime_3840x2160_rows", t4k, [(y, y + 1) for y in (600, 800, 900, 1000, 1080, 1150, 1250, 1400, 1600, 1800)]
    yield "teatime_ortho_640x320_rows", ortho_teatime(640, 320), [(y, y + 1) for y in range(20, 320, 37)]
    soup = soup_scene(1024, 1024, "persp")
    yield "soup_1024x1024_rows", soup, [(y, y + 1) for y in (3, 130, 257, 384, 511, 640, 770, 900, 1020)]
    soup_o = soup_scene(1024, 1024, "ortho")
    yield "soup_ortho_1024x1024_rows", soup_o, [(y, y + 1) for y in (3, 130, 257, 384, 511, 640, 770, 900, 1020)]
    # an eighth of the benchmark frame: ten rows out of every eighty, top to bottom
    yield "teatime_3840x2160_row_blocks", t4k, [(y, y + 10) for y in range(0, 2160, 80)]
    # the two large benchmark frames at their own size: rows of teatime at 7680 x 4320 and of the soup at 16384^2
    t8k = scene_py.load_scene(ROOT / "scenes" / "teatime.json", xres=7680, yres=4320,
                              viewport=(-7680 / 4320, 7680 / 4320, 1.0, -1.0))
    yield "teatime_7680x4320_rows", t8k, [(y, y + 1) for y in (1300, 1700, 2000, 2160, 2400, 2800, 3200)]
    yield "soup_16384x16384_rows", soup_scene(16384, 16384, "persp"), [(y, y + 1) for y in (2049, 8191, 12001)]


if __name__ == "__main__":
    # with arguments: only the named cases, merged into the existing record
    only = set(sys.argv[1:])
    res = json.loads(OUT.read_text()) if (only and OUT.exists()) else {}
    for name, sc, rows in cases():
        if only and name not in only:
            continue
        t0 = time.time()
        r = compare(sc, rows)
        r["rows"] = rows if len(rows) > 1 else [list(rows[0])]
        r["seconds"] = round(time.time() - t0, 1)
        res[name] = r
        print(name, r, flush=True)
    OUT.write_text(json.dumps(res, indent=1) + "\n")
    print("wrote", OUT)
