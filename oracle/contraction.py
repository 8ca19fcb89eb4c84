"""How far may the reference's own output move with the compiler?  -- TEST INFRASTRUCTURE ONLY.

An OpenCL compiler may fuse the a*b+c expressions of kernel.cl:149-172 (determinant3x3, cl_cross,
cl_dot) into multiply-adds, or not; the reference's pixels are device-dependent to that extent.
The oracle (libmrt_oracle.so) fixes the unfused reading.  This script renders the same frames
through the same source built with contraction on (libmrt_oracle_fma.so: gcc -ffp-contract=fast
-mfma) and records, per frame, the pixels whose winner, hit/miss state or bytes differ between the
two builds into tests/golden/oracle_contraction.json: the nine shipped scenes at their native
400 x 400 in full, and eighty rows of the 3840 x 2160 teatime benchmark frame.
tests/test_oracle_kat.py recomputes the shipped scenes and holds them to the record;
tests/test_gpu_parity.py holds the HIP frames to the parity criterion against BOTH builds.

Neither build is the reference; together they bracket what a conforming OpenCL device may produce
from kernel.cl as far as contraction goes (the precision of the built-ins -- division, sqrt,
normalize, dot, powr -- is a second, separate freedom: oracle/builtin_latitude.py).

Run:  python oracle/contraction.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))
import oracle_py  # noqa: E402
import scene_py  # noqa: E402
from parity import edge_mask  # noqa: E402

OUT = ROOT / "tests" / "golden" / "oracle_contraction.json"
SHIPPED = ["scene_2spheres_ortho_flat", "scene_2spheres_ortho_phong", "scene_2spheres_persp_flat",
           "scene_2spheres_persp_phong", "scene_gtri_ortho_flat", "scene_gtri_ortho_phong",
           "scene_gtri_persp_flat", "scene_gtri_persp_phong", "teatime"]
ROWS_4K = [(r, r + 2) for r in range(30, 2160, 54)]     # 80 rows across the frame


def compare(sc, rows_list, threads=8):
    """Unfused vs contracted build over the given row ranges; counts summed over the ranges."""
    tot = {"pixels": 0, "hit_pixels": 0, "winner_differs": 0, "hit_miss_differs": 0, "t_bits_differ": 0,
           "bytes_differ": 0, "beyond_1lsb": 0, "beyond_1lsb_off_edge": 0, "max_byte_diff": 0}
    for r0, r1 in rows_list:
        # one row of context either side, so that the edge mask of the first and last row is whole
        a0, a1 = max(0, r0 - 1), min(sc.struct.yres, r1 + 1)
        off = oracle_py.render(sc, rows=(a0, a1), threads=threads)
        fma = oracle_py.render(sc, rows=(a0, a1), threads=threads, contracted=True)
        edge = edge_mask(off["ids"][a0:a1])[r0 - a0:r1 - a0]
        io, jf = off["ids"][r0:r1], fma["ids"][r0:r1]
        d = np.abs(off["rgba"][r0:r1, :, :3].astype(np.int16) - fma["rgba"][r0:r1, :, :3].astype(np.int16)).max(axis=-1)
        tot["pixels"] += int(io.size)
        tot["hit_pixels"] += int((io >= 0).sum())
        tot["winner_differs"] += int((io != jf).sum())
        tot["hit_miss_differs"] += int(((io >= 0) != (jf >= 0)).sum())
        tot["t_bits_differ"] += int((off["t"][r0:r1].view(np.uint32) != fma["t"][r0:r1].view(np.uint32)).sum())
        tot["bytes_differ"] += int((d > 0).sum())
        tot["beyond_1lsb"] += int((d > 1).sum())
        tot["beyond_1lsb_off_edge"] += int(((d > 1) & ~edge).sum())
        tot["max_byte_diff"] = max(tot["max_byte_diff"], int(d.max()))
    tot["rows"] = [list(r) for r in rows_list]
    return tot


def cases():
    import mkscene
    for name in SHIPPED:
        sc = scene_py.load_scene(ROOT / "scenes" / f"{name}.json")
        yield f"{name}_400x400", sc, [(0, sc.struct.yres)]
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        f.write(mkscene.teatime(3840, 2160))
        path = f.name
    try:
        yield "teatime_3840x2160_rows", scene_py.load_scene(path), ROWS_4K
    finally:
        Path(path).unlink()


def main():
    rec = {}
    for name, sc, rows in cases():
        rec[name] = compare(sc, rows)
        print(name, rec[name], flush=True)
    OUT.write_text(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
