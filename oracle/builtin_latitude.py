"""How far may the reference's own output move with the device's built-ins?  -- TEST INFRASTRUCTURE ONLY.

kernel.cl divides (`/`, :187-195, :262-283), takes square roots (sqrt, :174-180, :262) and calls the
built-ins normalize() (:227), dot() (:423) and powr() (:426).  None of these is correctly rounded by
rule: the OpenCL 1.2 full profile (section 7.4 of the specification) lets a conforming device be off
by 2.5 ulp for x / y, 3 ulp for sqrt, 2 + n ulp for normalize of an n-vector and 16 ulp for powr, and
bounds dot() by an absolute error.  The oracle (libmrt_oracle.so) takes IEEE division and sqrt and
glibc's powf.  This script renders the same frames through the same source built with
-DMRT_ORACLE_ULP_LATITUDE (libmrt_oracle_ulp.so), which moves the result of every such operation by
a whole number of ulps inside those limits (division 2, sqrt 3, normalize 5 per component, dot 5,
powr 16) -- pseudo-randomly, all up, all down -- and records, per frame and mode, the pixels whose
winner, hit/miss state or bytes differ from the oracle's into
tests/golden/oracle_builtin_latitude.json: the nine shipped scenes at their native 400 x 400 in
full, and eighty rows of the 3840 x 2160 teatime benchmark frame.
tests/test_oracle_kat.py recomputes the shipped scenes and holds them to the record;
tests/test_gpu_parity.py holds the HIP frames to the parity criterion against every mode.

None of these builds is the reference; together with the contracted build (oracle/contraction.py)
they bracket what a conforming OpenCL device may produce from kernel.cl.

Run:  python oracle/builtin_latitude.py
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
from contraction import ROWS_4K, cases  # noqa: E402,F401  (same frames and rows as the contraction record)
from parity import edge_mask  # noqa: E402

OUT = ROOT / "tests" / "golden" / "oracle_builtin_latitude.json"
MODES = {"random": 0, "all_up": 1, "all_down": 2}
KEYS = ("pixels", "hit_pixels", "winner_differs", "hit_miss_differs", "t_bits_differ", "bytes_differ",
        "beyond_1lsb", "beyond_1lsb_off_edge", "differs_off_edge", "max_byte_diff")


def compare(sc, rows_list, mode, threads=8):
    """The oracle against the build whose built-ins are moved (mode: a key of MODES), over the given
    row ranges; counts summed over the ranges.  `differs_off_edge`: pixels whose winner differs and
    that have no 8-neighbour with another winner in the oracle's frame."""
    tot = dict.fromkeys(KEYS, 0)
    for r0, r1 in rows_list:
        a0, a1 = max(0, r0 - 1), min(sc.struct.yres, r1 + 1)     # a row of context for the edge mask
        ref = oracle_py.render(sc, rows=(a0, a1), threads=threads)
        mov = oracle_py.render(sc, rows=(a0, a1), threads=threads, ulp_mode=MODES[mode])
        edge = edge_mask(ref["ids"][a0:a1])[r0 - a0:r1 - a0]
        io, jm = ref["ids"][r0:r1], mov["ids"][r0:r1]
        d = np.abs(ref["rgba"][r0:r1, :, :3].astype(np.int16) - mov["rgba"][r0:r1, :, :3].astype(np.int16)).max(axis=-1)
        tot["pixels"] += int(io.size)
        tot["hit_pixels"] += int((io >= 0).sum())
        tot["winner_differs"] += int((io != jm).sum())
        tot["hit_miss_differs"] += int(((io >= 0) != (jm >= 0)).sum())
        tot["t_bits_differ"] += int((ref["t"][r0:r1].view(np.uint32) != mov["t"][r0:r1].view(np.uint32)).sum())
        tot["bytes_differ"] += int((d > 0).sum())
        tot["beyond_1lsb"] += int((d > 1).sum())
        tot["beyond_1lsb_off_edge"] += int(((d > 1) & ~edge).sum())
        tot["differs_off_edge"] += int(((io != jm) & ~edge).sum())
        tot["max_byte_diff"] = max(tot["max_byte_diff"], int(d.max()))
    return tot


def main():
    rec = {"limits_ulp": {"divide": 2, "sqrt": 3, "normalize_per_component": 5, "dot": 5, "powr": 16}}
    for name, sc, rows in cases():
        rec[name] = {mode: compare(sc, rows, mode) for mode in MODES}
        rec[name]["rows"] = [list(r) for r in rows]
        print(name, {m: {k: v for k, v in rec[name][m].items() if v and k not in ("pixels", "hit_pixels")} for m in MODES}, flush=True)
    OUT.write_text(json.dumps(rec, indent=1) + "\n")


if __name__ == "__main__":
    main()
