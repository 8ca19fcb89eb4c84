"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/mrt_oracle.c).

The reference ships no golden images and its code may not be executed here, so
these fixtures are outputs of this repository's own restatement of the
reference arithmetic, committed after that restatement agreed with the
analytic known-answer values in tests/test_oracle_kat.py and with the
independent float64 model oracle/specmodel.py.  Each file holds, for one
shipped scene at its native 400x400: rgba (uint8 [400,400,4]), ids (int32
winning object id, -1 = miss) and t (float32).

Run:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import oracle_py  # noqa: E402
import scene_py  # noqa: E402

SCENES = sorted(p.stem for p in (ROOT / "scenes").glob("*.json"))

if __name__ == "__main__":
    for name in SCENES:
        sc = scene_py.load_scene(ROOT / "scenes" / f"{name}.json")
        r = oracle_py.render(sc, threads=8)
        out = Path(__file__).with_name(f"{name}.npz")
        np.savez_compressed(out, rgba=r["rgba"], ids=r["ids"], t=r["t"])
        print(name, "hits", int((r["ids"] >= 0).sum()), out.stat().st_size, "bytes")
