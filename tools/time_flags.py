#!/usr/bin/env python3
"""Kernel time of one scene under a list of mrt_config flag sets (GPU box only).

usage: time_flags.py [teatime|soup|<scene.json>] XRES YRES FLAGS [FLAGS ...]     (FLAGS are integers)
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402,F401  (one HIP runtime for the process)
import mrt_pkg  # noqa: E402
import mkscene  # noqa: E402

mrt = mrt_pkg.load()
kind, xres, yres = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if kind.endswith(".json"):          # any scene file, at the requested resolution
    scene = mrt.LoadedScene(path=kind).copy().with_resolution(xres, yres)
else:
    text = mkscene.teatime(xres, yres) if kind == "teatime" else mkscene.soup(xres, yres)
    scene = mrt.LoadedScene(text=text).copy()
out = torch.empty(xres * yres, dtype=torch.int32, device="cuda:0")
for flags in (int(a, 0) for a in sys.argv[4:]):
    ctx = mrt.Context(device=0, flags=flags)
    ctx.set_scene(scene)
    reps = 4 if xres * yres > (1 << 26) else 40          # the first launches run at an idle clock
    ms = sorted(ctx.render_bands_device(64, 0, 1, out.data_ptr(), out.numel() * 4, timed=True).trace_ms
                for _ in range(reps))
    best, med = ms[0], ms[len(ms) // 2]
    print(f"flags={flags:#x} trace_ms={best:.3f} median={med:.3f} mrays_per_s={xres * yres / best * 1e-3:.1f}", flush=True)
    ctx.close()
