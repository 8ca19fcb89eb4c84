#!/usr/bin/env python3
"""Frame and per-view build times of the accelerated entry points (GPU box).
usage: time_accel.py [teatime4k|teatime8k|soup16k|teatime400 ...]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for d in ("", "tools"):
    sys.path.insert(0, str(ROOT / d))
import torch  # noqa: E402
import mkscene  # noqa: E402
import mrt_pkg  # noqa: E402

mrt = mrt_pkg.load()
SCENES = {"teatime4k": lambda: mkscene.teatime(3840, 2160), "teatime8k": lambda: mkscene.teatime(7680, 4320),
          "soup16k": lambda: mkscene.soup(16384, 16384), "teatime400": lambda: mkscene.teatime(400, 400),
          "soup4k": lambda: mkscene.soup(4096, 4096)}
for name in (sys.argv[1:] or ["teatime4k", "soup16k"]):
    sc = mrt.LoadedScene(text=SCENES[name]()).copy()
    st = sc.struct
    buf = torch.empty(st.xres * st.yres * 4, dtype=torch.uint8, device="cuda:0")
    for kind in (0, "block", "cull", "grid"):
        r = mrt.renderer(device=0, kind=kind)
        r.set_scene(sc)
        reps = 3 if (kind == 0 and st.xres > 8000) else 20
        for _ in range(2):
            r.render_bands_device(st.yres, 0, 1, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r.render_bands_device(st.yres, 0, 1, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        frame_ms = (time.perf_counter() - t0) / reps * 1e3
        s = r.render_bands_device(st.yres, 0, 1, buf.data_ptr(), buf.numel(), timed=True)
        # a new view per frame: set_camera + frame, back to back
        t0 = time.perf_counter()
        for _ in range(reps):
            r.set_camera(sc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(reps):
            r.set_camera(sc)
            r.render_bands_device(st.yres, 0, 1, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        view_ms = (time.perf_counter() - t2) / reps * 1e3
        extra = f" build {s.build_ms:.3f} ms cells {s.cells} refs {s.refs}" if isinstance(kind, str) and kind == "grid" else ""
        print(f"{name:10s} {str(kind):6s} frame {frame_ms:8.3f} ms (kernels {s.trace_ms:.3f} + {s.shade_ms:.3f})  "
              f"set_camera host {(t1 - t0) / reps * 1e3:.3f} ms, with device {(t2 - t0) / reps * 1e3:.3f} ms; "
              f"set_camera + frame {view_ms:.3f} ms{extra}", flush=True)
        r.close()
