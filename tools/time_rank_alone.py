#!/usr/bin/env python3
"""One rank's share of a strong-scaled frame, alone on the GPU (GPU box only).

The compute side of the multi-GPU scaling curve can be bounded on one GPU: render the interleaved
64-row bands that rank r of `world` owns, with nothing else on the device, and compare with the
whole frame on the same device.  eff = T(whole frame) / (world * T(rank alone)) is the strong-scaling
efficiency N GPUs would reach if the gather cost nothing; 1 / eff - 1 is what cutting the frame into
bands costs by itself (fewer waves per launch, a shading pass of its own, the lane shape chosen on
the rank's rows).

usage: time_rank_alone.py [soup|teatime8k|both] [--json PATH] [--reps N]
"""
import argparse
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402  (one HIP runtime for the process)
import mrt_pkg  # noqa: E402
import mkscene  # noqa: E402



WARM_SECONDS = 0.06


def rank_alone(mrt, scene, world, rank, reps, flags=0, band=None):
    """Median device time (HIP events inside the library: trace + shade), wall time per frame of
    `reps` frames enqueued back to back on one context, and wall time per frame with two frames in
    flight (an mrt_sequence of two slots, what bench.py's jobs use), for rank `rank` of `world`."""
    BAND = band or 64
    st = scene.struct
    rows = mrt.band_rows(st.yres, BAND, rank, world)
    buf = torch.empty(rows * st.xres, dtype=torch.int32, device="cuda:0")
    ctx = mrt.Context(device=0, flags=flags)
    try:
        ctx.set_scene(scene)
        # the clock climbs for the first 25 ms of work or so: a short launch timed cold looks
        # slower than it is, and a rank's share more so than the whole frame
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < WARM_SECONDS:
            for _ in range(4):
                ctx.render_bands_device(BAND, rank, world, buf.data_ptr(), buf.numel() * 4)
            torch.cuda.synchronize()
        ev = []
        for _ in range(reps):
            s = ctx.render_bands_device(BAND, rank, world, buf.data_ptr(), buf.numel() * 4, timed=True)
            ev.append((s.trace_ms + s.shade_ms, s.trace_ms, s.shade_ms, s.kernel, s.prim_slices))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.render_bands_device(BAND, rank, world, buf.data_ptr(), buf.numel() * 4)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
    finally:
        ctx.close()
    seq = mrt.Sequence(device=0, flags=flags, in_flight=2)
    try:
        seq.set_scene(scene)
        bufs = [buf, torch.empty_like(buf)]
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < WARM_SECONDS:
            for k in range(4):
                seq.submit(BAND, rank, world, bufs[k % 2].data_ptr(), buf.numel() * 4)
            seq.wait()
        n = max(reps, 4) * 2
        t0 = time.perf_counter()
        for k in range(n):
            seq.submit(BAND, rank, world, bufs[k % 2].data_ptr(), buf.numel() * 4)
        seq.wait()
        wall2 = (time.perf_counter() - t0) / n * 1e3
    finally:
        seq.close()
    med = statistics.median(e[0] for e in ev)
    return {"rows": rows, "band_rows": BAND, "kernel_ms": med, "trace_ms": statistics.median(e[1] for e in ev),
            "shade_ms": statistics.median(e[2] for e in ev), "back_to_back_ms": wall, "two_in_flight_ms": wall2,
            "kernel": int(ev[0][3]), "slices": int(ev[0][4])}


def strong_scaling_bound(mrt, scene, reps=10, worlds=(2, 4, 8), flags=0, log=None, band=64):
    """The whole frame and, for every world size, ranks 0 and world - 1 alone on the GPU.  Returns
    {"full_frame": {...}, "w<world>": {"rank_alone_ms": slowest of the two ranks (device time),
    "compute_eff": T(whole frame) / (world * that), "ranks": {...}}}.  band: rows per band, or
    "balanced" = mrt_balanced_band_rows for every world size."""
    full = rank_alone(mrt, scene, 1, 0, reps, flags)
    res = {"full_frame": full}
    if log:
        log(f"whole frame: {full}")
    for world in worlds:
        ranks = {}
        rows_per_band = mrt.balanced_band_rows(scene.struct.yres, world) if band == "balanced" else band
        for rank in sorted({0, world - 1}):
            ranks[rank] = rank_alone(mrt, scene, world, rank, reps, flags, band=rows_per_band)
            if log:
                log(f"world {world} rank {rank}: {ranks[rank]}")
        slow = max(r["kernel_ms"] for r in ranks.values())
        slow_b2b = max(r["back_to_back_ms"] for r in ranks.values())
        slow_2 = max(r["two_in_flight_ms"] for r in ranks.values())
        res[f"w{world}"] = {"band_rows": rows_per_band,
                            "rank_alone_ms": slow, "compute_eff": full["kernel_ms"] / (world * slow),
                            "rank_alone_back_to_back_ms": slow_b2b,
                            "compute_eff_back_to_back": full["back_to_back_ms"] / (world * slow_b2b),
                            "rank_alone_two_in_flight_ms": slow_2,
                            "compute_eff_two_in_flight": full["two_in_flight_ms"] / (world * slow_2),
                            "ranks": {str(k): v for k, v in ranks.items()}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="both")
    ap.add_argument("--json", default=None)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--flags", type=lambda s: int(s, 0), default=0)
    ap.add_argument("--band", default="64", help='rows per band, or "balanced"')
    args = ap.parse_args()
    mrt = mrt_pkg.load()
    frames = []
    if args.which in ("soup", "both"):
        frames.append(("soup16k", mkscene.soup(16384, 16384)))
    if args.which in ("teatime8k", "both"):
        frames.append(("teatime8k", mkscene.teatime(7680, 4320)))
    if args.which == "teatime4k":
        frames.append(("teatime4k", mkscene.teatime(3840, 2160)))
    res = {}
    for name, text in frames:
        scene = mrt.LoadedScene(text=text).copy()
        res[name] = strong_scaling_bound(mrt, scene, args.reps, flags=args.flags,
                                         log=lambda m, n=name: print(n, m, flush=True),
                                         band="balanced" if args.band == "balanced" else int(args.band))
        for world in (2, 4, 8):
            e = res[name][f"w{world}"]
            print(f"{name} world {world} band {e['band_rows']}: rank alone {e['rank_alone_ms']:.3f} ms, compute efficiency "
                  f"{e['compute_eff']:.3f} (back to back {e['compute_eff_back_to_back']:.3f}, two frames in flight "
                  f"{e['compute_eff_two_in_flight']:.3f})", flush=True)
    if args.json:
        Path(args.json).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
