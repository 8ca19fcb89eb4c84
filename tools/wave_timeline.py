#!/usr/bin/env python3
"""Where one frame's time goes, wave by wave (GPU box only; needs the diagnostic build).

Build the library with -DMRT_TIMELINE (tools/build_timeline_variant.sh), point MRT_HIP_LIB at it and
run this: every wave of trace_kernel records when it began, when its first chunk of records was in
LDS and when it ended, on the 100 MHz clock all XCDs share, and where it ran.  From those records:

  * the frame's span (first begin to last end) against the busy time per SIMD slot,
  * how many waves are resident over time: the ramp at the start, the tail at the end,
  * the gap between a wave's end and the next wave's begin in the same SIMD slot,
  * how evenly the eight XCDs finish.

usage: MRT_HIP_LIB=variants/timeline/libmrt_hip.so wave_timeline.py [teatime4k|teatime8k|teatime1080] [--world W --rank R] [--json PATH]
"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import torch  # noqa: E402
import mrt_pkg  # noqa: E402
import mkscene  # noqa: E402

ENTRY = np.dtype([("t_begin", "<u8"), ("t_staged", "<u8"), ("t_end", "<u8"), ("hw_id", "<u4"), ("xcc_id", "<u4"),
                  ("block_x", "<u4"), ("block_y", "<u4"), ("block_z", "<u4"), ("wave", "<u4"), ("cycles", "<u8"),
                  ("t_setup", "<u8"), ("t_sync", "<u8"), ("t_loaded", "<u8"), ("pad", "<u8")])
MAX_ENTRIES = 1 << 18
TICK_US = 0.01          # the shared clock runs at 100 MHz


def read_timeline(L):
    buf = np.zeros(MAX_ENTRIES, dtype=ENTRY)
    n = C.c_uint32(0)
    rc = L.mrt_debug_timeline_(buf.ctypes.data_as(C.c_void_p), C.c_uint32(MAX_ENTRIES), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"mrt_debug_timeline_ returned {rc}")
    return buf[: n.value].copy()


def analyse(e):
    t0 = int(e["t_begin"].min())
    b = (e["t_begin"].astype(np.int64) - t0) * TICK_US
    s = (e["t_staged"].astype(np.int64) - t0) * TICK_US
    x = (e["t_end"].astype(np.int64) - t0) * TICK_US
    span = float(x.max())
    dur = x - b
    # resident waves over time, 1 us bins
    bins = int(np.ceil(span)) + 1
    delta = np.zeros(bins + 1)
    np.add.at(delta, np.floor(b).astype(int), 1)
    np.add.at(delta, np.minimum(np.floor(x).astype(int) + 1, bins), -1)
    resident = np.cumsum(delta)[:bins]
    peak = resident.max()
    full = np.nonzero(resident >= 0.9 * peak)[0]
    # the slot a wave ran in: XCD, shader engine, CU, SIMD, wave slot
    hw = e["hw_id"]
    wave_slot = hw & 0xF
    simd = (hw >> 4) & 0x3
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    xcc = e["xcc_id"] & 0xF
    simd_key = ((((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    slot_key = simd_key * 16 + wave_slot
    order = np.lexsort((b, slot_key))
    same = slot_key[order][1:] == slot_key[order][:-1]
    gaps = (b[order][1:] - x[order][:-1])[same]
    # busy time of each SIMD: the union of its waves' intervals is what matters for issue, but the
    # simple figure is the sum of wave time over slots against span * slots
    n_simd = len(np.unique(simd_key))
    n_cu = len(np.unique(simd_key // 4))
    per_simd_last = {}
    for k, v in zip(simd_key, x):
        if v > per_simd_last.get(k, 0.0):
            per_simd_last[k] = v
    last = np.array(list(per_simd_last.values()))
    per_xcc_last = [float(x[xcc == i].max()) if np.any(xcc == i) else None for i in range(8)]
    per_xcc_first = [float(b[xcc == i].min()) if np.any(xcc == i) else None for i in range(8)]
    per_xcc_waves = [int(np.sum(xcc == i)) for i in range(8)]
    waves_per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1])
    # issue-side view: wave-microseconds resident per SIMD against the span
    wave_us = float(dur.sum())
    res = {
        "waves": int(len(e)), "span_us": span, "cus_seen": int(n_cu), "simds_seen": int(n_simd),
        "wave_us_median": float(np.median(dur)), "wave_us_p10": float(np.percentile(dur, 10)),
        "wave_us_p90": float(np.percentile(dur, 90)), "wave_us_max": float(dur.max()),
        "stage_us_median": float(np.median(s - b)),
        "setup_us_median": float(np.median((e["t_setup"].astype(np.int64) - t0) * TICK_US - b)),
        "first_barrier_us_median": float(np.median((e["t_sync"].astype(np.int64) - e["t_setup"].astype(np.int64)) * TICK_US)),
        "load_us_median": float(np.median((e["t_loaded"].astype(np.int64) - e["t_sync"].astype(np.int64)) * TICK_US)),
        "second_barrier_us_median": float(np.median((e["t_staged"].astype(np.int64) - e["t_loaded"].astype(np.int64)) * TICK_US)),
        "resident_peak": float(peak), "resident_mean": float(wave_us / span),
        "ramp_us_to_90pct": float(full[0]) if len(full) else None,
        "tail_us_below_90pct": float(span - full[-1]) if len(full) else None,
        "slot_gap_us_median": float(np.median(gaps)) if len(gaps) else None,
        "slot_gap_us_p90": float(np.percentile(gaps, 90)) if len(gaps) else None,
        "slot_gap_us_mean": float(gaps.mean()) if len(gaps) else None,
        "slot_turnovers": int(len(gaps)),
        "simd_last_end_us_min": float(last.min()), "simd_last_end_us_median": float(np.median(last)),
        "simd_last_end_us_max": float(last.max()),
        "waves_per_simd_min": int(waves_per_simd.min()), "waves_per_simd_max": int(waves_per_simd.max()),
        "xcc_first_begin_us": per_xcc_first, "xcc_last_end_us": per_xcc_last, "xcc_waves": per_xcc_waves,
        "cycles_per_us_median": float(np.median(e["cycles"] / np.maximum(dur, 1e-9))),
        "resident_every_10pct": [float(resident[min(bins - 1, int(span * f))]) for f in np.linspace(0, 1, 11)],
    }
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="teatime4k")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--band", type=int, default=64)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--dump", default=None, help="write the raw records of the last frame (npy)")
    args = ap.parse_args()
    mrt = mrt_pkg.load()
    from opencl_raytracing_amd import capi
    L = capi.lib()
    if not hasattr(L, "mrt_debug_timeline_"):
        sys.exit("this library has no timeline: build it with -DMRT_TIMELINE and set MRT_HIP_LIB")
    L.mrt_debug_timeline_.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    size = {"teatime4k": (3840, 2160), "teatime8k": (7680, 4320), "teatime1080": (1920, 1080),
            "teatime400": (400, 400)}[args.which]
    scene = mrt.LoadedScene(text=mkscene.teatime(*size)).copy()
    st = scene.struct
    rows = mrt.band_rows(st.yres, args.band, args.rank, args.world)
    buf = torch.empty(rows * st.xres, dtype=torch.int32, device="cuda:0")
    ctx = mrt.Context(device=0, flags=0)
    out = []
    try:
        ctx.set_scene(scene)
        for _ in range(3):
            ctx.render_bands_device(args.band, args.rank, args.world, buf.data_ptr(), buf.numel() * 4)
        torch.cuda.synchronize()
        read_timeline(L)
        for k in range(args.frames):
            s = ctx.render_bands_device(args.band, args.rank, args.world, buf.data_ptr(), buf.numel() * 4, timed=True)
            torch.cuda.synchronize()
            e = read_timeline(L)
            r = analyse(e)
            r["event_trace_ms"] = s.trace_ms
            r["event_shade_ms"] = s.shade_ms
            r["slices"] = int(s.prim_slices)
            out.append(r)
            print(json.dumps(r), flush=True)
        if args.dump:
            np.save(args.dump, e)
    finally:
        ctx.close()
    if args.json:
        Path(args.json).write_text(json.dumps({"frame": args.which, "world": args.world, "rank": args.rank, "frames": out}, indent=1))


if __name__ == "__main__":
    main()
