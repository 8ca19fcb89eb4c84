#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (metric "Mrays/sec + ray-prim tests/sec vs FP32 peak, teatime 4K"):
the teatime mesh (2 spheres + 4032 triangles, 2 lights,
perspective, Blinn-Phong) at 3840x2160, every ray tested against every
primitive.  One step = one frame: ray generation, closest hit, shading, RGBA8
bytes in HBM.  With N > 1 ranks (one process per GPU, launched by
torch.distributed.run) the frame grows to 3840 x (2160*N) with the viewport's
top/bottom scaled by N, so every rank still renders 2160 rows' worth of
interleaved 64-row bands ("weak" scaling: fixed work per GPU), and every step
gathers all bands to rank 0 over RCCL; the gather of frame k runs on RCCL's
stream while frame k+1 renders (two buffer sets).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel
(trace_kernel): algorithmic FLOPs per launch = rays * (4032 * 45 + 2 * 20)
(45 FLOPs per ray-triangle test, 20 per ray-sphere test, brute force, no
culling credit) over the kernel's mean duration measured with HIP events on
the launch stream, against the 157.3 TFLOP/s FP32 vector peak.  `cpu_baseline`
times the CPU oracle (oracle/mrt_oracle.c, a restatement of the reference
arithmetic -- the reference has no CPU path of its own and may not be run) on
a bounded row sample of the same frame.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

FLOPS_PER_TRI_TEST = 45.0
FLOPS_PER_SPH_TEST = 20.0
FP32_PEAK_TFLOPS = 157.3          # 256 CU x 128 lanes x 2 FLOP x 2.4 GHz
BAND_ROWS = 64
XRES, YRES = 3840, 2160


def build_scene(mrt, mkscene, world):
    """teatime at 3840 x (2160*world); world == 1 is the 4K benchmark frame."""
    text = mkscene.teatime(XRES, YRES)
    ls = mrt.LoadedScene(text=text)          # the product's own loader
    sc = ls.copy()
    if world > 1:
        sc = sc.with_resolution(XRES, YRES * world, top=float(world), bottom=-float(world))
    return sc


def cpu_baseline(sc, seconds_target=12.0):
    """Oracle on the host cores, bounded sample: rows from the middle of the frame."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py
    threads = min(16, os.cpu_count() or 1)
    st = sc.struct
    n_prims = st.n_spheres + st.n_tris
    # calibrate on 4 rows, then size the sample for ~seconds_target
    mid = st.yres // 2
    t0 = time.perf_counter()
    oracle_py.render(sc, rows=(mid, mid + threads), threads=threads)
    dt = time.perf_counter() - t0
    rows = int(max(threads, min(st.yres // 2, threads * seconds_target / max(dt, 1e-3))))
    rows -= rows % threads
    r0 = max(0, mid - rows // 2)
    t0 = time.perf_counter()
    oracle_py.render(sc, rows=(r0, r0 + rows), threads=threads)
    dt = time.perf_counter() - t0
    rays = rows * st.xres
    return {
        "value": rays / dt * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"rows {r0}..{r0 + rows} of the {st.xres}x{st.yres} frame ({rays} rays x {n_prims} "
                  f"primitives, {dt:.2f} s, {threads} threads)",
        "prim_tests_per_s": rays * n_prims / dt,
    }


def profiled_traffic():
    """HBM bytes per trace_kernel launch from the committed rocprofv3 counter passes of this same
    command (FETCH_SIZE and WRITE_SIZE are collected in separate --pmc runs, in KiB; on gfx950
    FETCH_SIZE counts 16-byte-per-lane reads at half their size, hence the factor 2)."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*/teatime4k_pmc_summary.json")):
        best = f
    if best is None:
        return None, None
    try:
        data = json.loads(best.read_text())
        for name, entry in data.items():
            if "trace_kernel" in name:
                c = entry["counters_mean_per_launch"]
                return c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024, str(best.relative_to(ROOT))
    except (KeyError, ValueError):
        pass
    return None, None


def secondary_workloads(mrt, mkscene, ctx):
    """The other benchmark configurations on this one GPU, kernel time from HIP events (not the headline)."""
    import torch
    res = {}
    # The headline frame again with the opt-in 4x4 block reject and with no prefilter at all.
    sc4k = mrt.LoadedScene(text=mkscene.teatime(XRES, YRES)).copy()
    for label, flags, reps in (("teatime_4k_opt_in_block_reject", mrt.FLAG_BLOCK_REJECT, 5),
                               ("teatime_4k_edge_test_only_no_circle", mrt.FLAG_NO_CIRCLE, 5),
                               ("teatime_4k_exact_test_only_no_prefilter", mrt.FLAG_NO_PREFILTER, 2)):
        c2 = mrt.Context(device=torch.cuda.current_device(), flags=flags)
        c2.set_scene(sc4k)
        buf = torch.empty(XRES * YRES * 4, dtype=torch.uint8, device="cuda")
        ms = [c2.render_bands_device(YRES, 0, 1, buf.data_ptr(), buf.numel(), timed=True).kernel_ms
              for _ in range(reps)]
        c2.close()
        res[label] = {"kernel_ms": min(ms), "mrays_per_s": XRES * YRES / min(ms) * 1e-3}
    cases = {
        "teatime_8k_7680x4320": lambda: mrt.LoadedScene(text=mkscene.teatime(7680, 4320)).copy(),
        "soup16k_16384x16384": lambda: mrt.LoadedScene(text=mkscene.soup(16384, 16384)).copy(),
    }
    for name, make in cases.items():
        sc = make()
        st = sc.struct
        ctx.set_scene(sc)
        buf = torch.empty(st.xres * st.yres * 4, dtype=torch.uint8, device="cuda")
        ms = []
        for _ in range(3):
            s = ctx.render_bands_device(st.yres, 0, 1, buf.data_ptr(), buf.numel(), timed=True)
            ms.append(s.kernel_ms)
        best = min(ms)
        rays = st.xres * st.yres
        flops = rays * (st.n_tris * FLOPS_PER_TRI_TEST + st.n_spheres * FLOPS_PER_SPH_TEST)
        res[name] = {"rays": rays, "primitives": st.n_spheres + st.n_tris, "kernel_ms": best,
                     "mrays_per_s": rays / best * 1e-3, "prim_tests_per_s": rays * (st.n_spheres + st.n_tris) / best * 1e3,
                     "algorithmic_tflops": flops / best * 1e-9, "frac_of_fp32_peak": flops / best * 1e-9 / FP32_PEAK_TFLOPS}
        del buf
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the soup / 8K frames reported next to the headline")
    args = ap.parse_args()

    import torch          # before the HIP library: one HIP runtime per process
    import torch.distributed as dist
    import mrt_pkg
    import mkscene
    mrt = mrt_pkg.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # MRT_BENCH_REHEARSAL=1: every rank on GPU 0 and the gather over gloo through host
    # memory -- only to exercise the N > 1 code path on a one-GPU box; the JSON says so.
    rehearsal = os.environ.get("MRT_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    sc = build_scene(mrt, mkscene, world)
    st = sc.struct
    ctx = mrt.Context(device=dev_index)
    ctx.set_scene(sc)
    rows = mrt.band_rows(st.yres, BAND_ROWS, rank, world)
    row_bytes = st.xres * 4
    max_rows = max(mrt.band_rows(st.yres, BAND_ROWS, r, world) for r in range(world))
    # Two sets of buffers: the gather of frame k (RCCL, its own stream) overlaps the
    # render of frame k+1; a buffer is reused only after its gather has completed.
    gdev = torch.device("cpu") if rehearsal else dev
    local = [torch.empty(max_rows * row_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    gathered = [([torch.empty(max_rows * row_bytes, dtype=torch.uint8, device=gdev) for _ in range(world)]
                 if (world > 1 and rank == 0) else None) for _ in range(2)]
    works = [None, None]
    stream = torch.cuda.current_stream(dev)

    def step(k):
        b = k & 1
        if works[b] is not None:
            works[b].wait()          # stream-level wait: frame k-2's gather is done with buffer b
            works[b] = None
        ctx.render_bands_device(BAND_ROWS, rank, world, local[b].data_ptr(), local[b].numel(),
                                stream=stream.cuda_stream)
        if world > 1:
            if rehearsal:
                dist.gather(local[b].cpu(), gathered[b], dst=0)
            else:
                works[b] = dist.gather(local[b], gathered[b], dst=0, async_op=True)

    def fence():
        for b in range(2):
            if works[b] is not None:
                works[b].wait()
                works[b] = None
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Kernel-only duration of the dominant kernel: HIP events on the launch stream.
    trace_ms = []
    shade_ms = []
    for _ in range(max(10, min(args.steps, 50))):
        s = ctx.render_bands_device(BAND_ROWS, rank, world, local[0].data_ptr(), local[0].numel(),
                                    stream=stream.cuda_stream, timed=True)
        trace_ms.append(s.trace_ms)
        shade_ms.append(s.shade_ms)
    trace_avg = sum(trace_ms) / len(trace_ms)
    shade_avg = sum(shade_ms) / len(shade_ms)

    if rank == 0:
        total_rays = st.xres * st.yres                      # all ranks together, per step
        n_prims = st.n_spheres + st.n_tris
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays * args.steps / elapsed * 1e-6
        rays_rank = rows * st.xres
        flops_launch = rays_rank * (st.n_tris * FLOPS_PER_TRI_TEST + st.n_spheres * FLOPS_PER_SPH_TEST)
        achieved = flops_launch / (trace_avg * 1e-3) * 1e-12
        out = {
            "metric": "Mrays/s, teatime 3840x2160 Blinn-Phong, brute-force closest hit over 4034 primitives",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "teatime_4k" if world == 1 else f"teatime_3840x{st.yres}_bands{BAND_ROWS}",
                       "xres": st.xres, "yres": st.yres, "rays_per_step": total_rays,
                       "primitives": n_prims, "band_rows": BAND_ROWS,
                       "gather": ("none" if world == 1 else
                                  "REHEARSAL: all ranks on GPU 0, gloo gather through host memory" if rehearsal
                                  else "rccl gather to rank 0"),
                       "output": "RGBA8 resident in HBM (no D2H in the timed region)"},
            "prim_tests_per_s": value * 1e6 * n_prims,
            "roofline": {
                "bound": "valu", "kernel": "trace_kernel", "achieved": achieved, "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
                "traffic": profiled_traffic()[0] if world == 1 else None,
                "traffic_source": profiled_traffic()[1] if world == 1 else None,
                "kernel_ms": trace_avg, "shade_kernel_ms": shade_avg,
                "flops_per_launch": flops_launch,
                "note": "algorithmic FLOPs: 45 per ray-triangle test, 20 per ray-sphere test, every ray x "
                        "every primitive; the kernel rejects most pairs with a cheaper conservative test, so "
                        "executed FLOPs are lower",
            },
        }
        if world == 1:
            # What the CLI pays on top: the same frame into a pageable host buffer (PCIe D2H).
            host_ms = []
            for _ in range(5):
                t1 = time.perf_counter()
                _, hs = ctx.render()
                host_ms.append((time.perf_counter() - t1) * 1e3)
            out["host_frame"] = {"ms_per_frame": min(host_ms), "download_ms": hs.download_ms,
                                 "mrays_per_s": total_rays / min(host_ms) * 1e-3,
                                 "note": "mrt_render_rgba8 into pageable host memory; not the headline value"}
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_workloads(mrt, mkscene, ctx)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sc)
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
