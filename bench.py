#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (metric "Mrays/sec + ray-prim tests/sec vs FP32 peak, teatime 4K"):
the teatime mesh (2 spheres + 4032 triangles, 2 lights, perspective,
Blinn-Phong) at 3840x2160, every ray tested against every primitive.  One step
= one frame: ray generation, closest hit, shading, RGBA8 bytes in HBM.  With
N > 1 ranks (one process per GPU, launched by torch.distributed.run) the frame
grows to 3840 x (2160*N) with the viewport's top/bottom scaled by N, so every
rank still renders 2160 rows' worth of interleaved 64-row bands ("weak"
scaling: fixed work per GPU), and every step gathers all bands to rank 0 over
RCCL; the gather of frame k runs on RCCL's stream while frame k+1 renders (two
buffer sets per frame in flight).  The headline keeps FRAMES_IN_FLIGHT = 2
frames enqueued side by side, each on an mrt_ctx and a stream of its own, in
rotation -- the library's mrt_sequence_* entry points (a 4K frame alone leaves
SIMDs idle while its last waves drain and the next frame's first ones start); the same K steps with one frame at a time are
the one_frame_in_flight_* keys.

Rank 0 prints ONE JSON line.  `value` is the throughput of the library's default path
(trace_kernel: every ray x every primitive, most pairs settled by a conservative per-pixel
predicate, the reference arithmetic for the survivors).  `roofline` is NOT about that kernel --
its executed arithmetic is a small fraction of the priced arithmetic -- but about
full_trace_kernel (MRT_FLAG_FULL_TEST), which evaluates the determinants of the 3x3 solve for
every (ray, triangle) pair of the frame with no rejection in screen space, timed in the same run
on the same frame with HIP events on its launch stream: algorithmic FLOPs per launch = rays *
(4032 * 45 + 2 * 20) (45 per ray-triangle test, 20 per ray-sphere test, no culling credit) over
its mean duration, against the 157.3 TFLOP/s FP32 vector peak.  That kernel is NOT a general-ray
kernel: primary rays share their origin (pinhole) or direction (orthographic), its per-view
records fold the shared half in, and per pair it executes 37 of the 45 priced FLOPs
(`roofline.executed_flops_per_pair`, `roofline.frac_executed`).  The kernel that takes arbitrary
rays -- an origin and a direction per ray, nothing shared -- is the one behind mrt_stage_intersect;
`general_ray` times it on the reference's own workload (teatime's 160 000 primary rays as an
array of rays, 4034 objects) and prices it by the same 45 / 20 FLOPs.  The default path is described beside it by flat keys (default_*), among them the
executed VALU issue rate as a fraction of the nominal 78.6 T lane-ops/s.  `cpu_baseline` times
the CPU oracle (oracle/mrt_oracle.c, a restatement of the reference arithmetic -- the reference
has no CPU path of its own and may not be run) on a bounded row sample of the same frame.
`secondary` carries the other benchmark configurations; with N > 1 the strong-scaling frames
(16384^2 soup, 8K teatime: one fixed frame over the N ranks) are also reported as flat soup16k_* /
teatime8k_* keys, among them *_one_gpu_ms_per_frame (the same frame on rank 0's GPU alone, timed
in the same run) and *_speedup_vs_one_gpu.  With N = 1 the compute side of that scaling curve is
bounded on the one GPU: *_rank_alone_ms_w{2,4,8} is the device time of the slower of ranks 0 and
w - 1 rendering its interleaved bands of the frame alone, *_compute_eff_w{2,4,8} = T(whole frame)
/ (w * that) -- the efficiency w GPUs would reach if the gather cost nothing
(*_compute_eff_two_in_flight_w*: the same with two frames in flight on either side, which is how
these jobs run).
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

FLOPS_PER_TRI_TEST = 45.0
FLOPS_PER_SPH_TEST = 20.0
# what the kernels execute per (ray, triangle) pair, a fused multiply-add counted as two:
# full_trace_kernel: two cross products (6 mul + 6 fma) + three dot products (9 fma) + 1 sub
FULL_TRACE_EXECUTED_FLOPS = 37.0
# general_intersect_kernel on this workload: five dot products (15 fma) + 2 sub -- the sixth, the
# determinant that carries the sign of t, joins a wave's filter only where it pays (not here)
GENERAL_RAY_EXECUTED_FLOPS = 32.0
FP32_PEAK_TFLOPS = 157.3          # 256 CU x 128 lanes x 2 FLOP x 2.4 GHz
VALU_ISSUE_PEAK = 78.6e12         # lane-ops/s: 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz
BAND_ROWS = 64
XRES, YRES = 3840, 2160
FRAMES_IN_FLIGHT = 2              # headline: frames enqueued side by side (a context and a stream each)


def algorithmic_flops(st, rays):
    return rays * (st.n_tris * FLOPS_PER_TRI_TEST + st.n_spheres * FLOPS_PER_SPH_TEST)


def cpu_baseline(sc, seconds_target=12.0):
    """Oracle on the host cores, bounded sample: rows from the middle of the frame."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py
    threads = min(16, os.cpu_count() or 1)
    st = sc.struct
    n_prims = st.n_spheres + st.n_tris
    mid = st.yres // 2
    t0 = time.perf_counter()
    oracle_py.render(sc, rows=(mid, mid + threads), threads=threads)     # calibration
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


def profiled_counters(kernel, name="teatime4k"):
    """Mean hardware counters per launch of one kernel from the newest committed rocprofv3 passes
    of this same command (profiles/rNN/teatime4k_pmc_summary.json, made by tools/profile.sh and
    tools/summarize_profile.py; name = soup16k / teatime8k / stage_intersect: the passes of
    tools/profile_frames.sh and tools/profile_stages.sh), or (None, None).  The summary has an entry per launch shape; the
    one with the largest grid is a whole frame of one context, which is what the kernel times
    quoted beside the counters are measured on."""
    for f in sorted((ROOT / "profiles").glob(f"r*/{name}_pmc_summary.json"), reverse=True):
        try:
            found = [e for name, e in json.loads(f.read_text()).items() if kernel in name]
            if found:
                best = max(found, key=lambda e: int(e["dispatch"]["Grid_Size"]))
                return best["counters_mean_per_launch"], str(f.relative_to(ROOT))
        except (KeyError, ValueError):
            continue
    return None, None


def profiled_clock(section="full"):
    """Mean shader clock (GHz) over the XCDs and the samples of one section of the newest committed
    profiles/rNN/clocks_under_load.txt (amd-smi beside a workload kept running: tools/sample_clocks.sh;
    section "full" = MRT_FLAG_FULL_TEST frames back to back), or (None, None)."""
    import re
    for f in sorted((ROOT / "profiles").glob("r*/clocks_under_load.txt"), reverse=True):
        mhz, inside = [], False
        for line in f.read_text().splitlines():
            if line.startswith("=="):
                inside = line.strip() == f"== {section}"
            elif inside and "sclk" in line:
                mhz += [float(x) for x in re.findall(r"\b(\d{3,4})\b", line.split("sclk", 1)[1])]
        if mhz:
            return sum(mhz) / len(mhz) * 1e-3, str(f.relative_to(ROOT))
    return None, None


def clock_under_load(env, job, seconds=2.0):
    """Socket power and mean shader clock over the XCDs while `job`'s frames run back to back for
    `seconds`: amd-smi in a child process, again and again, beside the frames.  The first sample is
    dropped (taken while the clock settles).  None where amd-smi is missing or says nothing."""
    import re
    import subprocess
    import threading
    import torch
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run(["amd-smi", "metric", "-g", str(env.dev_index), "--clock", "--power"],
                                   capture_output=True, text=True, timeout=10)
            except Exception:                         # noqa: BLE001
                return
            mhz = [float(x) for x in re.findall(r"GFX_\d+:\s*\n\s*CLK:\s*(\d+)\s*MHz", r.stdout)]
            watts = re.search(r"SOCKET_POWER:\s*(\d+)\s*W", r.stdout)
            if mhz:
                samples.append((sum(mhz) / len(mhz) * 1e-3, min(mhz) * 1e-3, max(mhz) * 1e-3,
                                float(watts.group(1)) if watts else None))

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t_end, k = time.perf_counter() + seconds, 0
    while time.perf_counter() < t_end:
        for _ in range(4):
            job.step(k)
            k += 1
        torch.cuda.synchronize(env.dev)
    stop.set()
    th.join(timeout=15)
    use = samples[1:] if len(samples) > 1 else samples
    if not use:
        return None
    watts = [w for _, _, _, w in use if w is not None]
    return {"ghz": sum(g for g, _, _, _ in use) / len(use), "slowest_xcd_ghz": min(lo for _, lo, _, _ in use),
            "fastest_xcd_ghz": max(hi for _, _, hi, _ in use), "socket_watts": sum(watts) / len(watts) if watts else None,
            "samples": len(use), "frames": k}


def profiled_traffic(c):
    """Bytes per launch that crossed the L2 -> fabric boundary (FETCH_SIZE and WRITE_SIZE are
    collected in separate --pmc runs, in KiB; on gfx950 FETCH_SIZE counts 16-byte-per-lane reads
    at half their size, hence the factor 2)."""
    try:
        return c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    except (KeyError, TypeError):
        return None


class Env:
    """Process-wide state: rank, devices, the loaded package."""


class Job:
    """One process-per-GPU render job: this rank's interleaved bands of one scene, gathered to
    rank 0.  Two buffer sets, so the gather of frame k overlaps the render of frame k+1."""

    def __init__(self, env, scene, flags=0, in_flight=1):
        """flags: MRT_FLAG_* of a brute-force context, or the name of an mrt_accel method.
        in_flight > 1: the frames go through an mrt_sequence, which keeps that many enqueued side
        by side, each in a slot with a context and a stream of its own, handed out in rotation: the
        GPU then starts a frame while the last waves of the one before drain."""
        import torch
        self.env, self.scene = env, scene
        st = scene.struct
        self.st = st
        self.in_flight = in_flight
        # one frame at a time (and the kernel timings of every job): a plain context on torch's stream
        self.ctx = env.mrt.renderer(device=env.dev_index, kind=flags)
        self.ctx.set_scene(scene)
        self.seq = None
        if in_flight > 1:
            self.seq = env.mrt.Sequence(device=env.dev_index, flags=int(flags), in_flight=in_flight)
            self.seq.set_scene(scene)
        self.slot_streams = [env.stream] if in_flight == 1 else [None] * in_flight
        mrt, world = env.mrt, env.world
        # 64 rows per band, or 32 when that leaves the busiest rank fewer rows (7680x4320 over 8 ranks)
        self.band = mrt.balanced_band_rows(st.yres, world) if world > 1 else BAND_ROWS
        self.rows = mrt.band_rows(st.yres, self.band, env.rank, world)
        nbytes = max(mrt.band_rows(st.yres, self.band, r, world) for r in range(world)) * st.xres * 4
        # two buffer sets per frame in flight: the gather of a frame overlaps the next frame of its slot
        self.n_sets = 2 * in_flight
        self.local = [torch.empty(nbytes, dtype=torch.uint8, device=env.dev) for _ in range(self.n_sets)]
        self.gathered = [([torch.empty(nbytes, dtype=torch.uint8, device=env.gdev) for _ in range(world)]
                          if (env.use_dist and env.rank == 0) else None) for _ in range(self.n_sets)]
        self.works = [None] * self.n_sets

    def step(self, k):
        import torch
        env, b, f = self.env, k % self.n_sets, k % self.in_flight
        if self.works[b] is not None:
            # stream-level wait on the slot's stream: the last gather out of buffer b is done
            # (a buffer comes round again only after its slot's stream is known)
            with torch.cuda.stream(self.slot_streams[f]):
                self.works[b].wait()
            self.works[b] = None
        if self.seq is None:
            self.ctx.render_bands_device(self.band, env.rank, env.world, self.local[b].data_ptr(),
                                         self.local[b].numel(), stream=env.stream.cuda_stream)
        else:
            ptr = self.seq.submit(self.band, env.rank, env.world, self.local[b].data_ptr(), self.local[b].numel())
            if self.slot_streams[f] is None:
                self.slot_streams[f] = torch.cuda.ExternalStream(ptr, device=env.dev)
        if env.use_dist:
            with torch.cuda.stream(self.slot_streams[f]):
                if env.rehearsal:
                    env.dist.gather(self.local[b].cpu(), self.gathered[b], dst=0)
                else:
                    self.works[b] = env.dist.gather(self.local[b], self.gathered[b], dst=0, async_op=True)

    def fence(self):
        import torch
        env = self.env
        for b in range(self.n_sets):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
        torch.cuda.synchronize(env.dev)
        if env.use_dist:
            env.dist.barrier()
        torch.cuda.synchronize(env.dev)

    def timed(self, steps, warmup):
        """Seconds for `steps` frames between two fences, max over ranks."""
        import torch
        env = self.env
        for k in range(warmup):
            self.step(k)
        self.fence()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(k)
        self.fence()
        elapsed = time.perf_counter() - t0
        if env.use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=env.gdev)
            env.dist.all_reduce(tt, op=env.dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def kernel_ms(self, reps):
        """Mean (trace_ms, shade_ms) of this rank's launches: HIP events on the launch stream.
        Reading the events waits for the stream, so every timed frame is enqueued right behind an
        untimed one: its kernels then start on a busy GPU, as the frames of the timed region do,
        not after an idle gap."""
        tr, sh = [], []
        for _ in range(reps):
            self.ctx.render_bands_device(self.band, self.env.rank, self.env.world,
                                         self.local[1].data_ptr(), self.local[1].numel(),
                                         stream=self.env.stream.cuda_stream)
            s = self.ctx.render_bands_device(self.band, self.env.rank, self.env.world,
                                             self.local[0].data_ptr(), self.local[0].numel(),
                                             stream=self.env.stream.cuda_stream, timed=True)
            tr.append(s.trace_ms)
            sh.append(s.shade_ms)
        return sum(tr) / len(tr), sum(sh) / len(sh)

    def close(self):
        self.ctx.close()
        if self.seq is not None:
            self.seq.close()
        self.local = self.gathered = None


def secondary_workloads(env, mkscene):
    """The other benchmark configurations (not the headline).  Every rank takes part."""
    mrt = env.mrt
    res = {}

    def frame_stats(job, name, steps, warmup, extra=None, kernel_reps=3):
        st = job.st
        elapsed = job.timed(steps, warmup)
        tr, sh = job.kernel_ms(kernel_reps)
        rays = st.xres * st.yres
        n_prims = st.n_spheres + st.n_tris
        e = {"xres": st.xres, "yres": st.yres, "n_gpus": env.world, "primitives": n_prims, "band_rows": job.band,
             "frames_in_flight": job.in_flight,
             "ms_per_frame": elapsed / steps * 1e3, "mrays_per_s": rays * steps / elapsed * 1e-6,
             "prim_tests_per_s": rays * n_prims * steps / elapsed,
             "rank0_trace_kernel_ms": tr, "rank0_shade_kernel_ms": sh,
             "rank0_algorithmic_tflops": algorithmic_flops(st, job.rows * st.xres) / tr * 1e-9}
        if extra:
            e.update(extra)
        res[name] = e

    def accel_extra(job):
        """Grid statistics of an mrt_accel job (build time per view, cells, references)."""
        if not isinstance(job.ctx, mrt.Accel) or job.ctx.method != mrt.ACCEL_GRID:
            return None
        s = job.ctx.render_bands_device(job.band, env.rank, env.world, job.local[0].data_ptr(),
                                        job.local[0].numel(), stream=env.stream.cuda_stream, timed=True)
        return {"grid_build_ms_per_view": s.build_ms, "grid_cells": s.cells, "grid_refs": s.refs}

    def one_gpu_reference(scene, name, steps):
        """With N > 1: the same frame on rank 0's GPU alone, in this same run (no gather; the other
        ranks wait on the host), so that the strong-scaling speedup has its own denominator."""
        if env.world == 1:
            return
        if env.rank == 0:
            solo = Env()
            solo.__dict__.update(env.__dict__)
            solo.world, solo.rank, solo.use_dist = 1, 0, False
            job = Job(solo, scene, in_flight=FRAMES_IN_FLIGHT)
            elapsed = job.timed(steps, 1)
            job.close()
            res[name]["one_gpu_ms_per_frame"] = elapsed / steps * 1e3
            res[name]["speedup_vs_one_gpu"] = res[name]["one_gpu_ms_per_frame"] / res[name]["ms_per_frame"]
        host_side_handshake(env, f"one_gpu_{name}")
        env.dist.barrier()

    def rank_alone(scene, name, reps):
        """One GPU only: the compute side of the strong-scaling curve (tools/time_rank_alone.py)."""
        if env.world != 1:
            return
        import time_rank_alone
        # bands as the multi-GPU paths cut them (mrt_balanced_band_rows: 64 rows, or 32 where that leaves
        # the busiest rank fewer -- 4320 rows over 8 ranks: 544 instead of 576 against a mean of 540)
        b = time_rank_alone.strong_scaling_bound(mrt, scene, reps=reps, band="balanced")
        res[name]["whole_frame_kernel_ms"] = b["full_frame"]["kernel_ms"]
        for w in (2, 4, 8):
            res[name][f"rank_alone_band_rows_w{w}"] = b[f"w{w}"]["band_rows"]
            res[name][f"rank_alone_ms_w{w}"] = b[f"w{w}"]["rank_alone_ms"]
            res[name][f"compute_eff_w{w}"] = b[f"w{w}"]["compute_eff"]
            res[name][f"compute_eff_back_to_back_w{w}"] = b[f"w{w}"]["compute_eff_back_to_back"]
            res[name][f"compute_eff_two_in_flight_w{w}"] = b[f"w{w}"]["compute_eff_two_in_flight"]
        res[name]["rank_alone"] = b

    # The 16384 x 16384 triangle soup, one frame split over all ranks (strong scaling).
    soup16k = mrt.LoadedScene(text=mkscene.soup(16384, 16384)).copy()
    job = Job(env, soup16k, in_flight=FRAMES_IN_FLIGHT)
    # (with N ranks the last frame's gather is not hidden behind a next frame: more steps share it)
    frame_stats(job, "soup16k_16384x16384", steps=3 if env.world == 1 else 10, warmup=1, extra={"scaling": "strong"})
    job.close()
    one_gpu_reference(soup16k, "soup16k_16384x16384", steps=3)
    rank_alone(soup16k, "soup16k_16384x16384", reps=10)
    # teatime at 7680 x 4320, split over all ranks.
    tea8k = mrt.LoadedScene(text=mkscene.teatime(7680, 4320)).copy()
    job = Job(env, tea8k, in_flight=FRAMES_IN_FLIGHT)
    frame_stats(job, "teatime_8k_7680x4320", steps=5 if env.world == 1 else 20, warmup=2, extra={"scaling": "strong"})
    job.close()
    one_gpu_reference(tea8k, "teatime_8k_7680x4320", steps=5)
    rank_alone(tea8k, "teatime_8k_7680x4320", reps=20)
    if env.world == 1:
        # The shipped scenes at their native 400 x 400 (the only size the reference itself can render).
        for name in ("scene_2spheres_persp_phong", "scene_gtri_persp_phong", "teatime"):
            job = Job(env, mrt.LoadedScene(path=ROOT / "scenes" / f"{name}.json").copy())
            frame_stats(job, f"native_400x400_{name}", steps=20, warmup=2)
            job.close()
        # The headline frame with parts of the rejection funnel switched off, and through the
        # accelerated entry points (mrt_accel_*: not brute force, same pixels).
        sc4k = mrt.LoadedScene(text=mkscene.teatime(XRES, YRES)).copy()
        for label, flags, steps in (("teatime_4k_edge_test_only_no_circle", mrt.FLAG_NO_CIRCLE, 5),
                                    ("teatime_4k_exact_test_only_no_prefilter", mrt.FLAG_NO_PREFILTER, 2),
                                    ("teatime_4k_accel_block_reject", "block", 5),
                                    ("teatime_4k_accel_tile_cull", "cull", 5),
                                    ("teatime_4k_accel_grid", "grid", 20)):
            job = Job(env, sc4k, flags=flags)
            frame_stats(job, label, steps=steps, warmup=1, extra=accel_extra(job))
            job.close()
        soup = mrt.LoadedScene(text=mkscene.soup(16384, 16384)).copy()
        for label, method in (("soup16k_accel_tile_cull", "cull"), ("soup16k_accel_grid", "grid")):
            job = Job(env, soup, flags=method)
            frame_stats(job, label, steps=3, warmup=1, extra=accel_extra(job))
            job.close()
        job = Job(env, soup, flags=mrt.FLAG_FULL_TEST)          # 4.7e12 pairs, about 3 s per frame
        frame_stats(job, "soup16k_full_test_every_pair", steps=1, warmup=0, kernel_reps=1)
        job.close()
    return res


def set_camera_cost(env, sc, reps=50):
    """mrt_set_camera (per-view records rebuilt on the GPU, asynchronous): milliseconds the call
    keeps the host, and milliseconds per view including the device work, back to back."""
    import torch
    ctx = env.mrt.Context(device=env.dev_index)
    try:
        ctx.set_scene(sc)
        ctx.set_camera(sc)
        torch.cuda.synchronize(env.dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.set_camera(sc)
        t1 = time.perf_counter()
        torch.cuda.synchronize(env.dev)
        t2 = time.perf_counter()
    finally:
        ctx.close()
    return {"host_call_ms": (t1 - t0) / reps * 1e3, "with_device_work_ms": (t2 - t0) / reps * 1e3,
            "primitives": sc.struct.n_spheres + sc.struct.n_tris}


def native_multi_gpu(env, mkscene, world):
    """The library's own single-process multi-GPU path (mrt_multi_*: one context and stream per
    GPU, bands gathered to GPU 0 with RCCL, band-wise copies to the host image) through the
    mrraytracer binary, as a child process of rank 0 while the other ranks wait on the host: the
    7680x4320 teatime frame and the 16384^2 soup on `world` GPUs.  Frame time includes the copy of
    the image to page-locked host memory (unlike `value`).  A failure is reported, not raised; if
    a run fails or times out with the default transport the next one copies through the host."""
    import subprocess
    cli = ROOT / "opencl-raytracing_amd" / "mrraytracer"
    res = {}
    gather = None                                       # the library's choice: RCCL between distinct GPUs
    def run(scene, png, gather):
        child_env = dict(os.environ, MRT_GPUS=str(world), MRT_STATS="1", MRT_REPEAT="5")
        if gather:
            child_env["MRT_GATHER"] = gather
        if env.rehearsal:
            child_env["MRT_GPU_LIST"] = ",".join(["0"] * world)
        elif world == 1:
            child_env["MRT_GPU_LIST"] = "0"          # one rank through the same multi-GPU code
        try:
            r = subprocess.run([str(cli), str(scene), str(png)], capture_output=True, text=True,
                               env=child_env, timeout=100)
            if r.returncode != 0:
                return {"error": r.stderr.strip()[-300:]}
            st = json.loads(r.stderr.strip().splitlines()[-1])
            return {k: st[k] for k in ("n_gpus", "gather", "note", "render_ms_max", "render_ms_sum",
                                       "best_frame_ms", "mrays_per_s", "png_ms", "total_ms")}
        except Exception as e:            # noqa: BLE001 -- a report, never a crash of the benchmark
            return {"error": repr(e)[:300]}

    with tempfile.TemporaryDirectory() as d:
        for key, text in (("teatime8k", mkscene.teatime(7680, 4320)), ("soup16k", mkscene.soup(16384, 16384))):
            scene, png = Path(d) / f"{key}.json", Path(d) / f"{key}.png"
            scene.write_text(text)
            res[key] = run(scene, png, gather)
            if "error" in res[key]:
                gather = "host"
            # with several GPUs also the transport that gathers nothing: every GPU copies its own
            # bands to the host image over its own PCIe link (the gathered frame leaves GPU 0 over one)
            if world > 1 and res[key].get("gather") != "host":
                res[f"{key}_host_copies"] = run(scene, png, "host")
    return res


def general_ray(env, reps=12):
    """mrt_stage_intersect's kernels on the reference's own workload: teatime's 400 x 400 primary
    rays handed over as an array of arbitrary rays (origin and direction each), 4034 objects in
    the reference's 144-byte records.  Device time of the call's kernels (HIP events, without the
    copies either side), median over `reps` calls after two warm-up calls."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "tests"))
    sys.path.insert(0, str(ROOT / "oracle"))
    import scene_py
    from test_gpu_stages import make_inputs
    mrt = env.mrt
    sc = scene_py.load_scene(ROOT / "scenes" / "teatime.json")
    pos, vp, cam, objs, lights, phong = make_inputs(mrt, sc, 400)
    ctx = mrt.Context(device=env.dev_index)
    try:
        rays = ctx.stage_viewrays(cam, ctx.stage_uv(pos, vp), sc.struct.focal)
        flt_max = float(np.finfo(np.float32).max)
        ms = []
        for k in range(reps + 2):
            ctx.stage_intersect(objs, rays, flt_max, 0.0)
            if k >= 2:
                ms.append(ctx.stage_kernel_ms())
        import mkscene
        import torch
        from test_gpu_stages import frame_rays
        big_scene = mrt.LoadedScene(text=mkscene.teatime(XRES, YRES)).copy()
        # The calls above come one after the other, as a host program makes them: the GPU is idle while
        # the host packs, copies and scatters, and its clock follows its load.  The frame kernels of this
        # bench line are timed on a busy GPU (Job.kernel_ms), so the same call is also timed right behind
        # 25 ms of other work (two full-test frames of another context, waited for: nothing runs beside
        # the call's kernels).
        warm_ms = []
        warm = mrt.Context(device=env.dev_index, flags=mrt.FLAG_FULL_TEST)
        try:
            warm.set_scene(big_scene)
            frame = torch.empty(XRES * YRES * 4, dtype=torch.uint8, device=env.dev)
            for k in range(reps + 2):
                for _ in range(2):
                    warm.render_bands_device(BAND_ROWS, 0, 1, frame.data_ptr(), frame.numel(), stream=env.stream.cuda_stream)
                torch.cuda.synchronize(env.dev)
                ctx.stage_intersect(objs, rays, flt_max, 0.0)
                if k >= 2:
                    warm_ms.append(ctx.stage_kernel_ms())
            del frame
        finally:
            warm.close()
        warm_ms.sort()
        # The whole calls, copies and host work included, as the reference's render() makes them
        # (gfxraytrace.hpp:857-944): into a fresh result array per call, as its helpers return one
        # (CLHelpers.hpp:91-293), and into arrays kept from frame to frame.  Best of five frames.
        import time as _time
        bg = tuple(sc.struct.background) + (1.0,)
        calls = {}
        for label, keep in (("fresh_arrays", False), ("arrays_kept", True)):
            best = None
            r_out = x_out = c_out = None
            for _ in range(6):
                t = [_time.perf_counter()]
                uv = ctx.stage_uv(pos, vp); t.append(_time.perf_counter())
                r_out = ctx.stage_viewrays(cam, uv, sc.struct.focal, out=r_out if keep else None); t.append(_time.perf_counter())
                x_out = ctx.stage_intersect(objs, r_out, flt_max, 0.0, out=x_out if keep else None); t.append(_time.perf_counter())
                c_out = ctx.stage_phong(x_out, phong, bg, cam, lights, out=c_out if keep else None); t.append(_time.perf_counter())
                ms4 = [(b - a) * 1e3 for a, b in zip(t, t[1:])]
                if best is None or sum(ms4) < sum(best):
                    best = ms4
            calls[label] = {"uv_ms": best[0], "rays_ms": best[1], "intersect_ms": best[2], "phong_ms": best[3],
                            "frame_ms": sum(best)}
        # ... and on the benchmark frame's own rays: 3840 x 2160 of them as an array (265 MB in, 663 MB
        # of hit records out: the call takes a second, its kernels are what is timed)
        big_rays, big_objs = frame_rays(mrt, ctx, big_scene)
        big_ms = []
        for k in range(3):
            ctx.stage_intersect(big_objs, big_rays, flt_max, 0.0)
            big_ms.append(ctx.stage_kernel_ms())
        big_ms.sort()
        big = {"rays": len(big_rays), "objects": len(big_objs), "kernel_ms": big_ms[1],
               "mrays_per_s": len(big_rays) / big_ms[1] * 1e-3,
               "prim_tests_per_s": len(big_rays) * len(big_objs) / (big_ms[1] * 1e-3)}
        big["achieved"] = len(big_rays) * (int(big_objs["is_triangle"].sum()) * FLOPS_PER_TRI_TEST +
                                           int((big_objs["is_triangle"] == 0).sum()) * FLOPS_PER_SPH_TEST) / (big_ms[1] * 1e-3) * 1e-12
        big["frac"] = big["achieved"] / FP32_PEAK_TFLOPS
        big["frac_executed"] = big["frac"] * GENERAL_RAY_EXECUTED_FLOPS / FLOPS_PER_TRI_TEST
        del big_rays
    finally:
        ctx.close()
    ms.sort()
    med = ms[len(ms) // 2]
    n_tri = int(objs["is_triangle"].sum())
    n_sph = len(objs) - n_tri
    flops = len(rays) * (n_tri * FLOPS_PER_TRI_TEST + n_sph * FLOPS_PER_SPH_TEST)
    achieved = flops / (med * 1e-3) * 1e-12
    gc, gsrc = profiled_counters("mrt::general_intersect_kernel", "stage_intersect")
    return {"insts_valu_per_pair": (gc["SQ_INSTS_VALU"] * 64.0 / (len(rays) * len(objs))) if gc and "SQ_INSTS_VALU" in gc else None,
            "counters_source": gsrc,
            "kernel": "general_intersect_kernel (+ ray bounds, records, finish: every kernel of one mrt_stage_intersect call)",
            "rays": len(rays), "objects": len(objs), "kernel_ms": med, "best_kernel_ms": ms[0],
            "stage_calls_400x400": calls,
            "kernel_ms_behind_other_work": warm_ms[len(warm_ms) // 2],
            "frac_behind_other_work": flops / (warm_ms[len(warm_ms) // 2] * 1e-3) * 1e-12 / FP32_PEAK_TFLOPS,
            "prim_tests_per_s": len(rays) * len(objs) / (med * 1e-3),
            "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_PEAK_TFLOPS,
            "executed_flops_per_pair": GENERAL_RAY_EXECUTED_FLOPS,
            "frac_executed": achieved / FP32_PEAK_TFLOPS * GENERAL_RAY_EXECUTED_FLOPS / FLOPS_PER_TRI_TEST,
            "teatime_4k_rays": big,
            "note": "kernel_ms / frac: call after call, the GPU idle between the calls' kernels; *_behind_other_work: the "
                    "same call right behind 25 ms of frame kernels, as the frame kernels of this line are timed.  "
                    "Arbitrary rays: origin and direction per ray from memory, caller's t range, objects in any order; "
                    "per pair five dot products of the ray's (direction, moment) with per-triangle vectors (three of the "
                    "four determinants of tri_intersect's Cramer solve; the fourth, which carries the sign of t, joins a "
                    "wave's filter only where lines meet many triangles behind their rays) as a conservative filter, the "
                    "reference's unfused arithmetic for the survivors; bit-identical records to the oracle's per-ray loop"}


def host_side_handshake(env, key):
    """Rank 0 announces `key`, the other ranks wait for it through the process group's key-value
    store, i.e. on the host: their GPUs stay idle meanwhile (a collective barrier would spin on
    them).  Best effort: without the store the caller's collective barrier does the waiting."""
    from datetime import timedelta
    try:
        store = env.dist.distributed_c10d._get_default_store()
        if env.rank == 0:
            store.set(key, "1")
        else:
            store.wait([key], timedelta(seconds=600))
    except Exception:                         # noqa: BLE001
        pass


def host_frame(env, job, total_rays):
    """What the CLI pays on top of the kernels: D2H into page-locked host memory (mrt_host_alloc,
    as the CLI does) and into ordinary pageable memory, then the PNG encode."""
    res = {}
    img = None
    for label, out in (("pinned", env.mrt.host_array((job.st.yres, job.st.xres, 4))), ("pageable", None)):
        host_ms, dl = [], 0.0
        for _ in range(5):
            t1 = time.perf_counter()
            img, hs = job.ctx.render(out=out)
            host_ms.append((time.perf_counter() - t1) * 1e3)
            dl = hs.download_ms
        res[label] = {"ms_per_frame": min(host_ms), "download_ms": dl,
                      "mrays_per_s": total_rays / min(host_ms) * 1e-3}
    # A stream of frames into page-locked host memory through mrt_sequence_submit_host: the copy of
    # frame k runs on the sequence's copy stream while its slots go on rendering.
    seq = env.mrt.Sequence(device=env.dev_index, in_flight=FRAMES_IN_FLIGHT)
    try:
        seq.set_scene(job.scene)
        outs = [env.mrt.host_array((job.st.yres, job.st.xres, 4)) for _ in range(2 * FRAMES_IN_FLIGHT)]
        for k in range(len(outs)):
            seq.submit_host(outs[k])
        seq.wait()
        n = 20
        t1 = time.perf_counter()
        for k in range(n):
            seq.submit_host(outs[k % len(outs)])
        seq.wait()
        dt = (time.perf_counter() - t1) / n * 1e3
        same = all(bool((o == img).all()) for o in outs)
        res["pinned_sequence"] = {"ms_per_frame": dt, "mrays_per_s": total_rays / dt * 1e-3, "frames": n,
                                  "frames_in_flight": FRAMES_IN_FLIGHT, "same_bytes_as_one_frame": same}
    finally:
        seq.close()
    with tempfile.TemporaryDirectory() as d:
        t1 = time.perf_counter()
        env.mrt.write_png_rgb(Path(d) / "frame.png", img)
        png_ms = (time.perf_counter() - t1) * 1e3
    res["png_encode_ms"] = png_ms
    res["note"] = "mrt_render_rgba8 into host memory, then mrt_write_png_rgb; not the headline value"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the soup / 8K / funnel-variant frames reported next to the headline")
    args = ap.parse_args()

    # The contract is ONE line on stdout.  Libraries print there too (RCCL writes a five-line
    # version banner to stdout when a communicator is made), so file descriptor 1 is pointed at
    # stderr for the length of the run and the JSON line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch          # before the HIP library: one HIP runtime per process
    import torch.distributed as dist
    import mrt_pkg
    import mkscene

    env = Env()
    env.mrt = mrt_pkg.load()
    env.dist = dist
    env.world = int(os.environ.get("WORLD_SIZE", "1"))
    env.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env.world != args.gpus:
        if env.world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = env.world
    # MRT_BENCH_REHEARSAL=1: every rank on GPU 0 and the gather over gloo through host
    # memory -- only to exercise the N > 1 code path on a one-GPU box; the JSON says so.
    env.rehearsal = os.environ.get("MRT_BENCH_REHEARSAL") == "1"
    env.dev_index = 0 if env.rehearsal else local_rank
    torch.cuda.set_device(env.dev_index)
    env.dev = torch.device("cuda", env.dev_index)
    env.gdev = torch.device("cpu") if env.rehearsal else env.dev
    env.stream = torch.cuda.current_stream(env.dev)
    # MRT_BENCH_FORCE_DIST=1 runs the process-group code (RCCL gather, barrier, max-reduce) even
    # with a single rank, to exercise it on a one-GPU box.
    env.use_dist = env.world > 1 or os.environ.get("MRT_BENCH_FORCE_DIST") == "1"
    if env.use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if env.rehearsal:
            dist.init_process_group("gloo", rank=env.rank, world_size=env.world)
        else:
            dist.init_process_group("nccl", rank=env.rank, world_size=env.world, device_id=env.dev)
    world, rank, mrt = env.world, env.rank, env.mrt

    # teatime at 3840 x (2160*world), parsed by the product's own loader; world == 1 is the 4K benchmark frame.
    sc = mrt.LoadedScene(text=mkscene.teatime(XRES, YRES)).copy()
    if world > 1:
        sc = sc.with_resolution(XRES, YRES * world, top=float(world), bottom=-float(world))
    st = sc.struct
    # Three timed regions of K steps back to back, each with its own W warmup steps: one frame at
    # a time on a GPU that has been idle (its clock climbs for the first 25 ms or so of work: the
    # cold_start_* keys), the headline with two frames in flight (a context and a stream each, in
    # rotation, so that the head of a frame fills the SIMDs the draining tail of the one before
    # leaves idle), and one frame at a time again, now warm (one_frame_in_flight_*).
    job1 = Job(env, sc)
    job = Job(env, sc, in_flight=FRAMES_IN_FLIGHT)
    elapsed_cold = job1.timed(args.steps, args.warmup)
    elapsed = job.timed(args.steps, args.warmup)
    elapsed_one = job1.timed(args.steps, args.warmup)
    job1.close()
    trace_avg, shade_avg = job.kernel_ms(max(10, min(args.steps, 50)))
    # Everything beyond the contract's keys is reported when it works and named when it does not:
    # a failure in an extra must not cost the run its JSON line.
    extras_failed = {}

    def attempt(name, fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as e:                      # noqa: BLE001
            extras_failed[name] = repr(e)[:300]
            return None

    host = attempt("host_frame", host_frame, env, job, st.xres * st.yres) if (world == 1 and rank == 0) else None
    job.close()

    # The same frame (this rank's rows of it) with MRT_FLAG_FULL_TEST: no rejection in screen space,
    # the determinants of the 3x3 solve for every (ray, triangle) pair (37 of the 45 priced FLOPs: the
    # frame's rays share origin or direction) -- the kernel the roofline is about.
    fjob = Job(env, sc, flags=mrt.FLAG_FULL_TEST)
    full_reps = max(3, min(args.steps, 10))
    full_elapsed = fjob.timed(full_reps, 1)
    full_trace, _ = fjob.kernel_ms(full_reps)
    full_rows = fjob.rows
    # ... and what the chip runs at under that kernel, here and now (one GPU: amd-smi's index is the HIP ordinal)
    live_clock = None
    # (not under a profiler, whose preloaded library every child process would inherit)
    profiled = any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY"))
    if world == 1 and rank == 0 and not profiled:
        try:
            live_clock = clock_under_load(env, fjob)
        except Exception:                             # noqa: BLE001 -- an extra: never the run's JSON line
            live_clock = None
    fjob.close()

    out = None
    if rank == 0:
        total_rays = st.xres * st.yres                      # all ranks together, per step
        n_prims = st.n_spheres + st.n_tris
        value = total_rays * args.steps / elapsed * 1e-6
        flops_launch = algorithmic_flops(st, full_rows * st.xres)
        pairs_launch = full_rows * st.xres * n_prims
        achieved = flops_launch / (full_trace * 1e-3) * 1e-12
        fc, fsrc = profiled_counters("mrt::full_trace_kernel") if world == 1 else (None, None)
        dc, dsrc = profiled_counters("mrt::trace_kernel") if world == 1 else (None, None)
        clock_ghz, clock_src = profiled_clock("full")
        out = {
            "metric": "Mrays/s, teatime 3840x2160 Blinn-Phong, exact closest hit of every ray over all 4034 "
                      "primitives (default path: conservative per-pixel prefilter, reference arithmetic on survivors)",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "teatime_4k" if world == 1 else f"teatime_3840x{st.yres}_bands{job.band}",
                       "xres": st.xres, "yres": st.yres, "rays_per_step": total_rays,
                       "primitives": n_prims, "band_rows": job.band, "frames_in_flight": FRAMES_IN_FLIGHT,
                       "gather": ("none" if not env.use_dist else
                                  "REHEARSAL: all ranks on GPU 0, gloo gather through host memory"
                                  if env.rehearsal else "rccl gather to rank 0, overlapped with the next frame"),
                       "output": "RGBA8 resident in HBM (no D2H in the timed region)"},
            "prim_tests_per_s": value * 1e6 * n_prims,
            # the same K steps with one frame in flight (one context, one stream)
            "one_frame_in_flight_ms_per_step": elapsed_one / args.steps * 1e3,
            "one_frame_in_flight_mrays_per_s": total_rays * args.steps / elapsed_one * 1e-6,
            # ... and on a GPU that was idle before the W warmup steps
            "cold_start_ms_per_step": elapsed_cold / args.steps * 1e3,
            "cold_start_mrays_per_s": total_rays * args.steps / elapsed_cold * 1e-6,
            "roofline": {
                "bound": "valu", "kernel": "full_trace_kernel", "kernel_ms": full_trace,
                "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_PEAK_TFLOPS, "traffic": profiled_traffic(fc),
                "flops_per_launch": flops_launch, "pairs_per_launch": pairs_launch,
                "insts_valu_per_pair": (fc["SQ_INSTS_VALU"] * 64.0 / pairs_launch) if fc and "SQ_INSTS_VALU" in fc else None,
                "exec_issue_frac": (fc["SQ_INSTS_VALU"] * 64.0 / (full_trace * 1e-3) / VALU_ISSUE_PEAK)
                if fc and "SQ_INSTS_VALU" in fc else None,
                "counters_source": fsrc,
                "executed_flops_per_pair": FULL_TRACE_EXECUTED_FLOPS,
                "frac_executed": achieved / FP32_PEAK_TFLOPS * FULL_TRACE_EXECUTED_FLOPS / FLOPS_PER_TRI_TEST,
                # amd-smi beside this kernel's frames back to back, in this run (the socket at its power limit,
                # every XCD below the 2.4 GHz the peak is made of); where amd-smi says nothing, the committed sample
                "clock_under_load": live_clock,
                "clock_under_load_ghz": live_clock["ghz"] if live_clock else clock_ghz,
                "clock_source": "amd-smi in this run" if live_clock else clock_src,
                "frac_of_peak_at_that_clock": (achieved / (FP32_PEAK_TFLOPS * (live_clock["ghz"] if live_clock else clock_ghz) / 2.4))
                if (live_clock or clock_ghz) else None,
                "note": "MRT_FLAG_FULL_TEST kernel on the same frame in the same run: the determinants of "
                        "tri_intersect's 3x3 solve for every (ray, triangle) pair, no screen-space rejection, as a fused "
                        "conservative filter with the reference's unfused arithmetic for the survivors.  NOT a general-ray "
                        "kernel: the frame's rays share their origin (pinhole) or direction (orthographic) and the per-view "
                        "records fold that half in, so per pair it executes 37 of the 45 priced FLOPs (two cross products, "
                        "three dot products, one subtraction): frac is priced FLOPs over time, frac_executed executed "
                        "FLOPs over time.  45 FLOP per ray-triangle test, 20 per ray-sphere test; same pixels as the "
                        "default path.  The arbitrary-ray kernel is `general_ray`",
            },
            "full_test_ms_per_frame": full_elapsed / full_reps * 1e3,
            "full_test_mrays_per_s": total_rays * full_reps / full_elapsed * 1e-6,
            "full_test_prim_tests_per_s": total_rays * n_prims * full_reps / full_elapsed,
            # the default path, whose throughput is `value`
            "default_kernel": "trace_kernel (+ shade_keys_kernel)",
            "default_kernel_ms": trace_avg, "default_shade_kernel_ms": shade_avg,
            "default_insts_valu_per_pair": (dc["SQ_INSTS_VALU"] * 64.0 / pairs_launch) if dc and "SQ_INSTS_VALU" in dc else None,
            "default_exec_issue_frac": (dc["SQ_INSTS_VALU"] * 64.0 / (trace_avg * 1e-3) / VALU_ISSUE_PEAK)
            if dc and "SQ_INSTS_VALU" in dc else None,
            "default_traffic": profiled_traffic(dc), "default_counters_source": dsrc,
            "default_note": "the default kernel settles most pairs with a one-FMA bounding-circle test, so it is "
                            "priced by executed VALU issue rate (SQ_INSTS_VALU x 64 lanes / kernel time / 78.6e12), "
                            "not by algorithmic FLOPs",
        }
        if world == 1:
            g = attempt("general_ray", general_ray, env)
            if g is not None:
                out["general_ray"] = g
        if host is not None:
            out["host_frame"] = host
            cam_t = attempt("set_camera_teatime", set_camera_cost, env, sc)
            cam_s = attempt("set_camera_soup", set_camera_cost, env, mrt.LoadedScene(text=mkscene.soup(1024, 1024)).copy())
            if cam_t is not None and cam_s is not None:
                out["set_camera_ms_teatime"] = cam_t["with_device_work_ms"]
                out["set_camera_ms_soup"] = cam_s["with_device_work_ms"]
                out["set_camera"] = {"teatime_4k": cam_t, "soup_17408_primitives": cam_s}
    if not args.no_secondary:
        # (every rank takes part; a failure that is the same on every rank leaves them in step)
        sec = attempt("secondary", secondary_workloads, env, mkscene)
        if rank == 0 and sec is not None:
            out["secondary"] = sec
            # strong scaling of one fixed frame over the N ranks, as flat keys
            if "soup16k_full_test_every_pair" in sec:
                # the roofline kernel on the other graded frame: 16 384 triangles + 1024 spheres at 16384^2
                e = sec["soup16k_full_test_every_pair"]
                sc_, _ = profiled_counters("mrt::full_trace_kernel", "soup16k")
                pairs = e["xres"] * e["yres"] * e["primitives"]
                out["soup16k_full_test_kernel_ms"] = e["rank0_trace_kernel_ms"]
                out["soup16k_full_test_tflops"] = e["rank0_algorithmic_tflops"]
                out["soup16k_full_test_frac"] = e["rank0_algorithmic_tflops"] / FP32_PEAK_TFLOPS
                out["soup16k_full_test_frac_executed"] = out["soup16k_full_test_frac"] * FULL_TRACE_EXECUTED_FLOPS / FLOPS_PER_TRI_TEST
                if sc_ and "SQ_INSTS_VALU" in sc_:
                    out["soup16k_full_test_insts_valu_per_pair"] = sc_["SQ_INSTS_VALU"] * 64.0 / pairs
            for key, name in (("soup16k", "soup16k_16384x16384"), ("teatime8k", "teatime_8k_7680x4320")):
                if name in sec:
                    out[f"{key}_ms_per_frame"] = sec[name]["ms_per_frame"]
                    out[f"{key}_mrays_per_s"] = sec[name]["mrays_per_s"]
                    out[f"{key}_scaling"] = "strong"
                    if "speedup_vs_one_gpu" in sec[name]:
                        out[f"{key}_one_gpu_ms_per_frame"] = sec[name]["one_gpu_ms_per_frame"]
                        out[f"{key}_speedup_vs_one_gpu"] = sec[name]["speedup_vs_one_gpu"]
                    for k2, v2 in sec[name].items():
                        if k2.startswith(("rank_alone_ms_w", "rank_alone_band_rows_w", "compute_eff_", "whole_frame_kernel_ms")):
                            out[f"{key}_{k2}"] = v2
    if not args.no_secondary:
        if env.use_dist:
            torch.cuda.synchronize(env.dev)
        if rank == 0:
            nat = native_multi_gpu(env, mkscene, world)
            out["native_multi_gpu"] = nat
            for key, e in nat.items():
                if "best_frame_ms" in e:
                    out[f"native_{key}_frame_ms"] = e["best_frame_ms"]
                    out[f"native_{key}_mrays_per_s"] = e["mrays_per_s"]
                    out[f"native_{key}_gather"] = e["gather"]
        if env.use_dist:
            host_side_handshake(env, "native_multi_gpu_done")
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cb = attempt("cpu_baseline", cpu_baseline, sc)
            if cb is not None:
                out["cpu_baseline"] = cb
                out["vs_cpu_baseline"] = value / cb["value"]
                out["full_test_vs_cpu_baseline"] = out["full_test_mrays_per_s"] / cb["value"]
        if extras_failed:
            out["extras_failed"] = extras_failed
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if env.use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
