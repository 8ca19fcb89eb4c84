"""mrt_multi_*: one frame over several ranks from one process, through the C ABI.

On a one-GPU machine the ranks share GPU 0 (an ordinal may repeat in the device list), which
exercises everything but the inter-GPU links: the band partition, one context, stream and buffer
per rank, the PEER and HOST transports, and the band-by-band copies that de-interleave straight
into the caller's image.  RCCL cannot place two ranks on one GPU; its call sequence
(ncclCommInitAll, grouped ncclGather, in place on the root) runs with a single rank.  The bar:
the bytes of the single-context frame, whatever the number of ranks and the transport.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, SCENES, GOLDEN
from parity import check

sys.path.insert(0, str(ROOT / "tools"))
import mkscene  # noqa: E402

pytestmark = pytest.mark.gpu


def _single(mrt, sc, flags=0):
    ctx = mrt.Context(device=0, flags=flags)
    try:
        ctx.set_scene(sc)
        img, _ = ctx.render()
    finally:
        ctx.close()
    return img


@pytest.mark.parametrize("gather", ["peer", "host"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_multi_rank_frame_equals_single_context(mrt, world, gather):
    kind = {"peer": mrt.GATHER_PEER, "host": mrt.GATHER_HOST}[gather]
    sc = mrt.LoadedScene(text=mkscene.teatime(1280, 720)).copy()
    want = _single(mrt, sc)
    m = mrt.MultiContext(devices=[0] * world, gather=kind)
    try:
        assert m.world == world and m.gather == kind
        m.set_scene(sc)
        img, st = m.render()
        assert np.array_equal(img, want)
        assert st.n_gpus == world and st.rays == 1280 * 720 and st.render_ms_max > 0
        # into page-locked memory, a second frame, then another view of another size
        pinned = mrt.host_array((720, 1280, 4))
        img2, _ = m.render(out=pinned)
        assert np.array_equal(img2, want)
        view = sc.with_resolution(999, 333)          # a short last band, rows not a multiple of 64
        m.set_camera(view)
        img3, _ = m.render()
        assert np.array_equal(img3, _single(mrt, view))
    finally:
        m.close()


def test_multi_auto_transport_and_flags(mrt):
    sc = mrt.LoadedScene(text=mkscene.soup(640, 480, 4096, 256, seed=3)).copy()
    want = _single(mrt, sc)
    # AUTO with a repeated GPU cannot be RCCL: it settles for peer copies and says so
    m = mrt.MultiContext(devices=[0, 0], band_rows_=16)
    try:
        assert m.gather == mrt.GATHER_PEER
        m.set_scene(sc)
        assert np.array_equal(m.render()[0], want)
    finally:
        m.close()
    # the other brute-force kernels, rank by rank; the reserved flag bits are refused
    with pytest.raises(mrt.MrtError):
        mrt.MultiContext(devices=[0, 0], flags=mrt.RESERVED_FLAG_BITS)
    for flags in (mrt.FLAG_FULL_TEST, mrt.FLAG_NO_CIRCLE):
        m = mrt.MultiContext(devices=[0, 0, 0], flags=flags, band_rows_=32, gather=mrt.GATHER_HOST)
        try:
            m.set_scene(sc)
            assert np.array_equal(m.render()[0], want), flags
        finally:
            m.close()
    # RCCL asked for explicitly with a repeated GPU: refused with a reason
    with pytest.raises(mrt.MrtError) as e:
        mrt.MultiContext(devices=[0, 0], gather=mrt.GATHER_RCCL)
    assert "RCCL" in e.value.message
    with pytest.raises(mrt.MrtError):
        mrt.MultiContext(devices=[0, 7])


def test_multi_rccl_call_sequence_single_rank(mrt):
    """One rank, transport RCCL: communicator from ncclCommInitAll, grouped in-place ncclGather
    on the rank's stream, band copies behind it."""
    sc = mrt.LoadedScene(text=mkscene.teatime(640, 360)).copy()
    m = mrt.MultiContext(devices=[0], gather=mrt.GATHER_RCCL)
    try:
        assert m.gather == mrt.GATHER_RCCL
        m.set_scene(sc)
        for _ in range(3):
            img, st = m.render()
        assert st.gather == mrt.GATHER_RCCL
        assert np.array_equal(img, _single(mrt, sc))
    finally:
        m.close()


def test_cli_multi_gpu_writes_the_same_png(tmp_path):
    """MRT_GPU_LIST=0,0 mrraytracer teatime.json out.png: the bytes of the single-GPU PNG."""
    cli = ROOT / "opencl-raytracing_amd" / "mrraytracer"
    scene = SCENES / "teatime.json"
    one, two, host = tmp_path / "one.png", tmp_path / "two.png", tmp_path / "host.png"
    r = subprocess.run([str(cli), str(scene), str(one)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, MRT_GPU_LIST="0,0", MRT_STATS="1", MRT_REPEAT="3")
    r = subprocess.run([str(cli), str(scene), str(two)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.endswith(f"raytracing...done\nwriting {two}...done\n\n")
    st = json.loads(r.stderr.strip().splitlines()[-1])
    assert st["n_gpus"] == 2 and st["gather"] == "peer" and st["best_frame_ms"] > 0
    assert two.read_bytes() == one.read_bytes()
    env = dict(os.environ, MRT_GPU_LIST="0,0,0", MRT_GATHER="host")
    r = subprocess.run([str(cli), str(scene), str(host)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert host.read_bytes() == one.read_bytes()
    # one rank with the RCCL transport: the system's librccl, loaded by the binary itself
    env = dict(os.environ, MRT_GPU_LIST="0", MRT_GATHER="rccl", MRT_STATS="1")
    r = subprocess.run([str(cli), str(scene), str(host)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stderr.strip().splitlines()[-1])["gather"] == "rccl"
    assert host.read_bytes() == one.read_bytes()
    # a GPU that does not exist is an error, exit status 1
    env = dict(os.environ, MRT_GPU_LIST="0,9")
    r = subprocess.run([str(cli), str(scene), str(host)], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "render failed" in r.stderr


def test_timed_rank_buffers_equal_the_whole_frames_rows(mrt):
    """What tools/time_rank_alone.py and bench.py time as "one rank alone" -- rank 0 and rank
    world - 1 of 2, 4 and 8 rendering their interleaved 64-row bands with the timed call -- is the
    rank's share of the frame: every buffer equals the rows band b = rank, rank + world, ... of
    the whole frame, on the 8K teatime frame and on the soup (at 4096^2; its 16384^2 rows are held
    to the oracle by test_soup_16k_rows_against_oracle)."""
    import torch
    for text in (mkscene.teatime(7680, 4320), mkscene.soup(4096, 4096)):
        sc = mrt.LoadedScene(text=text).copy()
        st = sc.struct
        whole = _single(mrt, sc)
        ctx = mrt.Context(device=0)
        try:
            ctx.set_scene(sc)
            for world in (2, 4, 8):
                for rank in (0, world - 1):
                    rows = mrt.band_rows(st.yres, 64, rank, world)
                    buf = torch.zeros(rows * st.xres * 4, dtype=torch.uint8, device="cuda:0")
                    s = ctx.render_bands_device(64, rank, world, buf.data_ptr(), buf.numel(), timed=True)
                    assert s.rays == rows * st.xres and s.trace_ms > 0
                    got = buf.cpu().numpy().reshape(rows, st.xres, 4)
                    want = np.concatenate([whole[b * 64:(b + 1) * 64] for b in range(rank, (st.yres + 63) // 64, world)])
                    assert np.array_equal(got, want), (st.xres, world, rank)
        finally:
            ctx.close()
