"""bench.py end to end on the GPU: the JSON line the driver parses has the contract's keys, the
roofline fraction is a fraction, and the two-rank code path (every rank on GPU 0, gloo gather
through host memory) runs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, free_port

pytestmark = pytest.mark.gpu


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--no-secondary", "--no-cpu-baseline"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2
    assert d["config"]["workload"] == "teatime_4k" and d["config"]["frames_in_flight"] == 2
    assert d["value"] > 1000 and abs(d["value"] - 8294400 / d["ms_per_step"] * 1e-3) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["kernel"] == "full_trace_kernel" and rf["peak"] == 157.3
    assert 0.3 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # achieved = algorithmic FLOPs per launch / mean launch duration
    assert abs(rf["achieved"] - rf["flops_per_launch"] / (rf["kernel_ms"] * 1e-3) * 1e-12) < 1e-6 * rf["achieved"]
    assert rf["flops_per_launch"] == 8294400 * (4032 * 45 + 2 * 20)
    assert d["one_frame_in_flight_mrays_per_s"] > 1000 and d["cold_start_ms_per_step"] > 0


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, MRT_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(ROOT / "bench.py"), "--gpus", "2",
                        "--steps", "3", "--warmup", "1", "--no-secondary"], capture_output=True, text=True,
                       timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["yres"] == 4320 and d["scaling"] == "weak"
    assert "REHEARSAL" in d["config"]["gather"]
    assert d["value"] > 0 and "cpu_baseline" not in d
