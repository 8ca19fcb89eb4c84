"""Worker for tests/test_multirank_cpu.py: one rank of the interleaved row-band
pipeline on the CPU (gloo).  Each rank fills its bands with the oracle standing in
for the GPU renderer, the bands are gathered to rank 0 exactly as bench.py does,
and rank 0 puts the image back together with the library's mrt_bands_to_image."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import mrt_pkg  # noqa: E402

mrt = mrt_pkg.load()
import oracle_py  # noqa: E402
import scene_py  # noqa: E402


def main():
    scene_name, xres, yres, band = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    out_path = sys.argv[5]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scene_py.load_scene(ROOT / "scenes" / f"{scene_name}.json", xres=xres, yres=yres)
    rows = [mrt.band_rows(yres, band, r, world) for r in range(world)]
    stride = max(rows) * xres * 4
    local = torch.zeros(stride, dtype=torch.uint8)
    buf = local.numpy()
    lr = 0
    n_bands = (yres + band - 1) // band
    for b in range(rank, n_bands, world):              # my bands, in increasing order
        r0, r1 = b * band, min(yres, (b + 1) * band)
        part = oracle_py.render(sc, rows=(r0, r1), threads=2)["rgba"][r0:r1]
        buf[lr * xres * 4:(lr + r1 - r0) * xres * 4] = part.reshape(-1)
        lr += r1 - r0
    assert lr == rows[rank]
    gathered = [torch.zeros(stride, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(local, gathered, dst=0)
    # max-over-ranks reduction of a timing, as bench.py does
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        g = torch.stack(gathered).numpy()
        img = mrt.bands_to_image(g, stride, xres, yres, band, world)
        np.save(out_path, img)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
