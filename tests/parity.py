"""Shared parity criterion: <= 1 LSB per 8-bit channel on
>= 99.9 % of pixels; every other pixel must lie on a primitive / silhouette
edge, i.e. have an 8-neighbour whose winning object id differs in the oracle."""
import numpy as np


def edge_mask(ids: np.ndarray) -> np.ndarray:
    """True where a pixel or one of its 8 neighbours has a different winner."""
    h, w = ids.shape
    pad = np.pad(ids, 1, mode="edge")
    m = np.zeros((h, w), bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                m |= pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] != ids
    return m


def check(rgba, ref_rgba, ref_ids, frac=0.999):
    """Returns a dict of counts; raises AssertionError when the criterion fails."""
    d = np.abs(rgba[..., :3].astype(np.int16) - ref_rgba[..., :3].astype(np.int16)).max(axis=-1)
    n = d.size
    within = int((d <= 1).sum())
    bad = d > 1
    off_edge = int((bad & ~edge_mask(ref_ids)).sum())
    stats = {"pixels": n, "exact": int((d == 0).sum()), "within_1lsb": within, "beyond_1lsb": int(bad.sum()),
             "beyond_off_edge": off_edge}
    assert within >= frac * n, stats
    assert off_edge == 0, stats
    assert (rgba[..., 3] == 255).all()
    return stats
