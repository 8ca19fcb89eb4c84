"""Float64 numpy model of the render path -- TEST INFRASTRUCTURE ONLY.

Written independently of mrt_oracle.c, from the formulas rather than from the
operation order: Moller-Trumbore barycentrics instead of Cramer's rule with
cofactor expansion, the textbook quadratic for spheres, everything in double.
It reproduces the reference's *behaviour* (kernel.cl:123-445 and the glue of
gfxraytrace.hpp:810-878): the (+1, +0.5) pixel offset, the orthographic origin
without the eye, un-normalised perspective directions, near-root-only spheres,
inclusive triangle edges, strict closest-hit updates in object order, the
Blinn-Phong variant that ignores light intensity, truncating byte conversion.
Two restatements that agree (up to FP32-vs-FP64 rounding at primitive edges)
are what pins the oracle in the absence of reference fixtures.
"""
from __future__ import annotations

import numpy as np


def render(scene, chunk=32, rows=None):
    """rows=(y0, y1): render only image rows y0 .. y1-1 (the arrays returned then have y1 - y0
    rows); the full frame otherwise."""
    st = scene.struct
    w, full_h = st.xres, st.yres
    y0, y1 = (0, full_h) if rows is None else rows
    h = y1 - y0
    f64 = np.float64
    eye = np.array(list(st.eye), f64)
    cu, cv, cw = (np.array(list(v), f64) for v in (st.u, st.v, st.w))
    px = np.arange(w, dtype=f64) + 1.0          # pixel grid, gfxraytrace.hpp:850-852
    py = np.arange(y0, y1, dtype=f64) + 1.0
    ux = f64(st.left) + (f64(st.right) - f64(st.left)) * (px + 0.5) / w      # kernel.cl:126
    uy = f64(st.bottom) + (f64(st.top) - f64(st.bottom)) * (py + 0.5) / full_h    # kernel.cl:127
    UX, UY = np.meshgrid(ux, uy)                # [h, w]
    n = w * h
    UX = UX.reshape(n, 1)
    UY = UY.reshape(n, 1)
    if st.projection == 0:   # kernel.cl:130-138
        O = cu * UX + cv * UY
        D = np.broadcast_to(-cw, (n, 3)).copy()
    else:                    # kernel.cl:140-147
        O = np.broadcast_to(eye, (n, 3)).copy()
        D = -cw * f64(st.focal) + cu * UX + cv * UY

    best_t = np.full(n, np.finfo(np.float32).max, f64)   # gfxraytrace.hpp:813
    best_id = np.full(n, -1, np.int64)
    t_lo = 0.0

    sph = scene.sph.astype(f64)
    for k in range(sph.shape[1]):
        c, r = sph[:3, k], sph[3, k]
        oc = O - c
        a = np.einsum("ij,ij->i", D, D)
        b = np.einsum("ij,ij->i", D, oc)
        cc = np.einsum("ij,ij->i", oc, oc) - r * r
        disc = b * b - a * cc
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (-b - np.sqrt(np.where(disc > 0, disc, 0.0))) / a
        ok = (disc >= 0) & (t > t_lo) & (t < best_t)
        best_t = np.where(ok, t, best_t)
        best_id = np.where(ok, k, best_id)

    tri = scene.tri.astype(f64)
    nt = tri.shape[1]
    ns = sph.shape[1]
    for base in range(0, nt, chunk):
        sl = slice(base, min(nt, base + chunk))
        A = tri[0:3, sl].T            # [m, 3]
        B = tri[3:6, sl].T
        Cc = tri[6:9, sl].T
        # Hit point P = A - beta*(A-B) - gamma*(A-C)  (kernel.cl:208-213), i.e.
        # edges e1 = B - A, e2 = C - A with barycentrics (beta, gamma).
        e1 = B - A
        e2 = Cc - A
        pvec = np.cross(D[:, None, :], e2[None, :, :])            # [n, m, 3]
        det = np.einsum("mj,nmj->nm", e1, pvec)
        tvec = O[:, None, :] - A[None, :, :]
        with np.errstate(invalid="ignore", divide="ignore"):
            inv = 1.0 / det
            beta = np.einsum("nmj,nmj->nm", tvec, pvec) * inv
            qvec = np.cross(tvec, e1[None, :, :])
            gamma = np.einsum("nj,nmj->nm", D, qvec) * inv
            t = np.einsum("mj,nmj->nm", e2, qvec) * inv
        hit = (t >= t_lo) & (gamma >= 0) & (gamma <= 1) & (beta >= 0) & (beta <= 1 - gamma)
        for m in range(A.shape[0]):      # object order matters for ties
            ok = hit[:, m] & (t[:, m] > t_lo) & (t[:, m] < best_t)
            best_t = np.where(ok, t[:, m], best_t)
            best_id = np.where(ok, ns + base + m, best_id)

    # shading
    bg = np.array(list(st.background), f64)
    col = np.broadcast_to(bg, (n, 3)).copy()
    hitm = best_id >= 0
    idx = np.nonzero(hitm)[0]
    if idx.size:
        ids = best_id[idx]
        P = O[idx] + D[idx] * best_t[idx, None]
        N = np.zeros((idx.size, 3))
        matid = np.zeros(idx.size, np.int64)
        is_s = ids < ns
        if is_s.any():
            k = ids[is_s]
            N[is_s] = (P[is_s] - sph[:3, k].T)
            matid[is_s] = scene.sph_mat[k]
        if (~is_s).any():
            k = ids[~is_s] - ns
            a_ = tri[0:3, k].T
            N[~is_s] = np.cross(a_ - tri[3:6, k].T, a_ - tri[6:9, k].T)   # kernel.cl:225-227
            matid[~is_s] = scene.tri_mat[k]
        N /= np.linalg.norm(N, axis=1, keepdims=True)
        mat = scene.mat.astype(f64)
        base_col = mat[:3, matid].T
        shin = mat[3, matid]
        if st.shader == 0:
            c = base_col
        else:
            amb = np.array(list(st.ambient), f64)
            c = np.broadcast_to(f64(st.ka) * amb, (idx.size, 3)).copy()
            V = eye - P
            V /= np.linalg.norm(V, axis=1, keepdims=True)
            light = scene.light.astype(f64)
            for l in range(light.shape[1]):
                Lv = light[:3, l] - P
                Lv /= np.linalg.norm(Lv, axis=1, keepdims=True)
                Hv = V + Lv
                Hv /= np.linalg.norm(Hv, axis=1, keepdims=True)
                ndl = np.maximum(0.0, np.einsum("ij,ij->i", N, Lv))
                ndh = np.maximum(0.0, np.einsum("ij,ij->i", N, Hv))
                c += base_col * f64(st.kd) * ndl[:, None]
                c += f64(st.ks) * (ndh ** shin)[:, None] * light[3:6, l]
            c = np.clip(c, 0.0, 1.0)
        col[idx] = c
    # gfximage.hpp:46-51 applied to the float the reference would hold.
    rgb = (col.astype(np.float32) * np.float32(255.0)).astype(np.uint32).astype(np.uint8)
    rgba = np.concatenate([rgb, np.full((n, 1), 255, np.uint8)], axis=1).reshape(h, w, 4)
    return {"rgba": rgba, "ids": best_id.reshape(h, w).astype(np.int32),
            "t": np.where(hitm, best_t, -1.0).reshape(h, w), "rgbf": col.reshape(h, w, 3)}
