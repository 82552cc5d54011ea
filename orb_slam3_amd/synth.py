"""Seeded synthetic test imagery (numpy only): cluttered grey-level scenes that give
FAST plenty of corners on every pyramid level, mono frames and rectified stereo pairs,
plus random / planted descriptor sets for the brute-force matcher.

The generator is deterministic for a given (seed, size): fixtures store a checksum of
the image they were made from so that generator drift is detected.
"""
from __future__ import annotations

import zlib

import numpy as np


def _value_noise(rng: np.random.Generator, w: int, h: int) -> np.ndarray:
    out = np.zeros((h, w), np.float32)
    for cell, amp in ((64, 40.0), (24, 22.0), (9, 12.0)):
        gw, gh = w // cell + 2, h // cell + 2
        g = rng.random((gh, gw), dtype=np.float32)
        ys = np.arange(h, dtype=np.float32) / cell
        xs = np.arange(w, dtype=np.float32) / cell
        y0 = ys.astype(np.int32)
        x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    return out


def _primitives(rng: np.random.Generator, w: int, h: int, n: int):
    kind = rng.integers(0, 3, n)
    cx = rng.integers(-20, w + 20, n)
    cy = rng.integers(-20, h + 20, n)
    sx = rng.integers(4, 121, n)
    sy = rng.integers(4, 121, n)
    grey = rng.integers(0, 256, n)
    disp = rng.integers(2, 61, n)
    ang = rng.random(n) * np.pi
    return kind, cx, cy, sx, sy, grey, disp, ang


def _draw(img: np.ndarray, prims, shift: int) -> None:
    h, w = img.shape
    kind, cx, cy, sx, sy, grey, disp, ang = prims
    for i in range(len(kind)):
        x = int(cx[i]) - shift * int(disp[i])
        y = int(cy[i])
        a, b = int(sx[i]), int(sy[i])
        g = float(grey[i])
        if kind[i] == 0:  # filled rectangle
            x0, x1 = max(0, x - a // 2), min(w, x + a // 2)
            y0, y1 = max(0, y - b // 2), min(h, y + b // 2)
            if x0 < x1 and y0 < y1:
                img[y0:y1, x0:x1] = g
        elif kind[i] == 1:  # filled disc
            r = max(2, a // 3)
            x0, x1 = max(0, x - r), min(w, x + r + 1)
            y0, y1 = max(0, y - r), min(h, y + r + 1)
            if x0 < x1 and y0 < y1:
                yy, xx = np.mgrid[y0:y1, x0:x1]
                m = (xx - x) ** 2 + (yy - y) ** 2 <= r * r
                img[y0:y1, x0:x1][m] = g
        else:  # thick line segment
            L = a
            dx, dy = np.cos(ang[i]), np.sin(ang[i])
            t = np.arange(-L // 2, L // 2 + 1)
            half = 1 + b // 40
            for o in range(-half, half + 1):
                px = np.round(x + t * dx - o * dy).astype(np.int64)
                py = np.round(y + t * dy + o * dx).astype(np.int64)
                ok = (px >= 0) & (px < w) & (py >= 0) & (py < h)
                img[py[ok], px[ok]] = g


def scene(w: int, h: int, seed: int, n_prims: int = 400, noise_sigma: float = 2.0) -> np.ndarray:
    """One mono frame, uint8 [h, w]."""
    rng = np.random.default_rng(seed)
    img = 60.0 + _value_noise(rng, w, h)
    prims = _primitives(rng, w, h, n_prims)
    _draw(img, prims, 0)
    img += rng.normal(0.0, noise_sigma, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def stereo_pair(w: int, h: int, seed: int, n_prims: int = 400, noise_sigma: float = 2.0):
    """Rectified pair: same primitives, right view shifted left by a per-primitive
    disparity in [2, 60] px, background shifted by 3 px, independent sensor noise."""
    rng = np.random.default_rng(seed)
    bg = 60.0 + _value_noise(rng, w + 8, h)
    prims = _primitives(rng, w, h, n_prims)
    order = np.argsort(prims[6], kind="stable")      # far (small disparity) first: consistent occlusion
    prims = tuple(p[order] for p in prims)
    left = bg[:, 0:w].copy()
    right = bg[:, 3:w + 3].copy()
    _draw(left, prims, 0)
    _draw(right, prims, 1)
    left += rng.normal(0.0, noise_sigma, left.shape).astype(np.float32)
    right += rng.normal(0.0, noise_sigma, right.shape).astype(np.float32)
    to8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)
    return to8(left), to8(right)


def checksum(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def random_descriptors(n: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def planted_map(queries: np.ndarray, n_map: int, seed: int, max_flips: int = 40):
    """Map descriptor set of n_map rows: uniform random bits, with every query planted
    once (0..max_flips random bit flips) at a seeded row, so best / second-best are
    non-trivial and exact ties occur."""
    rng = np.random.default_rng(seed)
    m = rng.integers(0, 256, (n_map, 32), dtype=np.uint8)
    nq = len(queries)
    rows = rng.choice(n_map, size=min(nq, n_map), replace=False)
    for qi, r in enumerate(rows):
        d = queries[qi].copy()
        for bit in rng.integers(0, 256, int(rng.integers(0, max_flips + 1))):
            d[bit >> 3] ^= np.uint8(1 << (bit & 7))
        m[r] = d
    return m, rows
