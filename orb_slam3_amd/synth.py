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


def vocabulary(k: int, depth: int, seed: int, prune: float = 0.0, shuffle_ids: bool = True):
    """A synthetic k-ary vocabulary tree of the given depth in flattened form
    (child_off, child_idx, node_desc, word_id, weight, depth).  Children descriptors are noisy copies of
    their parent (so that descents are decisive high up and ambiguous near the leaves); `prune` is the
    probability that an inner node below the root becomes a leaf early; node ids are shuffled like the
    ids of a vocabulary file need not be consecutive per parent.  Leaf weights are positive idf-like
    values, a few are 0 (stopped words)."""
    rng = np.random.default_rng(seed)
    parents, descs, levels = [-1], [rng.integers(0, 256, 32, dtype=np.uint8)], [0]
    frontier = [0]
    children = {0: []}
    for lvl in range(1, depth + 1):
        nxt = []
        for p in frontier:
            if lvl > 1 and rng.random() < prune:
                continue
            for _ in range(k):
                d = descs[p].copy()
                flips = rng.integers(0, 256, 24 if lvl < depth else 6)
                for b in flips:
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                i = len(parents)
                parents.append(p); descs.append(d); levels.append(lvl); children[i] = []
                children[p].append(i)
                nxt.append(i)
        frontier = nxt
    n = len(parents)
    perm = np.arange(n)
    if shuffle_ids:
        perm[1:] = 1 + rng.permutation(n - 1)          # new id of old node i; the root stays 0
    inv = np.argsort(perm)
    child_off = np.zeros(n + 1, np.int32)
    child_idx = []
    for new in range(n):
        ch = [int(perm[c]) for c in children[int(inv[new])]]
        child_idx.extend(ch)
        child_off[new + 1] = len(child_idx)
    node_desc = np.stack([descs[int(inv[new])] for new in range(n)])
    is_leaf = np.diff(child_off) == 0
    word_id = np.full(n, -1, np.int32)
    word_id[is_leaf] = np.arange(int(is_leaf.sum()), dtype=np.int32)
    weight = np.zeros(n, np.float64)
    weight[is_leaf] = rng.uniform(0.5, 9.0, int(is_leaf.sum()))
    stopped = np.nonzero(is_leaf)[0][::37]
    weight[stopped] = 0.0
    return child_off, np.array(child_idx, np.int32), node_desc, word_id, weight, depth
