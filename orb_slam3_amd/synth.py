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


# ---- inputs for the matcher tests (frames of keypoints with descriptors that are noisy copies of a
# ---- small base set, so that distances straddle the thresholds and ties occur) ----------------------
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("xr", "<f4"), ("radius", "<f4"),
                        ("min_level", "<i4"), ("max_level", "<i4")])
SCALE_FACTORS = np.array([1.2 ** i for i in range(8)], np.float32)


def noisy_copy(rng, desc, max_flips):
    out = desc.copy()
    for i in range(len(out)):
        for b in rng.integers(0, 256, int(rng.integers(0, max_flips))):
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def matcher_frame(rng, n, base_desc=None, flips=30):
    """Random keypoints in a 752x480 image with descriptors that are noisy copies of base_desc rows."""
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rng.uniform(5, 747, n).astype(np.float32)
    kps["y"] = rng.uniform(5, 475, n).astype(np.float32)
    kps["octave"] = rng.integers(0, 8, n)
    kps["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kps["size"] = 31
    if base_desc is None:
        desc = random_descriptors(n, int(rng.integers(1 << 30)))
    else:
        desc = noisy_copy(rng, base_desc[rng.integers(0, len(base_desc), n)], flips)
    return kps, desc


def projected_queries(rng, kps, n, th=7.0, jitter=6.0):
    """Queries projected near existing keypoints: (queries, predicted level, source keypoint)."""
    src = rng.integers(0, len(kps), n)
    q = np.zeros(n, QUERY_DTYPE)
    q["x"] = kps["x"][src] + rng.normal(0, jitter, n).astype(np.float32)
    q["y"] = kps["y"][src] + rng.normal(0, jitter, n).astype(np.float32)
    lvl = np.clip(kps["octave"][src] + rng.integers(-1, 2, n), 0, 7).astype(np.int32)
    q["radius"] = (th * SCALE_FACTORS[lvl]).astype(np.float32)
    q["min_level"] = lvl - 1
    q["max_level"] = lvl
    return q, lvl, src


def feature_vector(rng, n, n_nodes):
    """Random assignment of n features to n_nodes vocabulary nodes as CSR (idx, off), features ascending."""
    node = rng.integers(0, n_nodes, n)
    order = np.argsort(node, kind="stable")
    off = np.zeros(n_nodes + 1, np.int32)
    np.add.at(off, node + 1, 1)
    return order.astype(np.int32), np.cumsum(off).astype(np.int32)


def matcher_golden_cases():
    """The seeded inputs of the committed matcher golden vectors (tests/golden/matchers.npz): a dict of
    name -> dict of arrays.  Shared by tools/make_golden.py and the tests that replay them."""
    bounds = (0.0, 752.0, 0.0, 480.0)
    cases = {}
    rng = np.random.default_rng(101)
    base = random_descriptors(300, 102)
    kps, desc = matcher_frame(rng, 1200, base)
    ur = np.where(rng.random(len(kps)) < 0.6, kps["x"] - rng.uniform(1, 40, len(kps)), -1).astype(np.float32)
    q, lvl, src = projected_queries(rng, kps, 700)
    q["xr"] = q["x"] - rng.uniform(1, 40, len(q)).astype(np.float32)
    cases["window"] = dict(kps=kps, desc=desc, ur=ur, q=q, lvl=lvl, qdesc=noisy_copy(rng, desc[src], 60),
                           qangle=((kps["angle"][src] + rng.normal(40, 25, len(q))) % 360).astype(np.float32),
                           taken=(rng.random(len(kps)) < 0.2).astype(np.uint8),
                           marks=(rng.random(len(q)) < 0.8).astype(np.uint8), bounds=np.array(bounds, np.float32))
    rng = np.random.default_rng(103)
    base = random_descriptors(120, 104)
    k1, d1 = matcher_frame(rng, 800, base, 16)
    k2, d2 = matcher_frame(rng, 850, base, 16)
    fv1, fv2 = feature_vector(rng, 800, 30), feature_vector(rng, 850, 30)
    cases["bow"] = dict(d1=d1, a1=k1["angle"], v1=(rng.random(800) < 0.7).astype(np.uint8), i1=fv1[0], o1=fv1[1],
                        d2=d2, a2=k2["angle"], v2=(rng.random(850) < 0.7).astype(np.uint8), i2=fv2[0], o2=fv2[1])
    voc = vocabulary(10, 3, 105)
    rng = np.random.default_rng(106)
    leaves = np.nonzero(voc[3] >= 0)[0]
    feat = noisy_copy(rng, voc[2][rng.choice(leaves, 600)], 40)
    cases["vocabulary"] = dict(child_off=voc[0], child_idx=voc[1], node_desc=voc[2], word_id=voc[3], weight=voc[4],
                               depth=np.int32(voc[5]), features=feat)
    counts = rng.integers(1, 25, 120)
    off = np.zeros(len(counts) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    obs = noisy_copy(rng, np.repeat(random_descriptors(len(counts), 107), counts, axis=0), 30)
    cases["distinctive"] = dict(desc=obs, off=off)
    return cases


def rectify_maps(w: int, h: int, seed: int, out_size=None, strength: float = 1.0):
    """Float32 (map_x, map_y) of the kind cv::initUndistortRectifyMap(K, D, R, P, size, CV_32FC1) produces for a
    radial-tangential camera and a small rectifying rotation: for every output pixel the source position.
    Some positions fall outside the source image (the constant border of cv::remap)."""
    rng = np.random.default_rng(seed)
    ow, oh = out_size if out_size is not None else (w, h)
    fx, fy, cx, cy = 0.61 * w, 0.61 * w, 0.49 * w, 0.52 * h
    nfx, nfy, ncx, ncy = 0.47 * ow, 0.47 * ow, 0.5 * ow, 0.5 * oh   # wider than the source: the rim maps outside
    k1, k2, p1, p2 = (np.array([-0.28, 0.07, 2e-4, -1.5e-4]) * strength + rng.normal(0, 1e-3, 4) * strength)
    ax, ay, az = rng.normal(0, 0.01, 3) * strength
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    iR = np.linalg.inv(Rz @ Ry @ Rx)
    v, u = np.mgrid[0:oh, 0:ow].astype(np.float64)
    X = np.stack([(u - ncx) / nfx, (v - ncy) / nfy, np.ones_like(u)], -1) @ iR.T
    x, y = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
    r2 = x * x + y * y
    kr = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return (fx * xd + cx).astype(np.float32), (fy * yd + cy).astype(np.float32)


def color_scene(w: int, h: int, seed: int, channels: int = 3) -> np.ndarray:
    """An H x W x channels uint8 image whose channels differ (three differently seeded scenes; alpha = 255)."""
    planes = [scene(w, h, seed + 1000 * c, n_prims=150) for c in range(3)]
    base = planes[0].astype(np.int32)
    img = np.stack([np.clip((base + p.astype(np.int32)) // 2 + 10 * c, 0, 255).astype(np.uint8)
                    for c, p in enumerate(planes)], -1)
    if channels == 4:
        img = np.concatenate([img, np.full((h, w, 1), 255, np.uint8)], -1)
    return np.ascontiguousarray(img)


def staging_golden_cases():
    """The seeded inputs of tests/golden/staging.npz (input staging, undistortion, frame grid): name -> arrays.
    Shared by tools/make_golden.py and the tests that replay them."""
    w, h = 160, 120
    c = {}
    c["color"] = color_scene(w, h, 71, 3)
    c["grey"] = scene(w, h, 72, n_prims=60)
    c["maps"] = rectify_maps(w, h, 73)
    c["maps_small"] = rectify_maps(w, h, 74, (128, 90))
    c["resize"] = np.array([133, 101])
    c["half"] = np.array([80, 60])
    rng = np.random.default_rng(75)
    kps, desc = matcher_frame(rng, 1500, None, 30)
    kps["x"][:300] = rng.normal(300, 6, 300).astype(np.float32)
    kps["y"][:300] = rng.normal(200, 6, 300).astype(np.float32)
    kps["x"] = np.clip(kps["x"], 0, 751)
    kps["y"] = np.clip(kps["y"], 0, 479)
    c["kps"] = kps
    c["K"] = np.array([458.654, 457.296, 367.215, 248.375], np.float32)
    c["D"] = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
    return c
