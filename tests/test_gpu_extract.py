"""GPU parity of the HIP extractor against the CPU oracle, through the C ABI.
Bit-exact bar: keypoint coordinates, octaves, sizes, responses, angles, descriptors, output
order and monoIndex must all be identical (angles are compared exactly; north-star tolerance
would be 1e-3 rad)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


@pytest.fixture(autouse=True, params=["patch", "level"])
def blur_mode(request, monkeypatch):
    """Every test of this file runs twice: with the Gaussian blur computed per keypoint patch inside the descriptor
    kernel, and with whole levels blurred by a kernel of their own (the library picks one per handle by keypoint
    density; ORBX_BLUR_MODE, read when a handle lays out its buffers, forces it)."""
    monkeypatch.setenv("ORBX_BLUR_MODE", request.param)
    return request.param


def _same(ka, da, ma, kb, db, mb):
    assert ma == mb
    assert len(ka) == len(kb)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(ka[f], kb[f]), f
    assert np.array_equal(da, db)


@pytest.mark.parametrize("w,h,nf,nl,seed,lap", [
    (752, 480, 1000, 8, 1, (0, 1000)),      # EuRoC mono (Frame.cc:389)
    (752, 480, 1200, 8, 2, (0, 0)),         # EuRoC stereo
    (1241, 376, 2000, 8, 3, (0, 0)),        # KITTI
    (512, 512, 1000, 8, 4, (0, 511)),       # TUM-VI fisheye overlap
    (600, 350, 1000, 8, 5, (0, 1000)),      # the fork's resized EuRoC
    (641, 479, 700, 8, 6, (100, 300)),      # odd sizes, partial lapping
    (200, 160, 300, 4, 7, (50, 120)),       # few levels
    (320, 256, 50, 8, 8, (0, 0)),           # tiny quota: first sweep overshoots N
    (752, 480, 5000, 8, 9, (0, 0)),         # mono initialisation extractor (5 x nFeatures)
])
def test_extract_matches_oracle(oracle, w, h, nf, nl, seed, lap):
    A = _gpu()
    img = synth.scene(w, h, seed)
    o = oracle.Extractor(nf, 1.2, nl, 20, 7)
    g = A.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h)
    assert np.array_equal(g.mvScaleFactor, o.scale_factors)
    assert np.array_equal(g.mvInvScaleFactor, o.inv_scale_factors)
    assert np.array_equal(g.mvLevelSigma2, o.level_sigma2)
    assert np.array_equal(g.mvInvLevelSigma2, o.inv_level_sigma2)
    assert np.array_equal(g.mnFeaturesPerLevel, o.features_per_level)
    ko, do, mo = o(img, lap)
    kg, dg, mg = g(img, lap)
    _same(ko, do, mo, kg, dg, mg)
    for l in range(nl):
        assert np.array_equal(o.pyramid_level(l), g.pyramid_level(l)), f"pyramid level {l}"
        co = o.level_candidates(l)
        cg, _ = g.debug_level_candidates(l)
        assert np.array_equal(co, cg), f"FAST candidates level {l}"
        lo, lg = o.level_keypoints(l), g.debug_level_keypoints(l)
        assert np.array_equal(np.stack([lo["x"], lo["y"], lo["response"]], 1), lg), f"quadtree level {l}"
    g.close()


@pytest.mark.parametrize("ini,mn", [(20, 7), (12, 7), (40, 5), (7, 7), (5, 20)])
def test_thresholds(oracle, ini, mn):
    A = _gpu()
    img = synth.scene(640, 480, 21, n_prims=120, noise_sigma=1.0)
    o = oracle.Extractor(800, 1.2, 8, ini, mn)
    g = A.ORBextractor(800, 1.2, 8, ini, mn, max_width=640, max_height=480)
    _same(*o(img, (0, 0)), *g(img, (0, 0)))


@pytest.mark.parametrize("w,h", [(640, 480), (640, 400)])     # the second size has cells wider than the 48-byte tile pitch
@pytest.mark.parametrize("ini,mn", [(20, 7), (3, 1), (2, 0), (0, 0), (255, 200), (120, 127)])
def test_saturated_regions_and_extreme_thresholds(oracle, ini, mn, w, h):
    # The FAST kernel's first-stage test works on halved pixel values with byte-wise carries kept out by clamps
    # near 0 and 255, passes everything when the threshold is below its resolution and clamps thresholds above
    # it: pixels at both ends of the range, with every kind of threshold, against the oracle.
    A = _gpu()
    rng = np.random.default_rng(77)
    img = synth.scene(w, h, 33, n_prims=150, noise_sigma=1.5).astype(np.int32)
    img[:160] = np.clip(img[:160] + 150, 0, 255)          # bright band, large parts at 255
    img[h - 160:] = np.clip(img[h - 160:] - 150, 0, 255)  # dark band, large parts at 0
    img[200:280, 100:500] = rng.integers(0, 2, (80, 400)) * 255    # full-range salt and pepper
    img[100:140, 300:600] = 255 - rng.integers(0, 3, (40, 300))    # near-saturated noise
    img[h - 80:h - 40, 50:350] = rng.integers(0, 3, (40, 300))
    img = img.astype(np.uint8)
    o = oracle.Extractor(600, 1.2, 8, ini, mn)
    g = A.ORBextractor(600, 1.2, 8, ini, mn, max_width=w, max_height=h)
    _same(*o(img, (0, 0)), *g(img, (0, 0)))
    g.close()


def test_low_texture_and_flat_images(oracle):
    A = _gpu()
    g = A.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480)
    o = oracle.Extractor(500)
    flat = np.full((480, 640), 77, np.uint8)
    k, d, m = g(flat, (0, 0))
    assert len(k) == 0 and m == 0
    ko, do, mo = o(flat, (0, 0))
    assert len(ko) == 0 and mo == 0
    # a handful of corners only: quota never reached, cells fall back to the low threshold
    sparse = flat.copy()
    rng = np.random.default_rng(5)
    for _ in range(12):
        x, y = int(rng.integers(40, 600)), int(rng.integers(40, 440))
        sparse[y:y + 9, x:x + 9] = 77 + int(rng.integers(8, 60))
        sparse[y, x] += 30
    _same(*o(sparse, (0, 0)), *g(sparse, (0, 0)))


def test_scale_factor_and_levels_variants(oracle):
    A = _gpu()
    img = synth.scene(800, 600, 31)
    for sf, nl in ((1.2, 8), (1.5, 4), (1.1, 6), (2.0, 3)):
        o = oracle.Extractor(900, sf, nl, 20, 7)
        g = A.ORBextractor(900, sf, nl, 20, 7, max_width=800, max_height=600)
        _same(*o(img, (0, 0)), *g(img, (0, 0)))


def test_blurred_levels_equal_the_oracle_on_every_pixel(oracle, blur_mode):
    """k_blur_levels against the oracle's GaussianBlur on EVERY pixel of every level (the descriptor tests only see the
    pixels that some keypoint samples).  Widths are chosen for the kernel's edge cases: a level whose number of dword
    columns is 1 more than a multiple of the 62 a tile stores (the last column is then a tile's halo lane), all four
    values of the width modulo 4, levels narrower than one tile, fewer rows than one strip, and batches in frame slots
    other than 0 whose row padding holds other data."""
    if blur_mode != "level":
        pytest.skip("per-patch mode keeps no blurred level")
    A = _gpu()
    import torch
    sizes = [(1241, 376, 1.0 + 1e-6), (249, 130, 1.2), (250, 171, 1.2), (251, 102, 1.1), (997, 333, 1.25), (1553, 420, 1.25),
             (745, 480, 1.2), (497, 131, 1.5), (120, 110, 1.2)]
    for case, (w, h, sf) in enumerate(sizes):
        nl = 1 if sf < 1.001 else 4
        sf = max(sf, 1.1)
        while nl > 1 and min(w, h) / sf ** (nl - 1) < 40:
            nl -= 1
        imgs = np.stack([synth.scene(w, h, 700 + 3 * case + f, n_prims=150, noise_sigma=2.0) for f in range(3)])
        # rows and frames padded with bytes that are not image data
        buf = torch.full((3, h + 2, w + 37), 201, dtype=torch.uint8, device="cuda")
        buf[:, :h, :w] = torch.from_numpy(imgs).cuda()
        g = A.ORBextractor(800, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=3)
        g.extract_batch_device(buf[:, :h, :w], (0, 0))
        torch.cuda.synchronize()
        g.status()
        for f in range(3):
            o = oracle.Extractor(800, sf, nl, 20, 7)
            o(imgs[f], (0, 0))
            for l in range(nl):
                want = oracle.gaussian7(o.pyramid_level(l))
                got = g.debug_blurred_level(l, f)
                if not np.array_equal(want, got):
                    ys, xs = np.nonzero(want != got)
                    raise AssertionError(f"{w}x{h} sf={sf} frame {f} level {l} ({want.shape[1]}x{want.shape[0]}): {len(ys)} pixels "
                                         f"differ, first at x={xs[0]} y={ys[0]}, columns {sorted(set(xs.tolist()))[:8]}")
        g.close()


def test_random_geometries(oracle):
    # Sizes, pyramid shapes, quotas and thresholds drawn at random: the cell grid, the tile pitch, the rows a FAST
    # wave takes and the alignment of every level row all depend on them.
    A = _gpu()
    rng = np.random.default_rng(20240607)
    for case in range(24):
        w = int(rng.integers(120, 1300))
        h = int(rng.integers(100, min(800, w * 6 // 5) + 1))    # (a level much taller than wide has no quadtree root: refused)
        sf = float(rng.choice([1.1, 1.2, 1.25, 1.33, 1.5, 2.0]))
        nl = int(rng.integers(1, 9))
        while nl > 1 and min(w, h) / sf ** (nl - 1) < 12:      # levels without room for a FAST cell are wanted, levels of a few pixels are not
            nl -= 1
        nf = int(rng.integers(60, 3000))
        if nl == 1:
            nf = min(nf, 2500)                                  # one level takes the whole quota: the on-chip quadtree holds 2800 nodes
        ini = int(rng.integers(4, 40))
        mn = int(rng.integers(1, ini + 1))
        lap = (0, 0) if case % 3 else (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w)))
        img = synth.scene(w, h, 100 + case, n_prims=int(rng.integers(20, 200)), noise_sigma=float(rng.uniform(0, 3)))
        o = oracle.Extractor(nf, sf, nl, ini, mn)
        g = A.ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h)
        try:
            ko, do, mo = o(img, lap)
            kg, dg, mg = g(img, lap)
            _same(ko, do, mo, kg, dg, mg)
        except AssertionError as e:
            raise AssertionError(f"case {case}: {w}x{h} sf={sf} nl={nl} nf={nf} th=({ini},{mn}) lap={lap}: {e}")
        finally:
            g.close()


def test_random_batches_with_rects(oracle):
    # The device-resident batch path at random geometries, with erase rectangles and a lapping area: every frame
    # of the batch against the oracle.
    A = _gpu()
    import torch
    rng = np.random.default_rng(77001)
    for case in range(8):
        w = int(rng.integers(200, 1000))
        h = int(rng.integers(150, min(700, w * 6 // 5) + 1))
        sf = float(rng.choice([1.2, 1.25, 1.5]))
        nl = int(rng.integers(2, 9))
        while nl > 1 and min(w, h) / sf ** (nl - 1) < 12:
            nl -= 1
        nf = int(rng.integers(100, 2000))
        nb = int(rng.integers(2, 6))
        lap = (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w))) if case % 2 else (0, 0)
        rects = []
        for _ in range(int(rng.integers(0, 4))):
            rx, ry = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
            rects.append((rx, ry, int(rng.integers(10, w - rx)), int(rng.integers(10, h - ry))))
        imgs = [synth.scene(w, h, 500 + 10 * case + i, n_prims=int(rng.integers(30, 200))) for i in range(nb)]
        o = oracle.Extractor(nf, sf, nl, 20, 7)
        g = A.ORBextractor(nf, sf, nl, 20, 7, max_width=w, max_height=h, max_batch=nb)
        g.mvDynamicArea = rects
        try:
            kps, desc, counts = g.extract_batch_device(torch.from_numpy(np.stack(imgs)).cuda(), lap)
            torch.cuda.synchronize()
            g.status()
            kps, desc, counts = kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
            for f in range(nb):
                ko, do, mo = o(imgs[f], lap, rects)
                n = int(counts[f, 0])
                assert n == len(ko) and int(counts[f, 1]) == mo, f"frame {f}: counts"
                assert np.array_equal(np.frombuffer(kps[f, :n].tobytes(), dtype=A.KP_DTYPE), ko), f"frame {f}: keypoints"
                assert np.array_equal(desc[f, :n], do), f"frame {f}: descriptors"
        except AssertionError as e:
            raise AssertionError(f"case {case}: {nb} x {w}x{h} sf={sf} nl={nl} nf={nf} lap={lap} rects={rects}: {e}")
        finally:
            g.close()


def _pattern_images(w, h):
    """Image kinds that synth.scene does not make: periodic patterns (equal FAST scores everywhere: ties in the
    3x3 suppression, in the quadtree's response sort and in its final choice), dense noise (every cell full) and
    blocky noise (corners on a lattice)."""
    rng = np.random.default_rng(w * 1000 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    out = {}
    for period in (4, 7, 16):
        out[f"checkerboard {period}"] = ((((xx // period) + (yy // period)) & 1) * 200 + 20).astype(np.uint8)
    out["stripes"] = (((xx // 5) & 1) * 180 + ((yy // 9) & 1) * 60).astype(np.uint8)
    out["dots"] = np.where((xx % 11 == 5) & (yy % 13 == 6), 255, 30).astype(np.uint8)
    out["uniform noise"] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    blocks = rng.integers(0, 256, ((h + 5) // 6, (w + 5) // 6), dtype=np.uint8)
    out["blocky noise"] = np.kron(blocks, np.ones((6, 6), np.uint8))[:h, :w].copy()
    out["two values"] = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    ramp = (xx * 255 // (w - 1)).astype(np.uint8)
    out["ramp + sparse dots"] = np.where(rng.random((h, w)) < 0.002, 255 - ramp, ramp).astype(np.uint8)
    return out


@pytest.mark.parametrize("w,h,nf", [(640, 480, 1000), (453, 331, 2500), (1920, 1080, 5000)])
def test_periodic_and_noise_images(oracle, w, h, nf):
    A = _gpu()
    o = oracle.Extractor(nf, 1.2, 8, 20, 7)
    g = A.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    total = 0
    for name, img in _pattern_images(w, h).items():
        ko, do, mo = o(img, (0, 0))
        try:
            kg, dg, mg = g(img, (0, 0))
        except A.OrbxError as e:
            if e.status != -2:
                raise
            kg, dg, mg = g(img, (0, 0))      # the candidate buffers were regrown: the second call fits
        try:
            _same(ko, do, mo, kg, dg, mg)
        except AssertionError as e:
            raise AssertionError(f"{name} {w}x{h}: {e}")
        total += len(ko)
    assert total > 5 * nf
    g.close()


def test_wider_image_than_the_creation_size_holds_more_keypoints(oracle):
    # A tiny quota on a very wide image: every level keeps one keypoint per non-empty quadtree root
    # (ORBextractor.cc:665-700), more than the arrays sized for the creation size (few roots) can hold.
    A = _gpu()
    g = A.ORBextractor(30, 1.2, 8, 20, 7, max_width=1600, max_height=1200, max_batch=2)
    o = oracle.Extractor(30, 1.2, 8, 20, 7)
    cap0 = g.max_keypoints
    wide = synth.scene(1600, 140, 91, n_prims=300)
    ko, do, mo = o(wide, (0, 0))
    assert len(ko) > cap0
    _same(ko, do, mo, *g(wide, (0, 0)))
    assert g.max_keypoints >= len(ko)
    for kps, desc, mono in g.extract_batch([wide, wide[:, ::-1].copy()], (0, 0))[:1]:
        _same(ko, do, mo, kps, desc, mono)
    full = synth.scene(1600, 1200, 92)
    _same(*o(full, (0, 0)), *g(full, (0, 0)))        # back to the creation size
    g.close()


def test_dynamic_rects(oracle):
    A = _gpu()
    img = synth.scene(640, 480, 5)
    rects = [(100, 80, 200, 160), (400, 300, 120, 100)]
    o = oracle.Extractor(1000)
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    g.mvDynamicArea = rects
    ko, do, mo = o(img, (0, 0), rects)
    _same(ko, do, mo, *g(img, (0, 0)))
    g.mvDynamicArea = []
    _same(*o(img, (0, 0)), *g(img, (0, 0)))


def test_strided_input_and_size_change(oracle):
    A = _gpu()
    big = synth.scene(800, 600, 41)
    view = big[20:500, 30:782]            # 752 x 480 window with a 800-byte row stride
    assert view.strides[0] == 800
    o = oracle.Extractor(1000)
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=800, max_height=600)
    _same(*o(np.ascontiguousarray(view), (0, 0)), *g(view, (0, 0)))
    _same(*o(big, (0, 0)), *g(big, (0, 0)))          # same handle, other size
    small = np.ascontiguousarray(big[:300, :400])
    _same(*o(small, (0, 0)), *g(small, (0, 0)))


def test_empty_and_invalid(oracle):
    A = _gpu()
    g = A.ORBextractor(500, 1.2, 8, 20, 7, max_width=640, max_height=480)
    k, d, m = g(np.zeros((0, 0), np.uint8))
    assert m == -1 and len(k) == 0
    with pytest.raises(A.OrbxError):
        g(np.zeros((1000, 1000), np.uint8))       # larger than the handle
    with pytest.raises(A.OrbxError):
        g(np.zeros((60, 60), np.uint8))           # level 0 has no room for one 35-px cell
    with pytest.raises(A.OrbxError):
        g(np.zeros((480, 200), np.uint8))         # more than twice as tall as wide: DistributeOctTree would have no root node
    # upper levels without room for a cell are not an error: they contribute nothing (ORBextractor.cc:914-915)
    small = synth.scene(100, 100, 3)
    _same(*oracle.Extractor(500, 1.2, 8, 20, 7)(small, (0, 0)), *g(small, (0, 0)))
    with pytest.raises(A.OrbxError):
        A.ORBextractor(500, 1.2, 40, 20, 7)


def test_golden_vectors(golden_dir):
    A = _gpu()
    for f in sorted(glob.glob(os.path.join(golden_dir, "extract_*.npz"))):
        gld = np.load(f)
        w, h = int(gld["width"]), int(gld["height"])
        img = synth.scene(w, h, int(gld["seed"]))
        assert synth.checksum(img) == int(gld["image_crc"])
        p = gld["params"]
        g = A.ORBextractor(int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]), max_width=w, max_height=h)
        g.mvDynamicArea = [tuple(int(v) for v in r) for r in gld["rects"]]
        k, d, m = g(img, tuple(int(v) for v in gld["lapping"]))
        assert m == int(gld["mono_index"]), f
        assert np.array_equal(k, gld["kps"]), f
        assert np.array_equal(d, gld["desc"]), f


def test_batch_equals_single_and_is_deterministic(oracle):
    A = _gpu()
    import torch
    imgs = [synth.scene(752, 480, 100 + i) for i in range(6)]
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=6)
    single = [g(im, (0, 1000)) for im in imgs]
    batch = g.extract_batch(imgs, (0, 1000))
    for s, b in zip(single, batch):
        _same(*s, *b)
    o = oracle.Extractor(1000)
    _same(*o(imgs[3], (0, 1000)), *batch[3])
    # device-resident path, twice
    d = torch.from_numpy(np.stack(imgs)).cuda()
    outs = []
    for _ in range(2):
        kps, desc, counts = g.extract_batch_device(d, (0, 1000))
        torch.cuda.synchronize()
        g.status()
        outs.append((kps.cpu().numpy().copy(), desc.cpu().numpy().copy(), counts.cpu().numpy().copy()))
    c = outs[0][2]
    for f in range(6):
        n = c[f, 0]
        assert n == len(single[f][0]) and c[f, 1] == single[f][2]
        k = np.frombuffer(outs[0][0][f, :n].tobytes(), dtype=A.KP_DTYPE)
        assert np.array_equal(k, single[f][0])
        assert np.array_equal(outs[0][1][f, :n], single[f][1])
        assert np.array_equal(outs[0][0][f, :n], outs[1][0][f, :n]) and np.array_equal(outs[0][1][f, :n], outs[1][1][f, :n])


def test_pipelined_host_batch_equals_single_frames(oracle):
    """orbx_extract_batch with three chunks or more runs as a pipeline (staging copies on helper threads, uploads,
    kernels in per-chunk frame slots and downloads on three streams): every frame must come out as from the
    one-image entry point, for a row-strided input, a ragged last chunk and two calls in a row."""
    A = _gpu()
    n = 120                                                    # 752x480: 43 MB -> four chunks of 30 frames
    scenes = [synth.scene(752, 480, 300 + i) for i in range(10)]
    imgs = [scenes[i % 10] if i % 3 else np.ascontiguousarray(scenes[i % 10][:, ::-1]) for i in range(n)]
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=n)
    g1 = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480)
    single = {}
    for i in range(n):
        key = (i % 10, bool(i % 3))
        if key not in single:
            single[key] = g1(imgs[i], (0, 1000))
    for _ in range(2):
        batch = g.extract_batch(imgs, (0, 1000))
        assert len(batch) == n
        for i, b in enumerate(batch):
            _same(*single[(i % 10, bool(i % 3))], *b)
    _same(*oracle.Extractor(1000)(imgs[119], (0, 1000)), *batch[119])
    # a ragged last chunk (115 = 3 x 29 + 28 frames) of images that are views with a row stride
    wide = [np.pad(s, ((0, 0), (0, 16))) for s in scenes]
    many = [wide[i % 10][:, :752] for i in range(115)]
    out = g.extract_batch(many, (0, 0))
    for i in (0, 28, 29, 57, 58, 86, 87, 114):
        _same(*g1(scenes[i % 10], (0, 0)), *out[i])


def test_full_hd_5000_features(oracle):
    # the batched 1920x1080 / 5000-feature frame shape, one frame against the oracle
    A = _gpu()
    img = synth.scene(1920, 1080, 4, n_prims=1500)
    o = oracle.Extractor(5000)
    g = A.ORBextractor(5000, 1.2, 8, 20, 7, max_width=1920, max_height=1080)
    ko, do, mo = o(img, (0, 0))
    kg, dg, mg = g(img, (0, 0))
    assert len(ko) >= 5000
    _same(ko, do, mo, kg, dg, mg)


def test_failed_geometry_leaves_no_stale_state(oracle):
    # an image too small for the FAST grid is refused every time it is offered, and the handle keeps working
    A = _gpu()
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480)
    o = oracle.Extractor(1000)
    img = synth.scene(752, 480, 61)
    ref = o(img, (0, 0))
    _same(*ref, *g(img, (0, 0)))
    small = synth.scene(100, 60, 62)
    for _ in range(2):
        with pytest.raises(A.OrbxError) as e:
            g(small, (0, 0))
        assert e.value.status == -1
    _same(*ref, *g(img, (0, 0)))
    other = synth.scene(640, 400, 63)
    _same(*o(other, (0, 0)), *g(other, (0, 0)))


def test_full_hd_batch_device(oracle):
    # the batched 1920x1080 / 5000-feature configuration on the device-resident path: 8 distinct frames
    # in one launch sequence, two of them checked against the oracle, all of them against the one-frame path
    A = _gpu()
    import torch
    nb = 8
    imgs = [synth.scene(1920, 1080, 400 + i, n_prims=1500) for i in range(nb)]
    g = A.ORBextractor(5000, 1.2, 8, 20, 7, max_width=1920, max_height=1080, max_batch=nb)
    d = torch.from_numpy(np.stack(imgs)).cuda()
    kps, desc, counts = g.extract_batch_device(d, (0, 0))
    torch.cuda.synchronize()
    g.status()
    kps, desc, counts = kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
    o = oracle.Extractor(5000)
    one = A.ORBextractor(5000, 1.2, 8, 20, 7, max_width=1920, max_height=1080)
    for f in range(nb):
        n = int(counts[f, 0])
        k = np.frombuffer(kps[f, :n].tobytes(), dtype=A.KP_DTYPE)
        ref = o(imgs[f], (0, 0)) if f in (1, 6) else one(imgs[f], (0, 0))
        assert n >= 5000
        _same(ref[0], ref[1], ref[2], k, desc[f, :n], int(counts[f, 1]))


def test_candidate_buffer_regrows(oracle):
    # pure noise has far more FAST corners than the default candidate capacity
    A = _gpu()
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    o = oracle.Extractor(1000)
    g = A.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
    _same(*o(img, (0, 0)), *g(img, (0, 0)))


def test_wave_sort_equals_sequential_introsort():
    """The device sort (one workgroup) must return the permutation of the sequential restatement of
    libstdc++'s std::sort (which tests/test_introsort.py pins to std::sort itself), ties included."""
    import ctypes as C
    A = _gpu()
    rng = np.random.default_rng(11)
    arrays = []
    for n in list(range(0, 70)) + [100, 127, 128, 129, 255, 500, 777, 1000, 1500, 2047, 2048]:
        for mode in range(7):
            i = np.arange(n, dtype=np.uint64)
            if mode == 0:
                cnt, ulx = 2 + rng.integers(0, 2, n), 16 * rng.integers(0, 4, n)          # heavy ties
            elif mode == 1:
                cnt, ulx = 2 + rng.integers(0, 6, n), rng.integers(0, 40, n)
            elif mode == 2:
                cnt, ulx = 2 + rng.integers(0, 50, n), rng.integers(0, 700, n)
            elif mode == 3:
                cnt, ulx = i.copy(), np.zeros(n)                                          # sorted
            elif mode == 4:
                cnt, ulx = n - i, np.zeros(n)                                             # reversed
            elif mode == 5:
                cnt, ulx = np.where(i < n // 2, i, n - i), np.full(n, 3)                  # organ pipe
            else:
                cnt, ulx = np.full(n, 2), np.zeros(n)                                     # all equal
            e = (cnt.astype(np.uint64) << np.uint64(40)) | (ulx.astype(np.uint64) << np.uint64(24)) | i
            arrays.append(e.astype(np.uint64))
    off = np.zeros(len(arrays) + 1, np.int32)
    off[1:] = np.cumsum([len(a) for a in arrays])
    data = np.ascontiguousarray(np.concatenate(arrays))
    dev, host = np.zeros_like(data), np.zeros_like(data)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert A.lib().orbx_debug_wave_sort(p(data), p(off), len(arrays), p(dev), p(host)) == 0
    bad = [k for k in range(len(arrays)) if not np.array_equal(dev[off[k]:off[k + 1]], host[off[k]:off[k + 1]])]
    assert not bad, f"{len(bad)} arrays differ, first: n={len(arrays[bad[0]])} index {bad[0]}"
    # and it is a sort of the input
    for k in (5, 200, len(arrays) - 1):
        s = dev[off[k]:off[k + 1]]
        assert np.array_equal(np.sort(s), np.sort(arrays[k]))
        assert np.all(np.diff((s >> np.uint64(24)).astype(np.int64)) >= 0)
