"""Input staging kernel (remap / resize / grey conversion in front of the pyramid) against the oracle.
Bar: every output byte identical; staged extraction identical to extraction of the oracle-staged image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


@pytest.mark.parametrize("cn,rgb", [(1, False), (3, True), (3, False), (4, True), (4, False)])
@pytest.mark.parametrize("mode", ["none", "remap", "remap_smaller", "resize", "resize_half", "resize_up"])
def test_stage_equals_oracle(oracle, cn, rgb, mode):
    A = _gpu()
    w, h = 203, 131                      # odd sizes: ragged last dword of every row
    img = synth.scene(w, h, 5, n_prims=60) if cn == 1 else synth.color_scene(w, h, 5, cn)
    kw = {}
    if mode == "remap":
        kw["maps"] = synth.rectify_maps(w, h, 7)
    elif mode == "remap_smaller":
        kw["maps"] = synth.rectify_maps(w, h, 8, (160, 97))
    elif mode == "resize":
        kw["size"] = (150, 101)
    elif mode == "resize_half":
        img = np.ascontiguousarray(img[:130, :202])
        w, h = 202, 130
        kw["size"] = (101, 65)
    elif mode == "resize_up":
        kw["size"] = (260, 150)
    st = A.InputStager(w, h, cn, rgb, **kw)
    got = st(img)
    want = oracle.stage(img, rgb=rgb, **kw)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_stage_random_geometries(oracle):
    # Source and destination sizes, channel layouts and modes drawn at random (the scale factors of the resize and
    # the ragged ends of source and destination rows all depend on them).
    A = _gpu()
    rng = np.random.default_rng(515)
    for case in range(24):
        w, h = int(rng.integers(17, 700)), int(rng.integers(9, 500))
        cn = int(rng.choice([1, 3, 4]))
        rgb = bool(rng.integers(0, 2)) if cn > 1 else False
        mode = ["none", "remap", "remap_other_size", "resize"][int(rng.integers(0, 4))]
        img = synth.scene(w, h, 600 + case, n_prims=40) if cn == 1 else synth.color_scene(w, h, 600 + case, cn)
        kw = {}
        if mode == "remap":
            kw["maps"] = synth.rectify_maps(w, h, 700 + case, strength=float(rng.uniform(0.05, 0.6)))
        elif mode == "remap_other_size":
            kw["maps"] = synth.rectify_maps(w, h, 700 + case, (int(rng.integers(8, 800)), int(rng.integers(8, 600))))
        elif mode == "resize":
            kw["size"] = (int(rng.integers(8, 900)), int(rng.integers(8, 700)))
        st = A.InputStager(w, h, cn, rgb, **kw)
        got = st(img)
        want = oracle.stage(img, rgb=rgb, **kw)
        assert got.shape == want.shape and np.array_equal(got, want), f"case {case}: {w}x{h}x{cn} rgb={rgb} {mode} {kw.get('size')}"


def test_stage_maps_far_outside_and_fraction_rounding(oracle):
    A = _gpu()
    rng = np.random.default_rng(3)
    w, h = 64, 48
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mx = rng.uniform(-40, w + 40, (h, w)).astype(np.float32)
    my = rng.uniform(-40, h + 40, (h, w)).astype(np.float32)
    mx[0, :8] = [-1.0, -0.984375, -0.015625, w - 1, w - 0.99, w, 1e9, -1e9]    # edges, half-way fractions, saturation
    my[0, :8] = [0, 0, 0.046875, 0, 0, 0, 3, 3]
    my[1, :4] = [-1.0, h - 1, h - 0.5, h]
    st = A.InputStager(w, h, 1, False, maps=(mx, my))
    assert np.array_equal(st(img), oracle.remap_linear(img, mx, my))


def test_stage_batch_device_strided(oracle):
    A = _gpu()
    import torch
    w, h, n = 200, 120, 3
    imgs = np.stack([synth.color_scene(w, h, 20 + i, 3) for i in range(n)])
    maps = synth.rectify_maps(w, h, 4)
    st = A.InputStager(w, h, 3, True, maps=maps)
    # source rows padded to 640 bytes, destination rows padded to 256
    src = torch.zeros((n, h, 640), dtype=torch.uint8, device="cuda")
    src[:, :, :w * 3] = torch.from_numpy(imgs.reshape(n, h, w * 3)).cuda()
    d_src = src[:, :, :w * 3].view(n, h, w * 3).as_strided((n, h, w, 3), (h * 640, 640, 3, 1))
    dst = torch.zeros((n, h, 256), dtype=torch.uint8, device="cuda")
    out = st.stage_batch_device(d_src, out=dst[:, :, :w])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], oracle.stage(imgs[i], rgb=True, maps=maps))
    assert int(dst[:, :, w:].sum()) == 0          # padding untouched


@pytest.mark.parametrize("mode", ["remap", "resize", "none"])
def test_staged_extraction_equals_extraction_of_staged_image(oracle, mode):
    A = _gpu()
    import torch
    w, h, n = 752, 480, 2
    imgs = np.stack([synth.color_scene(w, h, 30 + i, 3) for i in range(n)])
    kw = {"maps": synth.rectify_maps(w, h, 9, strength=0.3)} if mode == "remap" else \
         ({"size": (640, 400)} if mode == "resize" else {})
    st = A.InputStager(w, h, 3, False, **kw)
    ex = A.ORBextractor(1000, 1.2, 8, 20, 7, st.width, st.height, max_batch=n)
    d_src = torch.from_numpy(imgs).cuda()
    kps, desc, counts = ex.extract_batch_staged_device(st, d_src, (0, 0))
    torch.cuda.synchronize()
    ex.status()
    counts = counts.cpu().numpy()
    for i in range(n):
        grey = oracle.stage(imgs[i], rgb=False, **kw)
        assert np.array_equal(ex.pyramid_level(0, i), grey)
        k_o, d_o, m_o = oracle.Extractor(1000)(grey, (0, 0))
        k = A.keypoints_from_device(kps, counts, i)
        assert counts[i, 0] == len(k_o) and counts[i, 1] == m_o and len(k_o) > 500
        for f in ("x", "y", "size", "angle", "response", "octave"):
            assert np.array_equal(k[f], k_o[f]), f
        assert np.array_equal(desc[i, :len(k_o)].cpu().numpy(), d_o)


def test_staging_against_committed_golden(golden_dir):
    """The HIP path on the seeded cases of synth.staging_golden_cases() against tests/golden/staging.npz."""
    import os
    A = _gpu()
    g = np.load(os.path.join(golden_dir, "staging.npz"))
    c = synth.staging_golden_cases()
    w, h = c["grey"].shape[1], c["grey"].shape[0]
    assert np.array_equal(A.InputStager(w, h, 3, False, maps=c["maps"])(c["color"]), g["remap_bgr"])
    assert np.array_equal(A.InputStager(w, h, 3, True, maps=c["maps_small"])(c["color"]), g["remap_rgb_small"])
    assert np.array_equal(A.InputStager(w, h, 1, False, maps=c["maps"])(c["grey"]), g["remap_grey"])
    assert np.array_equal(A.InputStager(w, h, 3, False, size=tuple(c["resize"]))(c["color"]), g["resize_bgr"])
    assert np.array_equal(A.InputStager(w, h, 1, False, size=tuple(c["half"]))(c["grey"]), g["half_grey"])
    assert np.array_equal(A.InputStager(w, h, 3, True)(c["color"]), g["grey_rgb"])
    kps = c["kps"]
    cam = A.Camera(*map(float, c["K"]), tuple(map(float, c["D"])))
    f = A.FrameFeatures().assign_distorted(cam, 752, 480, kps, synth.random_descriptors(len(kps), 1))
    un, b = f.keypoints()
    assert np.array_equal(np.stack([un["x"], un["y"]], 1), g["undistorted"]) and np.array_equal(b, g["bounds"])
    start, idx = f.grid()
    assert np.array_equal(start, g["cell_start"]) and np.array_equal(idx, g["cell_idx"])
