"""Known answers for the oracle's input staging (cv::cvtColor grey, cv::resize with channels, cv::remap).
No OpenCV here: the expected values are derived by hand from the fixed-point formulas."""
import numpy as np

from orb_slam3_amd import synth


def test_grey_coefficients_sum_to_one_and_pick_channel_order(oracle):
    flat = np.full((3, 5, 3), 200, np.uint8)
    assert np.all(oracle.cvt_gray(flat, True) == 200) and np.all(oracle.cvt_gray(flat, False) == 200)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)
    # (255 * c + 16384) >> 15 for c = 9798 (R), 19235 (G), 3735 (B)
    assert oracle.cvt_gray(px, True).tolist() == [[76, 150, 29]]
    assert oracle.cvt_gray(px, False).tolist() == [[29, 150, 76]]
    rgba = np.concatenate([px, np.full((1, 3, 1), 7, np.uint8)], -1)      # alpha ignored
    assert oracle.cvt_gray(rgba, True).tolist() == [[76, 150, 29]]
    # rounding: R = 1 alone gives (9798 + 16384) >> 15 = 0; R = 2: (19596 + 16384) >> 15 = 1
    assert oracle.cvt_gray(np.array([[[1, 0, 0], [2, 0, 0]]], np.uint8), True).tolist() == [[0, 1]]


def test_remap_identity_half_pixel_and_border(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    yy, xx = np.mgrid[0:12, 0:16].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(img, xx, yy), img)
    half = oracle.remap_linear(img, xx + 0.5, yy)
    want = (img[:, :-1].astype(int) + img[:, 1:] + 1) >> 1
    assert np.array_equal(half[:, :-1], want)
    assert np.array_equal(half[:, -1], (img[:, -1].astype(int) + 1) >> 1)       # right neighbour is the 0 border
    assert np.all(oracle.remap_linear(img, xx + 16, yy) == 0)                    # fully outside
    assert np.all(oracle.remap_linear(img, xx, yy - 13) == 0)
    up = oracle.remap_linear(img, xx, yy - 0.25)                                 # row -0.25: 1/4 border, 3/4 row 0
    assert np.array_equal(up[0], (img[0].astype(int) * 24 * 32 + 512) >> 10)
    # coordinates are rounded to 1/32 pixel, half to even: 0.015625 * 32 = 0.5 -> 0, 0.046875 * 32 = 1.5 -> 2
    a = oracle.remap_linear(img, xx + np.float32(0.015625), yy)
    assert np.array_equal(a, img)
    b = oracle.remap_linear(img, xx + np.float32(0.046875), yy)
    want = (img[:, :-1].astype(int) * 30 * 32 + img[:, 1:].astype(int) * 2 * 32 + 512) >> 10
    assert np.array_equal(b[:, :-1], want)
    # channels are interpolated independently
    rgb = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
    out = oracle.remap_linear(rgb, xx + 0.3, yy + 0.7)
    for c in range(3):
        assert np.array_equal(out[..., c], oracle.remap_linear(np.ascontiguousarray(rgb[..., c]), xx + 0.3, yy + 0.7))


def test_resize_with_channels(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (30, 44, 3), dtype=np.uint8)
    out = oracle.resize_linear_cn(img, 33, 21)
    for c in range(3):
        assert np.array_equal(out[..., c], oracle.resize_linear(np.ascontiguousarray(img[..., c]), 33, 21))
    # exact 2 x 2 decimation takes cv::resize's area path: rounded mean of the four pixels
    half = oracle.resize_linear_cn(img, 22, 15)
    want = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, want)
    assert np.array_equal(oracle.resize_linear_cn(img, 44, 30), img)             # same size: copy


def test_stage_composition_order(oracle):
    img = synth.color_scene(64, 48, 3, 3)
    mx, my = synth.rectify_maps(64, 48, 5)
    a = oracle.stage(img, rgb=True, maps=(mx, my))
    b = oracle.cvt_gray(oracle.remap_linear(img, mx, my), True)      # geometry on the colour image, then grey
    assert np.array_equal(a, b)
    c = oracle.remap_linear(oracle.cvt_gray(img, True), mx, my)      # the other order rounds differently
    assert a.shape == c.shape and not np.array_equal(a, c)
    assert np.array_equal(oracle.stage(img[..., 0].copy()), img[..., 0])


def test_undistort_points_known_answers(oracle):
    K = (458.654, 457.296, 367.215, 248.375)
    # no distortion coefficients: normalise and re-project = identity up to one float rounding
    pts = np.array([[0, 0], [367.215, 248.375], [751, 479], [100.5, 300.25]], np.float32)
    out = oracle.undistort_points(pts, K, (0, 0, 0, 0))
    assert np.allclose(out, pts, atol=1e-4)
    # the principal point is a fixed point of any radial-tangential model
    D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
    pp = np.array([[K[2], K[3]]], np.float32)
    assert np.allclose(oracle.undistort_points(pp, K, D), pp, atol=1e-4)
    # forward model applied to the result reproduces the input (5 iterations converge to << 1 px here)
    out = oracle.undistort_points(pts, K, D).astype(np.float64)
    x, y = (out[:, 0] - K[2]) / K[0], (out[:, 1] - K[3]) / K[1]
    r2 = x * x + y * y
    kr = 1 + D[0] * r2 + D[1] * r2 * r2
    xd = x * kr + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * kr + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    back = np.stack([xd * K[0] + K[2], yd * K[1] + K[3]], 1)
    assert np.abs(back - pts).max() < 0.05
    # barrel distortion (k1 < 0): undistorted corners lie outside the image
    b = oracle.image_bounds(752, 480, K, D)
    assert b[0] < 0 and b[1] > 752 and b[2] < 0 and b[3] > 480
    assert oracle.image_bounds(752, 480, K, (0, 0, 0, 0)).tolist() == [0, 752, 0, 480]
