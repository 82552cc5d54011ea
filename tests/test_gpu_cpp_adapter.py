"""The C++ host layer (include/orbx/extractor.hpp, matcher.hpp) over the C ABI, run as a
separate native program and compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "orb_slam3_amd", "lib", "adapter_demo")
MBF, FX = np.float32(47.90639384423901), np.float32(435.2046959714599)


def test_cpp_adapter_matches_oracle(oracle, tmp_path):
    assert os.path.exists(DEMO), "adapter_demo not built (make -C orb_slam3_amd/csrc)"
    from orb_slam3_amd import KP_DTYPE, MATCH2_DTYPE
    L, R = synth.stereo_pair(752, 480, 7)
    (tmp_path / "l.raw").write_bytes(L.tobytes())
    (tmp_path / "r.raw").write_bytes(R.tobytes())
    pre = str(tmp_path / "o")
    out = subprocess.run([DEMO, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "752", "480", pre],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "monoEmpty=-1" in out.stdout and "levels=8" in out.stdout
    eL, eR = oracle.Extractor(1200), oracle.Extractor(1200)
    kL, dL, mL = eL(L, (0, 0))
    kR, dR, mR = eR(R, (0, 0))
    assert f"monoL={mL} monoR={mR}" in out.stdout
    rd = lambda name, dt: np.fromfile(pre + name, dtype=dt)
    assert np.array_equal(rd("_kL.bin", KP_DTYPE), kL) and np.array_equal(rd("_kR.bin", KP_DTYPE), kR)
    assert np.array_equal(rd("_dL.bin", np.uint8).reshape(-1, 32), dL)
    assert np.array_equal(rd("_dR.bin", np.uint8).reshape(-1, 32), dR)
    pl = [eL.pyramid_level(l) for l in range(8)]
    pr = [eR.pyramid_level(l) for l in range(8)]
    uR, depth, _, _ = oracle.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors,
                                          float(MBF), float(MBF / FX))
    assert np.array_equal(rd("_uR.bin", np.float32), uR) and np.array_equal(rd("_depth.bin", np.float32), depth)
    knn = rd("_knn.bin", MATCH2_DTYPE)
    ref = oracle.knn2(dL, dR)
    assert np.array_equal(np.stack([knn["idx0"], knn["dist0"], knn["idx1"], knn["dist1"]], 1), ref)
    assert np.array_equal(rd("_pyr3.bin", np.uint8), pl[3].ravel())
    # orbx::Matcher::SearchForInitialization through the C++ layer
    g2 = oracle.Grid(kR, 0.0, 752.0, 0.0, 480.0)
    prev = np.stack([kL["x"], kL["y"]], 1).astype(np.float32)
    m12, prev2, n = oracle.search_for_initialization(kL, dL, g2, dR, prev, 100, 0.9, True)
    assert f"nInit={n}" in out.stdout and n > 10
    assert np.array_equal(rd("_m12.bin", np.int32), m12)
    assert np.array_equal(rd("_prev.bin", np.float32).reshape(-1, 2), prev2)
    # orbx::FrameFeatures: device-built grid and two successive SearchByProjection passes over it
    gs, gi = g2.csr()
    assert np.array_equal(rd("_gstart.bin", np.int32), gs) and np.array_equal(rd("_gidx.bin", np.int32), gi)
    sf = eL.scale_factors
    lv = kL["octave"]
    taken = np.zeros(len(kR), np.uint8)
    args = (g2, dR, None, kL["x"], kL["y"], np.zeros(len(kL), np.float32), lv, (np.float32(15.0) * sf[lv]).astype(np.float32), dL)
    o1, taken, n1 = oracle.search_by_projection(*args, taken, 0.9)
    o2, taken, n2 = oracle.search_by_projection(*args, taken, 0.9)
    assert f"nSbp1={n1} nSbp2={n2}" in out.stdout and n1 > 10
    assert np.array_equal(rd("_sbp1.bin", np.int32), o1) and np.array_equal(rd("_sbp2.bin", np.int32), o2)
    # orbx::InputStager
    yy, xx = np.mgrid[0:480, 0:752].astype(np.float32)
    want = oracle.remap_linear(L, xx + np.float32(2.25), yy - np.float32(1.5))
    assert np.array_equal(rd("_staged.bin", np.uint8).reshape(480, 752), want)
