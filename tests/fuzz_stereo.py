"""Random-case hunt for differences in ComputeStereoMatches between the HIP path and the CPU oracle (not collected by
pytest).

    python tests/fuzz_stereo.py [n_cases] [seed]

Draws frame size, pyramid shape, quota and calibration at random, extracts both images on both sides, matches, and
prints every case whose keypoints, uRight or depth differ.  Exit status 1 if any differ."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from orb_slam3_amd import synth          # noqa: E402
import orb_slam3_amd as A                # noqa: E402
from oracle import oracle as O           # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    O.build()
    O.lib()
    rng = np.random.default_rng(seed)
    bad = matched = 0
    t0 = time.time()
    for case in range(n):
        w = int(rng.integers(160, 1600))
        h = int(rng.integers(120, min(1000, w * 6 // 5) + 1))
        sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
        nl = int(rng.integers(1, 9))
        while nl > 1 and min(w, h) / sf ** (nl - 1) < 12:
            nl -= 1
        nf = int(rng.integers(50, 2500))
        ini = int(rng.integers(5, 40))
        mn = int(rng.integers(1, ini + 1))
        mbf, fx = float(rng.uniform(5, 800)), float(rng.uniform(150, 1500))
        p = dict(case=case, w=w, h=h, sf=sf, nl=nl, nf=nf, ini=ini, mn=mn, mbf=mbf, fx=fx)
        L, R = synth.stereo_pair(w, h, int(rng.integers(1 << 20)))
        eL, eR = O.Extractor(nf, sf, nl, ini, mn), O.Extractor(nf, sf, nl, ini, mn)
        kL, dL, _ = eL(L, (0, 0))
        kR, dR, _ = eR(R, (0, 0))
        pl = [eL.pyramid_level(l) for l in range(nl)]
        pr = [eR.pyramid_level(l) for l in range(nl)]
        uR, depth, _, _ = O.stereo_match(kL, dL, kR, dR, pl, pr, eL.scale_factors, eL.inv_scale_factors, mbf, mbf / fx)
        try:
            gl = A.ORBextractor(nf, sf, nl, ini, mn, w, h)
            gr = A.ORBextractor(nf, sf, nl, ini, mn, w, h)
        except A.OrbxError as e:
            print(f"refused {p}: {e}", flush=True)
            continue
        try:
            kgl, dgl, _ = gl(L, (0, 0))
            kgr, dgr, _ = gr(R, (0, 0))
            u, d = A.compute_stereo_matches(gl, gr, kgl, dgl, kgr, dgr, mbf, mbf / fx)
            same = np.array_equal(kgl, kL) and np.array_equal(kgr, kR) and np.array_equal(u, uR) and np.array_equal(d, depth)
        except A.OrbxError as e:
            print(f"error {p}: {e}", flush=True)
            same = False
        finally:
            gl.close()
            gr.close()
        matched += int((uR >= 0).sum())
        if not same:
            bad += 1
            print(f"DIFFERENT {p}: {len(kL)} / {len(kR)} keypoints, {int((uR >= 0).sum())} matches", flush=True)
        if case % 10 == 9:
            print(f"{case + 1} cases, {bad} different, {matched} matches so far, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} cases, {bad} different, {matched} matches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
