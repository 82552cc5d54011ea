"""A second, line-by-line Python transcription of ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>, th, ...)
(/root/reference/src/ORBmatcher.cc:42-216) for frames with two cameras, used by the tests to check the loop exits
of the C++ oracle and of the library independently of both.

`skip_right_after_left_ratio_failure = True` is the reference: the `continue` at :125-126 sits in the map-point
loop (the candidate loop closed at :120), so a left best within TH_HIGH that fails the ratio test ends the map
point and its right-camera pass (:146-213) never runs.  False replays the misreading in which only the left
assignment is skipped; the tests assert that the two differ on their fixture.

Also builds that fixture: a left / right keypoint set in which many map points have two equally good left
candidates on one pyramid level while their right window holds one clear match.
"""
import numpy as np


def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def search_by_projection_two_cameras(gridL, kL, dL, gridR, kR, dR, ql, qr, qdesc, l2r, r2l, taken, marks, nnratio,
                                     skip_right_after_left_ratio_failure=True):
    """ql / qr: structured query arrays (x, y, radius, min_level, max_level); radius <= 0 = not in view in that
    camera.  Returns (kp_query, taken, nmatches, n_left_ratio_failures_with_right_candidates)."""
    nL, nR = len(kL), (len(kR) if gridR is not None else 0)
    taken = taken.copy()
    kp_query = np.full(nL + nR, -1, np.int32)
    nmatches = 0
    n_exit = 0
    for q in range(len(ql)):                                             # :48
        mark = 1 if marks is None else int(marks[q])
        if ql["radius"][q] > 0:                                          # :60 mbTrackInView
            cand = gridL.features_in_area(float(ql["x"][q]), float(ql["y"][q]), float(ql["radius"][q]),
                                          int(ql["min_level"][q]), int(ql["max_level"][q]))
            if len(cand):                                                # :73
                best, best_level, best2, best_level2, best_idx = 256, -1, 256, -1, -1
                for idx in cand:                                         # :84-120
                    if taken[idx]:
                        continue
                    d = hamming(qdesc[q], dL[idx])
                    if d < best:
                        best2, best, best_level2, best_level, best_idx = best, d, best_level, int(kL["octave"][idx]), int(idx)
                    elif d < best2:
                        best_level2, best2 = int(kL["octave"][idx]), d
                if best <= 100:                                          # :123
                    failed = best_level == best_level2 and best > np.float32(nnratio) * np.float32(best2)
                    if failed and skip_right_after_left_ratio_failure:   # :125-126, leaves the map-point iteration
                        if nR and qr["radius"][q] > 0 and len(gridR.features_in_area(
                                float(qr["x"][q]), float(qr["y"][q]), float(qr["radius"][q]),
                                int(qr["min_level"][q]), int(qr["max_level"][q]))):
                            n_exit += 1
                        continue
                    if not failed:                                       # :128-141
                        kp_query[best_idx] = q
                        taken[best_idx] = mark
                        if nR and l2r is not None and l2r[best_idx] != -1:
                            kp_query[l2r[best_idx] + nL] = q
                            taken[l2r[best_idx] + nL] = mark
                            nmatches += 1
                        nmatches += 1
        if nR and qr["radius"][q] > 0:                                   # :146-149
            cand = gridR.features_in_area(float(qr["x"][q]), float(qr["y"][q]), float(qr["radius"][q]),
                                          int(qr["min_level"][q]), int(qr["max_level"][q]))
            if len(cand) == 0:                                           # :156-157
                continue
            best, best_level, best2, best_level2, best_idx = 256, -1, 256, -1, -1
            for idx in cand:                                             # :168-193
                if taken[idx + nL]:
                    continue
                d = hamming(qdesc[q], dR[idx])
                if d < best:
                    best2, best, best_level2, best_level, best_idx = best, d, best_level, int(kR["octave"][idx]), int(idx)
                elif d < best2:
                    best_level2, best2 = int(kR["octave"][idx]), d
            if best <= 100:                                              # :196
                if best_level == best_level2 and best > np.float32(nnratio) * np.float32(best2):
                    continue                                             # :198-199
                if r2l is not None and r2l[best_idx] != -1:              # :201-206
                    kp_query[r2l[best_idx]] = q
                    taken[r2l[best_idx]] = mark
                    nmatches += 1
                kp_query[best_idx + nL] = q                              # :208-210
                taken[best_idx + nL] = mark
                nmatches += 1
    return kp_query, taken, nmatches, n_exit


def ambiguous_left_fixture(seed, n_points=400, kp_dtype=None, query_dtype=None):
    """Keypoints on a jittered lattice of a 512 x 512 image, all on octave 1.  Every map point p has one right
    keypoint (descriptor 12 bits from the map point's) and, for the first half of the points, TWO left keypoints
    two pixels apart whose descriptors are 20 and 22 bits away: with nnratio 0.8 the left ratio test fails
    (20 > 0.8 * 22) while the right window holds a clear match.  The second half has one left keypoint each."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n_points)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    pos = np.stack([gx.ravel(), gy.ravel()], 1)[:n_points].astype(np.float32) * (440.0 / side) + 36.0
    pos += rng.uniform(-3, 3, pos.shape).astype(np.float32)
    base = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)

    def flipped(d, nbits):
        out = d.copy()
        for b in rng.choice(256, nbits, replace=False):
            out[b >> 3] ^= np.uint8(1 << (b & 7))
        return out

    half = n_points // 2
    lrows, ldesc, rrows, rdesc = [], [], [], []
    for p in range(n_points):
        lrows.append((pos[p, 0], pos[p, 1]))
        ldesc.append(flipped(base[p], 20))
        if p < half:
            lrows.append((pos[p, 0] + 2.0, pos[p, 1] + 1.0))
            ldesc.append(flipped(base[p], 22))
        rrows.append((pos[p, 0] - 7.0, pos[p, 1] + 0.5))
        rdesc.append(flipped(base[p], 12))

    def kps(rows):
        k = np.zeros(len(rows), kp_dtype)
        k["x"] = [r[0] for r in rows]
        k["y"] = [r[1] for r in rows]
        k["octave"] = 1
        k["size"] = 37.0
        k["angle"] = rng.uniform(0, 360, len(rows)).astype(np.float32)
        k["response"] = 50.0
        k["class_id"] = -1
        return k

    kL, kR = kps(lrows), kps(rrows)
    dL, dR = np.stack(ldesc), np.stack(rdesc)
    order = rng.permutation(n_points)                    # map points in an order unrelated to the keypoints'
    ql = np.zeros(n_points, query_dtype)
    qr = np.zeros(n_points, query_dtype)
    ql["x"], ql["y"] = pos[order, 0] + 0.5, pos[order, 1]
    qr["x"], qr["y"] = pos[order, 0] - 7.0, pos[order, 1]
    for q_ in (ql, qr):
        q_["radius"] = np.float32(4.0) * np.float32(1.2)
        q_["min_level"] = 0
        q_["max_level"] = 1
    # stereo partners for a third of the left keypoints of the unambiguous half
    l2r = np.full(len(kL), -1, np.int32)
    r2l = np.full(len(kR), -1, np.int32)
    left_of_point = np.cumsum([0] + [2 if p < half else 1 for p in range(n_points)])[:-1]
    for p in range(half, n_points, 3):
        l2r[left_of_point[p]] = p
        r2l[p] = left_of_point[p]
    return kL, dL, kR, dR, ql, qr, base[order].copy(), l2r, r2l, half
