"""Random-case hunt for extraction differences between the HIP path and the CPU oracle (not collected by pytest).

    python tests/fuzz_extract.py [n_cases] [seed] [batch]

With "reuse" as third argument one handle (created for 2000 x 1300) serves ten consecutive cases of different
sizes, images, lapping areas and rectangles (geometry rebuilt on every size change, refusals in between).
With any other third argument the device-resident batch entry point is used instead of the one-image host entry point:
1..8 frames per call in a device tensor whose rows and frames are padded by random amounts.

Draws image size, pyramid shape, quota, thresholds, lapping area, erase rectangles and image kind at random, runs both
sides and prints every case that differs (with the parameters needed to replay it).  Exit status 1 if any differ."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from orb_slam3_amd import synth          # noqa: E402
import orb_slam3_amd as A                # noqa: E402
from oracle import oracle as O           # noqa: E402


def draw(rng, case):
    w = int(rng.integers(100, 2000))
    h = int(rng.integers(100, min(1300, w * 6 // 5) + 1))
    sf = float(rng.choice([1.1, 1.15, 1.2, 1.25, 1.33, 1.5, 1.8, 2.0]))
    nl = int(rng.integers(1, 9))
    while nl > 1 and min(w, h) / sf ** (nl - 1) < 12:
        nl -= 1
    nf = int(rng.integers(20, 6000))
    if nl <= 2:
        nf = min(nf, 2500)
    ini = int(rng.integers(1, 60))
    mn = int(rng.integers(0, ini + 1))
    lap = (0, 0) if rng.integers(0, 2) else (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w)))
    rects = []
    if rng.integers(0, 3) == 0:
        for _ in range(int(rng.integers(1, 5))):
            rx, ry = int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40))
            rects.append((rx, ry, int(rng.integers(10, w - rx)), int(rng.integers(10, h - ry))))
    kind = ["scene", "scene", "scene", "noise", "blocky", "flat+dots"][int(rng.integers(0, 6))]
    return dict(case=case, w=w, h=h, sf=sf, nl=nl, nf=nf, ini=ini, mn=mn, lap=lap, rects=rects, kind=kind,
                img_seed=int(rng.integers(1 << 30)))


def image(p):
    w, h = p["w"], p["h"]
    rng = np.random.default_rng(p["img_seed"])
    if p["kind"] == "scene":
        return synth.scene(w, h, p["img_seed"] % 100000, n_prims=int(rng.integers(10, 400)), noise_sigma=float(rng.uniform(0, 4)))
    if p["kind"] == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if p["kind"] == "blocky":
        b = int(rng.integers(3, 12))
        blocks = rng.integers(0, 256, ((h + b - 1) // b, (w + b - 1) // b), dtype=np.uint8)
        return np.kron(blocks, np.ones((b, b), np.uint8))[:h, :w].copy()
    img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    img[rng.random((h, w)) < 0.003] = int(rng.integers(0, 256))
    return img


def run_batch(g, p, imgs, rng):
    """The frames through extract_batch_device from a padded device tensor; returns a list of (kps, desc, mono)."""
    import torch
    nb, h, w = len(imgs), p["h"], p["w"]
    row = w + int(rng.integers(0, 70))
    rows = h + int(rng.integers(0, 5))
    buf = torch.zeros((nb, rows, row), dtype=torch.uint8, device="cuda")
    buf[:, :h, :w] = torch.from_numpy(np.stack(imgs)).cuda()
    view = buf[:, :h, :w]
    for attempt in range(2):
        kps, desc, counts = g.extract_batch_device(view, p["lap"])
        torch.cuda.synchronize()
        try:
            g.status()
            break
        except A.OrbxError as e:
            if e.status != -2 or attempt:
                raise
    kps, desc, counts = kps.cpu().numpy(), desc.cpu().numpy(), counts.cpu().numpy()
    out = []
    for f in range(nb):
        m = int(counts[f, 0])
        out.append((np.frombuffer(kps[f, :m].tobytes(), dtype=A.KP_DTYPE), desc[f, :m], int(counts[f, 1])))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reuse = len(sys.argv) > 3 and sys.argv[3] == "reuse"
    batch = len(sys.argv) > 3 and not reuse
    shared, fixed = None, None
    O.build()
    O.lib()
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for case in range(n):
        p = draw(rng, case)
        if reuse:
            if case % 10 == 0:
                if shared is not None:
                    shared.close()
                    shared = None
                fixed = {k: p[k] for k in ("nf", "sf", "nl", "ini", "mn")}
                try:
                    shared = A.ORBextractor(p["nf"], p["sf"], p["nl"], p["ini"], p["mn"], max_width=2000, max_height=1300)
                except A.OrbxError as e:
                    print(f"refused {p}: {e}", flush=True)
            p.update(fixed)
            if shared is None:
                continue
            img = image(p)
            shared.mvDynamicArea = p["rects"]
            try:
                try:
                    kg, dg, mg = shared(img, p["lap"])
                except A.OrbxError as e:
                    if e.status != -2:
                        raise
                    kg, dg, mg = shared(img, p["lap"])
            except A.OrbxError as e:
                print(f"refused {p}: {e}", flush=True)      # e.g. a top level of a few pixels for this size
                continue
            ko, do, mo = O.Extractor(p["nf"], p["sf"], p["nl"], p["ini"], p["mn"])(img, p["lap"], p["rects"])
            same = mo == mg and len(ko) == len(kg) and np.array_equal(do, dg) and all(
                np.array_equal(ko[x], kg[x]) for x in ("x", "y", "size", "angle", "response", "octave", "class_id"))
            if not same:
                bad += 1
                print(f"DIFFERENT (shared handle) {p}: oracle {len(ko)} keypoints, handle {len(kg)}", flush=True)
            if case % 10 == 9:
                print(f"{case + 1} cases, {bad} different, {time.time() - t0:.0f} s", flush=True)
            continue
        img = image(p)
        o = O.Extractor(p["nf"], p["sf"], p["nl"], p["ini"], p["mn"])
        nb = int(rng.integers(1, 9)) if batch else 1
        try:
            g = A.ORBextractor(p["nf"], p["sf"], p["nl"], p["ini"], p["mn"], max_width=p["w"], max_height=p["h"], max_batch=nb)
        except A.OrbxError as e:
            print(f"refused {p}: {e}", flush=True)
            continue
        g.mvDynamicArea = p["rects"]
        if batch:
            imgs = [img]
            for f in range(1, nb):
                q = dict(p)
                q["img_seed"] = p["img_seed"] + f
                imgs.append(image(q))
            try:
                got = run_batch(g, p, imgs, rng)
                same = True
                for f in range(nb):
                    ko, do, mo = o(imgs[f], p["lap"], p["rects"])
                    kg, dg, mg = got[f]
                    same = same and mo == mg and len(ko) == len(kg) and np.array_equal(do, dg) and all(
                        np.array_equal(ko[x], kg[x]) for x in ("x", "y", "size", "angle", "response", "octave", "class_id"))
            except A.OrbxError as e:
                print(f"error {p}: {e}", flush=True)
                same = False
            finally:
                g.close()
            if not same:
                bad += 1
                print(f"DIFFERENT (batch of {nb}) {p}", flush=True)
            if case % 10 == 9:
                print(f"{case + 1} cases, {bad} different, {time.time() - t0:.0f} s", flush=True)
            continue
        ko, do, mo = o(img, p["lap"], p["rects"])
        try:
            try:
                kg, dg, mg = g(img, p["lap"])
            except A.OrbxError as e:
                if e.status != -2:
                    raise
                kg, dg, mg = g(img, p["lap"])          # buffers regrown by the first call
            same = mo == mg and len(ko) == len(kg) and np.array_equal(do, dg) and all(
                np.array_equal(ko[f], kg[f]) for f in ("x", "y", "size", "angle", "response", "octave", "class_id"))
        except A.OrbxError as e:
            print(f"error {p}: {e}", flush=True)
            same = False
        finally:
            g.close()
        if not same:
            bad += 1
            print(f"DIFFERENT {p}: oracle {len(ko)} keypoints (mono {mo})", flush=True)
        if case % 10 == 9:
            print(f"{case + 1} cases, {bad} different, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} cases, {bad} different")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
