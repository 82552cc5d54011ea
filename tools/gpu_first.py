"""First GPU bring-up: stage-by-stage comparison of the HIP extractor with the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O
import orb_slam3_amd as A
from orb_slam3_amd import synth

def cmp_extract(w, h, nf, seed, lap):
    img = synth.scene(w, h, seed)
    o = O.Extractor(nf, 1.2, 8, 20, 7)
    ko, do, mo = o(img, lap)
    g = A.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    assert np.array_equal(g.mvScaleFactor, o.scale_factors) and np.array_equal(g.mnFeaturesPerLevel, o.features_per_level)
    kg, dg, mg = g(img, lap)
    ok = True
    for l in range(8):
        pl_o, pl_g = o.pyramid_level(l), g.pyramid_level(l)
        same = pl_o.shape == pl_g.shape and np.array_equal(pl_o, pl_g)
        co = o.level_candidates(l)
        cg, order = g.debug_level_candidates(l)
        csame = co.shape == cg.shape and np.array_equal(co, cg)
        lo = o.level_keypoints(l)
        lg = g.debug_level_keypoints(l)
        lsame = len(lo) == len(lg) and np.array_equal(lo['x'], lg[:,0]) and np.array_equal(lo['y'], lg[:,1]) and np.array_equal(lo['response'], lg[:,2])
        print(f"  level {l}: pyr {same}  cands {len(co)}/{len(cg)} {csame}  kps {len(lo)}/{len(lg)} {lsame}")
        ok &= same and csame and lsame
        if not csame and len(co) and len(cg):
            n = min(len(co), len(cg))
            bad = np.nonzero(~np.all(co[:n] == cg[:n], axis=1))[0]
            print("    first cand mismatches:", bad[:5], co[bad[:3]], cg[bad[:3]])
    print(f" final: n {len(ko)}/{len(kg)} mono {mo}/{mg}")
    if len(ko) == len(kg):
        for f in ('x','y','size','angle','response','octave','class_id'):
            eq = np.array_equal(ko[f], kg[f])
            print(f"   {f}: {eq}", '' if eq else np.abs(ko[f].astype(np.float64)-kg[f]).max())
            ok &= eq
        deq = np.array_equal(do, dg)
        print("   desc:", deq, '' if deq else (do != dg).any(axis=1).sum())
        ok &= deq
    else:
        ok = False
    return ok

if __name__ == '__main__':
    print('devices', A.lib().orbx_device_count())
    allok = True
    for (w, h, nf, seed, lap) in [(752, 480, 1000, 1, (0, 1000)), (752, 480, 1200, 2, (0, 0)), (640, 480, 500, 3, (100, 300)), (1241, 376, 2000, 5, (0,0))]:
        print('case', w, h, nf, seed, lap)
        allok &= cmp_extract(w, h, nf, seed, lap)
    # knn2
    q = synth.random_descriptors(300, 5); t, rows = synth.planted_map(q, 20000, 6)
    a = O.knn2(q, t); b = A.knn_match2(q, t)
    bb = np.stack([b['idx0'], b['dist0'], b['idx1'], b['dist1']], 1)
    print('knn2 equal', np.array_equal(a, bb)); allok &= np.array_equal(a, bb)
    print('ALL OK' if allok else 'MISMATCH')
