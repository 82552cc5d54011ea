"""Python mirror of the matching side of the path over the C ABI:
ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:34-226) for the searches whose data-parallel
core runs on the GPU, Frame::ComputeStereoMatches and the fisheye brute-force kNN.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import KP_DTYPE, MATCH2_DTYPE, QUERY_DTYPE, check, lib, ptr

TH_HIGH = 100   # ORBmatcher.cc:34
TH_LOW = 50     # ORBmatcher.cc:35
HISTO_LENGTH = 30


def _desc(a):
    a = np.ascontiguousarray(a, np.uint8)
    assert a.ndim == 2 and a.shape[1] == 32
    return a


class ORBmatcher:
    """ORBmatcher(nnratio = 0.6, checkOri = true)  (ORBmatcher.cc:38)."""

    TH_HIGH = TH_HIGH
    TH_LOW = TH_LOW
    HISTO_LENGTH = HISTO_LENGTH

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True):
        self.mfNNratio = float(np.float32(nnratio))
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray):
        """ORBmatcher::DescriptorDistance (ORBmatcher.cc:2141): a, b are [32] or [n, 32] uint8."""
        a2 = _desc(np.atleast_2d(a))
        b2 = _desc(np.atleast_2d(b))
        assert a2.shape == b2.shape
        out = np.zeros(len(a2), np.int32)
        check(lib().orbx_descriptor_distance(ptr(a2), ptr(b2), len(a2), ptr(out)))
        return int(out[0]) if np.ndim(a) == 1 else out

    def SearchByProjection(self, kps, desc, uright, bounds, queries, query_desc, taken):
        """Frame / map-point variant (ORBmatcher.cc:42-147), mono or rectified stereo.
        kps: frame keypoints (KP_DTYPE, undistorted), desc [n,32]; uright: float32 [n] or None;
        bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY); queries: QUERY_DTYPE (projection, radius
        r*scale[level], level window [level-1, level]); taken: uint8 [n], updated in place.
        Returns (match index per query or -1, number of matches)."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = _desc(desc)
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        query_desc = _desc(query_desc) if len(queries) else np.zeros((0, 32), np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == len(kps)
        out = np.full(len(queries), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_by_projection(ptr(kps), ptr(desc), ptr(ur), len(kps), *map(float, bounds),
                                              ptr(queries), ptr(query_desc), len(queries), self.mfNNratio,
                                              ptr(taken), ptr(out), C.byref(nm)))
        return out, nm.value


def window_candidates(kps, desc, uright, bounds, queries, query_desc):
    """Ordered candidates of Frame::GetFeaturesInArea per query with their distances (CSR)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = _desc(desc)
    queries = np.ascontiguousarray(queries, QUERY_DTYPE)
    query_desc = _desc(query_desc) if len(queries) else np.zeros((0, 32), np.uint8)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    off = np.zeros(len(queries) + 1, np.int32)
    cap = max(1, len(kps)) * max(1, len(queries))
    cap = min(cap, 1 << 24)
    idx = np.zeros(cap, np.int32)
    dist = np.zeros(cap, np.int32)
    n = C.c_int()
    check(lib().orbx_window_candidates(ptr(kps), ptr(desc), ptr(ur), len(kps), *map(float, bounds), ptr(queries),
                                       ptr(query_desc), len(queries), ptr(off), ptr(idx), ptr(dist), cap, C.byref(n)))
    return off, idx[:n.value].copy(), dist[:n.value].copy()


def knn_match2(query, train) -> np.ndarray:
    """cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, 2) (Frame.cc:1309): MATCH2_DTYPE rows."""
    q, t = _desc(query), _desc(train) if len(train) else np.zeros((0, 32), np.uint8)
    out = np.zeros(len(q), MATCH2_DTYPE)
    check(lib().orbx_hamming_knn2(ptr(q), len(q), ptr(t), len(t), ptr(out)))
    return out


def knn_match2_device(d_query, d_train, base_index: int = 0, out=None, workspace=None):
    """Device-resident kNN(2): torch uint8 CUDA tensors [nq,32] / [nt,32]; returns an int32
    [nq,4] tensor (idx0, dist0, idx1, dist1) with indices offset by base_index."""
    import torch
    nq, nt = d_query.shape[0], d_train.shape[0]
    L = lib()
    ws_bytes = L.orbx_knn2_workspace_bytes(nq, nt)
    if workspace is None or workspace.numel() < ws_bytes:
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=d_query.device)
    if out is None:
        out = torch.empty((nq, 4), dtype=torch.int32, device=d_query.device)
    st = torch.cuda.current_stream(d_query.device).cuda_stream
    check(L.orbx_hamming_knn2_device(d_query.data_ptr(), nq, d_train.data_ptr(), nt, base_index, out.data_ptr(),
                                     workspace.data_ptr(), workspace.numel(), st))
    return out, workspace


def knn2_merge_device(d_parts, out=None):
    """d_parts: int32 CUDA tensor [n_parts, nq, 4] (e.g. all-gathered per-GPU shard results)."""
    import torch
    nparts, nq, _ = d_parts.shape
    if out is None:
        out = torch.empty((nq, 4), dtype=torch.int32, device=d_parts.device)
    st = torch.cuda.current_stream(d_parts.device).cuda_stream
    check(lib().orbx_knn2_merge_device(d_parts.data_ptr(), nparts, nq, out.data_ptr(), st))
    return out


def fisheye_ratio_matches(m2: np.ndarray, ratio: float = 0.7) -> np.ndarray:
    """The ratio test of Frame::ComputeStereoFishEyeMatches (Frame.cc:1316-1321):
    keeps queries with >= 2 neighbours and dist0 < dist1 * 0.7 (double arithmetic)."""
    ok = (m2["idx1"] >= 0) & (m2["dist0"].astype(np.float64) < m2["dist1"].astype(np.float64) * ratio)
    return np.nonzero(ok)[0]


def compute_stereo_matches(left, right, kpsL, descL, kpsR, descR, mbf: float, mb: float):
    """Frame::ComputeStereoMatches (Frame.cc:962-1133) on the pyramids of two ORBextractor
    handles (batch slot 0).  Returns (mvuRight, mvDepth)."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL, descR = _desc(descL), _desc(descR)
    ur = np.full(len(kpsL), -1, np.float32)
    depth = np.full(len(kpsL), -1, np.float32)
    check(lib().orbx_stereo_match(left.handle, right.handle, 0, ptr(kpsL), ptr(descL), len(kpsL), ptr(kpsR),
                                  ptr(descR), len(kpsR), mbf, mb, ptr(ur), ptr(depth)))
    return ur, depth


def compute_stereo_matches_device(left, right, outL, outR, mbf: float, mb: float):
    """Batched device variant: outL / outR are the (kps, desc, counts) tensors returned by
    ORBextractor.extract_batch_device on the two handles.  Returns (uRight, depth) float32
    CUDA tensors [n, cap]."""
    import torch
    kL, dL, cL = outL
    kR, dR, cR = outR
    nf, cap = kL.shape[0], kL.shape[1]
    ur = torch.empty((nf, cap), dtype=torch.float32, device=kL.device)
    depth = torch.empty((nf, cap), dtype=torch.float32, device=kL.device)
    st = torch.cuda.current_stream(kL.device).cuda_stream
    check(lib().orbx_stereo_match_batch_device(left.handle, right.handle, nf, kL.data_ptr(), dL.data_ptr(),
                                               cL.data_ptr(), kR.data_ptr(), dR.data_ptr(), cR.data_ptr(), cap,
                                               mbf, mb, ur.data_ptr(), depth.data_ptr(), st))
    return ur, depth


def bucket_best(desc1, idx1, off1, desc2, idx2, off2):
    """Distance core of ORBmatcher::SearchForTriangulation (ORBmatcher.cc:1051-1167)."""
    desc1, desc2 = _desc(desc1), _desc(desc2)
    idx1 = np.ascontiguousarray(idx1, np.int32)
    idx2 = np.ascontiguousarray(idx2, np.int32)
    off1 = np.ascontiguousarray(off1, np.int32)
    off2 = np.ascontiguousarray(off2, np.int32)
    bi = np.full(len(idx1), -1, np.int32)
    bd = np.full(len(idx1), 50, np.int32)
    check(lib().orbx_bucket_best(ptr(desc1), len(desc1), ptr(idx1), ptr(off1), ptr(desc2), len(desc2), ptr(idx2),
                                 ptr(off2), len(off1) - 1, ptr(bi), ptr(bd)))
    return bi, bd
