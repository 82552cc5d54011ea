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


class Camera(C.Structure):
    """Pinhole camera with radial-tangential distortion: K and DistCoef = (k1, k2, p1, p2[, k3[, k4, k5, k6]])."""
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("dist", C.c_float * 8),
                ("n_dist", C.c_int)]

    def __init__(self, fx, fy, cx, cy, dist=()):
        super().__init__(fx, fy, cx, cy, (C.c_float * 8)(*[float(d) for d in dist]), len(dist))

    def image_bounds(self, width, height):
        """Frame::ComputeImageBounds (Frame.cc:916-947)."""
        b = np.zeros(4, np.float32)
        check(lib().orbx_image_bounds(C.byref(self), int(width), int(height), ptr(b)))
        return b


class FrameFeatures:
    """What a Frame / KeyFrame holds for the windowed searches, resident on the device:
    mvKeysUn, mDescriptors, mvuRight and the 64 x 48 mGrid of Frame::AssignFeaturesToGrid
    (Frame.cc:468-506), which is built on the device.  bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY)."""

    def __init__(self, kps=None, desc=None, uright=None, bounds=None):
        h = C.c_void_p()
        check(lib().orbx_frame_create(C.byref(h)))
        self._h = h
        if kps is not None:
            self.assign(kps, desc, uright, bounds)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().orbx_frame_destroy(h)
            except Exception:
                pass

    def __len__(self):
        return int(lib().orbx_frame_size(self._h))

    def assign(self, kps, desc, uright, bounds):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = _desc(desc) if len(kps) else np.zeros((0, 32), np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        assert len(desc) == len(kps) and (ur is None or len(ur) == len(kps))
        check(lib().orbx_frame_assign(self._h, ptr(kps), ptr(desc), ptr(ur), len(kps), *map(float, bounds)))
        return self

    def assign_device(self, d_kps, d_desc, d_uright, n, bounds):
        """d_kps: torch uint8/byte tensor of n KP_DTYPE records, d_desc [n, 32] uint8, d_uright float32 [n] or None,
        all on the current device (for instance one frame's slice of the batched extractor output)."""
        check(lib().orbx_frame_assign_device(self._h, C.c_void_p(d_kps.data_ptr()), C.c_void_p(d_desc.data_ptr()),
                                             None if d_uright is None else C.c_void_p(d_uright.data_ptr()), int(n),
                                             *map(float, bounds)))
        return self

    def assign_distorted(self, camera: Camera, width, height, kps, desc, uright=None):
        """Frame::UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid on the device: kps are mvKeys."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = _desc(desc) if len(kps) else np.zeros((0, 32), np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        check(lib().orbx_frame_assign_distorted(self._h, C.byref(camera), int(width), int(height), ptr(kps), ptr(desc),
                                                ptr(ur), len(kps)))
        return self

    def assign_distorted_device(self, camera: Camera, width, height, d_kps, d_desc, d_uright, n):
        check(lib().orbx_frame_assign_distorted_device(self._h, C.byref(camera), int(width), int(height),
                                                       C.c_void_p(d_kps.data_ptr()), C.c_void_p(d_desc.data_ptr()),
                                                       None if d_uright is None else C.c_void_p(d_uright.data_ptr()),
                                                       int(n)))
        return self

    def keypoints(self):
        """(mvKeysUn, (mnMinX, mnMaxX, mnMinY, mnMaxY)) as the frame holds them."""
        kps = np.zeros(max(len(self), 1), KP_DTYPE)
        b = np.zeros(4, np.float32)
        check(lib().orbx_frame_keypoints(self._h, ptr(kps), len(kps), ptr(b)))
        return kps[:len(self)].copy(), b

    def grid(self):
        """(cell_start[64 * 48 + 1], cell_idx[]) with cell = column * 48 + row."""
        start = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(max(len(self), 1), np.int32)
        n = C.c_int()
        check(lib().orbx_frame_grid(self._h, ptr(start), ptr(idx), len(idx), C.byref(n)))
        return start, idx[:n.value].copy()

    def window_candidates(self, queries, query_desc):
        """GetFeaturesInArea + distance per query: (offset[nq + 1], idx[], dist[])."""
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        query_desc = _desc(query_desc) if len(queries) else np.zeros((0, 32), np.uint8)
        off = np.zeros(len(queries) + 1, np.int32)
        n = C.c_int()
        cap = max(64 * len(queries), 1)
        while True:
            idx = np.zeros(cap, np.int32)
            dist = np.zeros(cap, np.int32)
            st = lib().orbx_frame_window_candidates(self._h, ptr(queries), ptr(query_desc), len(queries), ptr(off),
                                                    ptr(idx), ptr(dist), cap, C.byref(n))
            if st == 0 or n.value <= cap:
                check(st)
                return off, idx[:n.value].copy(), dist[:n.value].copy()
            cap = n.value


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

    # ---- the same searches on a frame whose features are already on the device --------------
    def SearchByProjectionIn(self, frame: FrameFeatures, queries, query_desc, taken):
        """SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (ORBmatcher.cc:42-147) on a FrameFeatures."""
        queries, query_desc = self._queries(queries, query_desc)
        assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == len(frame)
        out = np.full(len(queries), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_by_projection(frame._h, ptr(queries), ptr(query_desc), len(queries),
                                                    self.mfNNratio, ptr(taken), ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchWindowBestIn(self, frame: FrameFeatures, queries, query_desc, max_dist, taken=None, query_marks=None,
                           query_angle=None, rot_factor=1.0 / HISTO_LENGTH):
        """SearchWindowBest on a FrameFeatures."""
        queries, query_desc = self._queries(queries, query_desc)
        if taken is not None:
            assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == len(frame)
        marks = None if query_marks is None else np.ascontiguousarray(query_marks, np.uint8)
        qa = None
        if query_angle is not None and self.mbCheckOrientation:
            qa = np.ascontiguousarray(query_angle, np.float32)
        out = np.full(len(queries), -1, np.int32)
        kpm = np.full(len(frame), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_window_best(frame._h, ptr(queries), ptr(query_desc), len(queries), float(max_dist),
                                                  ptr(taken), ptr(marks), ptr(qa), float(np.float32(rot_factor)),
                                                  ptr(out), ptr(kpm), C.byref(nm)))
        return out, kpm, nm.value

    def FuseIn(self, frame: FrameFeatures, queries, query_desc, inv_level_sigma2):
        """Fuse on a FrameFeatures."""
        queries, query_desc = self._queries(queries, query_desc)
        sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
        out = np.full(len(queries), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_fuse_search(frame._h, ptr(queries), ptr(query_desc), len(queries), ptr(sig), len(sig),
                                           ptr(out), C.byref(nm)))
        return out, nm.value

    # ---- the other order-dependent searches (flattened inputs, see include/orbx.h) ----------
    @staticmethod
    def _frame(kps, desc, uright):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        return kps, _desc(desc) if len(kps) else np.zeros((0, 32), np.uint8), \
            None if uright is None else np.ascontiguousarray(uright, np.float32)

    @staticmethod
    def _queries(queries, query_desc):
        queries = np.ascontiguousarray(queries, QUERY_DTYPE)
        return queries, _desc(query_desc) if len(queries) else np.zeros((0, 32), np.uint8)

    def SearchWindowBest(self, kps, desc, uright, bounds, queries, query_desc, max_dist, taken=None,
                         query_marks=None, query_angle=None, rot_factor=1.0 / HISTO_LENGTH):
        """Loop of SearchByProjection(Cur, Last) / (Cur, KF, sFound) / (KF, Sim3, points) and
        Fuse(KF, Sim3, points) (ORBmatcher.cc:1761-2094, :476-693, :1409-1524).  query_angle is
        used only when the matcher checks orientation.  Returns (match per query, query per
        keypoint, number of matches as the reference counts them); taken is updated in place."""
        kps, desc, ur = self._frame(kps, desc, uright)
        queries, query_desc = self._queries(queries, query_desc)
        if taken is not None:
            assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == len(kps)
        marks = None if query_marks is None else np.ascontiguousarray(query_marks, np.uint8)
        qa = None
        if query_angle is not None and self.mbCheckOrientation:
            qa = np.ascontiguousarray(query_angle, np.float32)
        out = np.full(len(queries), -1, np.int32)
        kpm = np.full(len(kps), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_window_best(ptr(kps), ptr(desc), ptr(ur), len(kps), *map(float, bounds), ptr(queries),
                                            ptr(query_desc), len(queries), float(max_dist), ptr(taken), ptr(marks),
                                            ptr(qa), float(np.float32(rot_factor)), ptr(out), ptr(kpm), C.byref(nm)))
        return out, kpm, nm.value

    def Fuse(self, kps, desc, uright, bounds, queries, query_desc, inv_level_sigma2):
        """Matching part of Fuse(KF, MapPoints, th, bRight) (ORBmatcher.cc:1214-1407)."""
        kps, desc, ur = self._frame(kps, desc, uright)
        queries, query_desc = self._queries(queries, query_desc)
        sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
        out = np.full(len(queries), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_fuse_search(ptr(kps), ptr(desc), ptr(ur), len(kps), *map(float, bounds), ptr(queries),
                                     ptr(query_desc), len(queries), ptr(sig), len(sig), ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchBySim3(self, kps1, desc1, bounds1, kps2, desc2, bounds2, into2, into1):
        """SearchBySim3 (ORBmatcher.cc:1526-1743).  into2 = (queries, descriptors, source keypoint
        of KF1) for KF1's map points projected into KF2; into1 the converse."""
        kps1, desc1, _ = self._frame(kps1, desc1, None)
        kps2, desc2, _ = self._frame(kps2, desc2, None)
        q2, d2 = self._queries(into2[0], into2[1])
        q1, d1 = self._queries(into1[0], into1[1])
        s1 = np.ascontiguousarray(into2[2], np.int32)
        s2 = np.ascontiguousarray(into1[2], np.int32)
        b1 = np.ascontiguousarray(bounds1, np.float32)
        b2 = np.ascontiguousarray(bounds2, np.float32)
        out = np.full(len(kps1), -1, np.int32)
        nf = C.c_int()
        check(lib().orbx_search_by_sim3(ptr(kps1), ptr(desc1), len(kps1), ptr(b1), ptr(kps2), ptr(desc2), len(kps2),
                                        ptr(b2), ptr(q2), ptr(d2), ptr(s1), len(q2), ptr(q1), ptr(d1), ptr(s2), len(q1),
                                        ptr(out), C.byref(nf)))
        return out, nf.value

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds, prev_matched, windowSize=10):
        """SearchForInitialization (ORBmatcher.cc:695-808); prev_matched [n1, 2] float32 is
        updated in place.  Returns (vnMatches12, number of matches)."""
        kps1, desc1, _ = self._frame(kps1, desc1, None)
        kps2, desc2, _ = self._frame(kps2, desc2, None)
        assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(kps1), 2)
        out = np.full(len(kps1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_for_initialization(ptr(kps1), ptr(desc1), len(kps1), ptr(kps2), ptr(desc2), len(kps2),
                                                   *map(float, bounds), ptr(prev_matched), int(windowSize),
                                                   self.mfNNratio, int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchByBoW(self, descKF, angleKF, validKF, featvecKF, descF, angleF, featvecF, n_left=-1):
        """SearchByBoW(KeyFrame, Frame) (ORBmatcher.cc:235-474).  featvec = (idx, off) CSR over the
        vocabulary nodes both feature vectors share.  Returns (KF keypoint per frame keypoint, n)."""
        descKF, descF = _desc(descKF), _desc(descF)
        a = [np.ascontiguousarray(angleKF, np.float32), np.ascontiguousarray(validKF, np.uint8),
             np.ascontiguousarray(featvecKF[0], np.int32), np.ascontiguousarray(featvecKF[1], np.int32)]
        b = [np.ascontiguousarray(angleF, np.float32), np.ascontiguousarray(featvecF[0], np.int32),
             np.ascontiguousarray(featvecF[1], np.int32)]
        out = np.full(len(descF), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_by_bow(ptr(descKF), ptr(a[0]), ptr(a[1]), len(descKF), ptr(a[2]), ptr(a[3]), ptr(descF),
                                       ptr(b[0]), len(descF), int(n_left), ptr(b[1]), ptr(b[2]), len(a[3]) - 1,
                                       self.mfNNratio, int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchByBoWKeyFrames(self, desc1, angle1, valid1, featvec1, desc2, angle2, valid2, featvec2):
        """SearchByBoW(KeyFrame, KeyFrame) (ORBmatcher.cc:824-966)."""
        desc1, desc2 = _desc(desc1), _desc(desc2)
        a = [np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(valid1, np.uint8),
             np.ascontiguousarray(featvec1[0], np.int32), np.ascontiguousarray(featvec1[1], np.int32)]
        b = [np.ascontiguousarray(angle2, np.float32), np.ascontiguousarray(valid2, np.uint8),
             np.ascontiguousarray(featvec2[0], np.int32), np.ascontiguousarray(featvec2[1], np.int32)]
        out = np.full(len(desc1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_by_bow_keyframes(ptr(desc1), ptr(a[0]), ptr(a[1]), len(desc1), ptr(a[2]), ptr(a[3]),
                                                 ptr(desc2), ptr(b[0]), ptr(b[1]), len(desc2), ptr(b[2]), ptr(b[3]),
                                                 len(a[3]) - 1, self.mfNNratio, int(self.mbCheckOrientation),
                                                 ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchForTriangulation(self, kps1, desc1, free1, uright1, featvec1, kps2, desc2, free2, uright2, featvec2,
                               F12, epipole, scale_factors2, level_sigma2_2, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation (ORBmatcher.cc:968-1212), pinhole cameras."""
        kps1, desc1, ur1 = self._frame(kps1, desc1, uright1)
        kps2, desc2, ur2 = self._frame(kps2, desc2, uright2)
        a = [np.ascontiguousarray(free1, np.uint8), np.ascontiguousarray(featvec1[0], np.int32),
             np.ascontiguousarray(featvec1[1], np.int32)]
        b = [np.ascontiguousarray(free2, np.uint8), np.ascontiguousarray(featvec2[0], np.int32),
             np.ascontiguousarray(featvec2[1], np.int32)]
        sf = np.ascontiguousarray(scale_factors2, np.float32)
        sg = np.ascontiguousarray(level_sigma2_2, np.float32)
        geom = Epipolar()
        geom.F12[:] = [float(v) for v in np.asarray(F12, np.float32).reshape(9)]
        geom.epipole_x, geom.epipole_y = float(epipole[0]), float(epipole[1])
        geom.scale_factors2 = sf.ctypes.data_as(C.POINTER(C.c_float))
        geom.level_sigma2_2 = sg.ctypes.data_as(C.POINTER(C.c_float))
        geom.n_levels = len(sf)
        out = np.full(len(kps1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_for_triangulation(ptr(kps1), ptr(desc1), ptr(a[0]), ptr(ur1), len(kps1), ptr(a[1]),
                                                  ptr(a[2]), ptr(kps2), ptr(desc2), ptr(b[0]), ptr(ur2), len(kps2),
                                                  ptr(b[1]), ptr(b[2]), len(a[2]) - 1, C.byref(geom), int(bOnlyStereo),
                                                  int(bCoarse), int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value


    # ---- frames with a second camera (Frame::Nleft != -1) -----------------------------------
    def SearchByProjectionStereoIn(self, left: FrameFeatures, right, queries_left, queries_right, query_desc, taken,
                                   left_to_right=None, right_to_left=None, query_marks=None):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) with the left and the right camera pass
        (ORBmatcher.cc:42-216).  right = FrameFeatures of the right camera or None; taken spans left + right
        keypoints and is updated in place.  Returns (kp_query, n_matches)."""
        ql, query_desc = self._queries(queries_left, query_desc)
        qr = None if right is None else np.ascontiguousarray(queries_right, QUERY_DTYPE)
        n = len(left) + (0 if right is None else len(right))
        assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == n
        l2r = None if left_to_right is None else np.ascontiguousarray(left_to_right, np.int32)
        r2l = None if right_to_left is None else np.ascontiguousarray(right_to_left, np.int32)
        marks = None if query_marks is None else np.ascontiguousarray(query_marks, np.uint8)
        kpq = np.full(n, -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_by_projection_stereo(left._h, None if right is None else right._h, ptr(ql), ptr(qr),
                                                           ptr(query_desc), len(ql), float(self.mfNNratio), ptr(l2r),
                                                           ptr(r2l), ptr(taken), ptr(marks), ptr(kpq), C.byref(nm)))
        return kpq, nm.value

    def SearchWindowBestStereoIn(self, left: FrameFeatures, right, queries_left, queries_right, query_desc, max_dist,
                                 taken, query_marks=None, query_angle=None, rot_factor=1.0 / HISTO_LENGTH):
        """SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) with the right-image pass
        (ORBmatcher.cc:1761-1972).  Returns (kp_query, n_matches)."""
        ql, query_desc = self._queries(queries_left, query_desc)
        qr = None if right is None else np.ascontiguousarray(queries_right, QUERY_DTYPE)
        n = len(left) + (0 if right is None else len(right))
        assert taken.dtype == np.uint8 and taken.flags.c_contiguous and len(taken) == n
        marks = None if query_marks is None else np.ascontiguousarray(query_marks, np.uint8)
        qa = None
        if query_angle is not None and self.mbCheckOrientation:
            qa = np.ascontiguousarray(query_angle, np.float32)
        kpq = np.full(n, -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_window_best_stereo(left._h, None if right is None else right._h, ptr(ql), ptr(qr),
                                                         ptr(query_desc), len(ql), float(max_dist), ptr(taken), ptr(marks),
                                                         ptr(qa), float(np.float32(rot_factor)), ptr(kpq), C.byref(nm)))
        return kpq, nm.value

    def SearchForTriangulationCameras(self, kps1, desc1, free1, uright1, n_left1, featvec1, kps2, desc2, free2, uright2,
                                      n_left2, featvec2, cam_types, cam_params, poses, epipole, scale_factors2,
                                      level_sigma2_1, level_sigma2_2, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation with the per-candidate camera pair (ORBmatcher.cc:968-1212 incl. :1099-1139).
        cam_types[4] / cam_params[4][8] = {cam1 left, cam1 right, cam2 left, cam2 right};
        poses[4][12] = (R12 row-major, t12) for ll, lr, rl, rr."""
        kps1, desc1, ur1 = self._frame(kps1, desc1, uright1)
        kps2, desc2, ur2 = self._frame(kps2, desc2, uright2)
        a = [np.ascontiguousarray(free1, np.uint8), np.ascontiguousarray(featvec1[0], np.int32),
             np.ascontiguousarray(featvec1[1], np.int32)]
        b = [np.ascontiguousarray(free2, np.uint8), np.ascontiguousarray(featvec2[0], np.int32),
             np.ascontiguousarray(featvec2[1], np.int32)]
        g, keep = triangulation_geometry(n_left1, n_left2, cam_types, cam_params, poses, epipole, scale_factors2,
                                         level_sigma2_1, level_sigma2_2)
        out = np.full(len(kps1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_search_for_triangulation_cameras(ptr(kps1), ptr(desc1), ptr(a[0]), ptr(ur1), len(kps1), ptr(a[1]),
                                                          ptr(a[2]), ptr(kps2), ptr(desc2), ptr(b[0]), ptr(ur2), len(kps2),
                                                          ptr(b[1]), ptr(b[2]), len(a[2]) - 1, C.byref(g), int(bOnlyStereo),
                                                          int(bCoarse), int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value


    # ---- the bucketed searches on resident frames ---------------------------------------------
    def SearchByBoWIn(self, kf: FrameFeatures, validKF, featvecKF, f: FrameFeatures, featvecF, n_left=-1):
        """SearchByBoW(KeyFrame, Frame) with both descriptor sets resident on the device."""
        a = [np.ascontiguousarray(validKF, np.uint8), np.ascontiguousarray(featvecKF[0], np.int32),
             np.ascontiguousarray(featvecKF[1], np.int32)]
        b = [np.ascontiguousarray(featvecF[0], np.int32), np.ascontiguousarray(featvecF[1], np.int32)]
        out = np.full(len(f), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_by_bow(kf._h, ptr(a[0]), ptr(a[1]), ptr(a[2]), f._h, int(n_left), ptr(b[0]), ptr(b[1]),
                                             len(a[2]) - 1, self.mfNNratio, int(self.mbCheckOrientation), ptr(out),
                                             C.byref(nm)))
        return out, nm.value

    def SearchByBoWKeyFramesIn(self, kf1: FrameFeatures, valid1, featvec1, kf2: FrameFeatures, valid2, featvec2):
        a = [np.ascontiguousarray(valid1, np.uint8), np.ascontiguousarray(featvec1[0], np.int32),
             np.ascontiguousarray(featvec1[1], np.int32)]
        b = [np.ascontiguousarray(valid2, np.uint8), np.ascontiguousarray(featvec2[0], np.int32),
             np.ascontiguousarray(featvec2[1], np.int32)]
        out = np.full(len(kf1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_by_bow_keyframes(kf1._h, ptr(a[0]), ptr(a[1]), ptr(a[2]), kf2._h, ptr(b[0]), ptr(b[1]),
                                                       ptr(b[2]), len(a[2]) - 1, self.mfNNratio,
                                                       int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value

    def SearchForTriangulationIn(self, kf1: FrameFeatures, free1, featvec1, kf2: FrameFeatures, free2, featvec2, geometry,
                                 bOnlyStereo=False, bCoarse=False):
        """geometry = triangulation_geometry(...)."""
        g, keep = geometry
        a = [np.ascontiguousarray(free1, np.uint8), np.ascontiguousarray(featvec1[0], np.int32),
             np.ascontiguousarray(featvec1[1], np.int32)]
        b = [np.ascontiguousarray(free2, np.uint8), np.ascontiguousarray(featvec2[0], np.int32),
             np.ascontiguousarray(featvec2[1], np.int32)]
        out = np.full(len(kf1), -1, np.int32)
        nm = C.c_int()
        check(lib().orbx_frame_search_for_triangulation(kf1._h, ptr(a[0]), ptr(a[1]), ptr(a[2]), kf2._h, ptr(b[0]), ptr(b[1]),
                                                        ptr(b[2]), len(a[2]) - 1, C.byref(g), int(bOnlyStereo), int(bCoarse),
                                                        int(self.mbCheckOrientation), ptr(out), C.byref(nm)))
        return out, nm.value


def triangulation_geometry(n_left1, n_left2, cam_types, cam_params, poses, epipole, scale_factors2, level_sigma2_1,
                           level_sigma2_2):
    """orbx_triangulation_geometry from flat arrays: cam_types[4] / cam_params[4][8] = {cam1 left, cam1 right, cam2 left,
    cam2 right}; poses[4][12] = (R12 row-major, t12) for ll, lr, rl, rr.  Returns (struct, arrays it points into)."""
    sf = np.ascontiguousarray(scale_factors2, np.float32)
    s1 = np.ascontiguousarray(level_sigma2_1, np.float32)
    s2 = np.ascontiguousarray(level_sigma2_2, np.float32)
    g = TriangulationGeometry()
    g.n_left1, g.n_left2 = int(n_left1), int(n_left2)
    cp = np.asarray(cam_params, np.float32).reshape(4, 8)
    ps = np.asarray(poses, np.float32).reshape(4, 12)
    for c in range(4):
        m = g.cam1[c] if c < 2 else g.cam2[c - 2]
        m.type = int(cam_types[c])
        m.params[:] = [float(v) for v in cp[c]]
        pose = g.pose[c >> 1][c & 1]
        pose.R12[:] = [float(v) for v in ps[c, :9]]
        pose.t12[:] = [float(v) for v in ps[c, 9:]]
    g.epipole_x, g.epipole_y = float(epipole[0]), float(epipole[1])
    g.scale_factors2 = sf.ctypes.data_as(C.POINTER(C.c_float))
    g.level_sigma2_1 = s1.ctypes.data_as(C.POINTER(C.c_float))
    g.level_sigma2_2 = s2.ctypes.data_as(C.POINTER(C.c_float))
    g.n_levels = len(sf)
    g.gate = None
    g.gate_user = None
    return g, (sf, s1, s2)


class CameraModel(C.Structure):
    """orbx_camera_model (include/orbx.h)."""
    _fields_ = [("type", C.c_int), ("params", C.c_float * 8)]


def camera_model(cam_type: int, params) -> "CameraModel":
    """orbx_camera_model from GeometricCamera::mvParameters: 0 = Pinhole {fx, fy, cx, cy}, 1 = KannalaBrandt8 {+ k0..k3}."""
    m = CameraModel()
    m.type = int(cam_type)
    p = [float(v) for v in params] + [0.0] * 8
    m.params[:] = p[:8]
    return m


def _f32c(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    return a if shape is None else a.reshape(shape)


def camera_project(cam: "CameraModel", xyz):
    """GeometricCamera::project(Eigen::Vector3f) for n points [n, 3] -> [n, 2] (host arithmetic)."""
    xyz = _f32c(xyz, (-1, 3))
    uv = np.zeros((len(xyz), 2), np.float32)
    check(lib().orbx_camera_project(C.addressof(cam), ptr(xyz), len(xyz), ptr(uv)))
    return uv


def camera_unproject(cam: "CameraModel", uv):
    """GeometricCamera::unprojectEig for n image points [n, 2] -> rays [n, 3]."""
    uv = _f32c(uv, (-1, 2))
    rays = np.zeros((len(uv), 3), np.float32)
    check(lib().orbx_camera_unproject(C.addressof(cam), ptr(uv), len(uv), ptr(rays)))
    return rays


def camera_triangulate_matches(cam1, cam2, uv1, uv2, R12, t12, sigma1, sigma2):
    """KannalaBrandt8::TriangulateMatches for n pairs: (depth or rejection code [n], points [n, 3])."""
    uv1, uv2 = _f32c(uv1, (-1, 2)), _f32c(uv2, (-1, 2))
    n = len(uv1)
    s1 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma1, np.float32), (n,)))
    s2 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma2, np.float32), (n,)))
    depth, p3d = np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    check(lib().orbx_camera_triangulate_matches(C.addressof(cam1), C.addressof(cam2), ptr(uv1), ptr(uv2), n,
                                                ptr(_f32c(R12, 9)), ptr(_f32c(t12, 3)), ptr(s1), ptr(s2), ptr(depth), ptr(p3d)))
    return depth, p3d


def camera_epipolar_constrain(cam1, cam2, uv1, uv2, R12, t12, sigma1, sigma2):
    """pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, sigma1, sigma2) for n pairs -> bool [n]."""
    uv1, uv2 = _f32c(uv1, (-1, 2)), _f32c(uv2, (-1, 2))
    n = len(uv1)
    s1 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma1, np.float32), (n,)))
    s2 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma2, np.float32), (n,)))
    ok = np.zeros(n, np.uint8)
    check(lib().orbx_camera_epipolar_constrain(C.addressof(cam1), C.addressof(cam2), ptr(uv1), ptr(uv2), n,
                                               ptr(_f32c(R12, 9)), ptr(_f32c(t12, 3)), ptr(s1), ptr(s2), ptr(ok)))
    return ok.astype(bool)


class RelPose(C.Structure):
    """orbx_rel_pose."""
    _fields_ = [("R12", C.c_float * 9), ("t12", C.c_float * 3)]


GATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)


class TriangulationGeometry(C.Structure):
    """orbx_triangulation_geometry."""
    _fields_ = [("n_left1", C.c_int), ("n_left2", C.c_int), ("cam1", CameraModel * 2), ("cam2", CameraModel * 2),
                ("pose", (RelPose * 2) * 2), ("epipole_x", C.c_float), ("epipole_y", C.c_float),
                ("scale_factors2", C.POINTER(C.c_float)), ("level_sigma2_1", C.POINTER(C.c_float)),
                ("level_sigma2_2", C.POINTER(C.c_float)), ("n_levels", C.c_int), ("gate", C.c_void_p),
                ("gate_user", C.c_void_p)]


class Epipolar(C.Structure):
    """orbx_epipolar (include/orbx.h)."""
    _fields_ = [("F12", C.c_float * 9), ("epipole_x", C.c_float), ("epipole_y", C.c_float),
                ("scale_factors2", C.POINTER(C.c_float)), ("level_sigma2_2", C.POINTER(C.c_float)),
                ("n_levels", C.c_int)]


def bucket_allpairs(desc1, idx1, off1, desc2, idx2, off2):
    """Distances between all entries of the same node (row_start per idx1 entry, uint16 distances)."""
    desc1, desc2 = _desc(desc1), _desc(desc2)
    idx1, idx2 = np.ascontiguousarray(idx1, np.int32), np.ascontiguousarray(idx2, np.int32)
    off1, off2 = np.ascontiguousarray(off1, np.int32), np.ascontiguousarray(off2, np.int32)
    rs = np.zeros(len(idx1) + 1, np.int32)
    cap = int(sum(int(off1[b + 1] - off1[b]) * int(off2[b + 1] - off2[b]) for b in range(len(off1) - 1)))
    dist = np.zeros(max(cap, 1), np.uint16)
    n = C.c_size_t()
    check(lib().orbx_bucket_allpairs(ptr(desc1), len(desc1), ptr(idx1), ptr(off1), ptr(desc2), len(desc2), ptr(idx2),
                                     ptr(off2), len(off1) - 1, ptr(rs), ptr(dist), cap, C.byref(n)))
    return rs, dist[:n.value].copy()


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


def knn_match2_device_static(d_query, d_train, out, base_index: int = 0):
    """orbx_hamming_knn2_device_static: the same search with its partial results in the library's
    fixed device array; `out` (int32 [nq,4] CUDA tensor) is the caller's, nothing is allocated."""
    import torch
    nq, nt = d_query.shape[0], d_train.shape[0]
    st = torch.cuda.current_stream(d_query.device).cuda_stream
    check(lib().orbx_hamming_knn2_device_static(d_query.data_ptr(), nq, d_train.data_ptr(), nt, base_index,
                                                out.data_ptr(), st))
    return out


def knn2_merge_device(d_parts, out=None):
    """d_parts: int32 CUDA tensor [n_parts, nq, 4] (e.g. all-gathered per-GPU shard results)."""
    import torch
    nparts, nq, _ = d_parts.shape
    if out is None:
        out = torch.empty((nq, 4), dtype=torch.int32, device=d_parts.device)
    st = torch.cuda.current_stream(d_parts.device).cuda_stream
    check(lib().orbx_knn2_merge_device(d_parts.data_ptr(), nparts, nq, out.data_ptr(), st))
    return out


def compute_stereo_fisheye_matches(kps_left, desc_left, mono_left, kps_right, desc_right, mono_right, cam_left: "CameraModel",
                                   cam_right: "CameraModel", R_lr, t_lr, level_sigma2):
    """Frame::ComputeStereoFishEyeMatches (Frame.cc:1286-1338): returns (mvLeftToRightMatch, mvRightToLeftMatch, mvDepth,
    mvStereo3Dpoints, nMatches).  cam_* are KannalaBrandt8 models; R_lr / t_lr = mRlr / mtlr."""
    kps_left, kps_right = np.ascontiguousarray(kps_left, KP_DTYPE), np.ascontiguousarray(kps_right, KP_DTYPE)
    desc_left, desc_right = np.ascontiguousarray(desc_left, np.uint8), np.ascontiguousarray(desc_right, np.uint8)
    R = np.ascontiguousarray(R_lr, np.float32).reshape(9)
    t = np.ascontiguousarray(t_lr, np.float32).reshape(3)
    sig = np.ascontiguousarray(level_sigma2, np.float32)
    nL, nR = len(kps_left), len(kps_right)
    l2r, r2l = np.empty(nL, np.int32), np.empty(nR, np.int32)
    depth, p3d = np.empty(nL, np.float32), np.zeros((nL, 3), np.float32)
    n = C.c_int(0)
    check(lib().orbx_stereo_fisheye_matches(ptr(kps_left), ptr(desc_left), nL, int(mono_left), ptr(kps_right), ptr(desc_right), nR,
                                            int(mono_right), C.byref(cam_left), C.byref(cam_right), ptr(R), ptr(t), ptr(sig),
                                            len(sig), ptr(l2r), ptr(r2l), ptr(depth), ptr(p3d), C.byref(n)))
    return l2r, r2l, depth, p3d, n.value


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


def distinctive_descriptors(desc, off):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:416-533) batched over map points: desc [total, 32],
    off [n_points + 1]; returns (best observation per point relative to its first row, its median distance)."""
    desc = _desc(desc) if len(desc) else np.zeros((0, 32), np.uint8)
    off = np.ascontiguousarray(off, np.int32)
    n = len(off) - 1
    bi = np.full(n, -1, np.int32)
    bm = np.zeros(n, np.int32)
    check(lib().orbx_distinctive_descriptors(ptr(desc), ptr(off), n, ptr(bi), ptr(bm)))
    return bi, bm
