"""Python mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:42-121) over the C ABI.

Host arrays go through orbx_extract / orbx_extract_batch; torch CUDA tensors go through
orbx_extract_batch_device (device resident, asynchronous on the current torch stream).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, RECT_DTYPE, OrbxError, OrbxParams, check, lib, ptr


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  (ORBextractor.cc:477).

    max_width / max_height / max_batch size the device buffers of the handle.
    """

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 max_width: int = 752, max_height: int = 480, max_batch: int = 1, device: int = -1):
        self._L = lib()
        self._h = C.c_void_p()
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.scaleFactor = float(np.float32(scaleFactor))
        self.max_batch = int(max_batch)
        p = OrbxParams(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST),
                       int(max_width), int(max_height), int(max_batch), int(device))
        check(self._L.orbx_extractor_create(C.byref(p), C.byref(self._h)))
        n = self.nlevels
        self.mvScaleFactor = np.zeros(n, np.float32)
        self.mvInvScaleFactor = np.zeros(n, np.float32)
        self.mvLevelSigma2 = np.zeros(n, np.float32)
        self.mvInvLevelSigma2 = np.zeros(n, np.float32)
        self.mnFeaturesPerLevel = np.zeros(n, np.int32)
        check(self._L.orbx_extractor_tables(self._h, ptr(self.mvScaleFactor), ptr(self.mvInvScaleFactor),
                                            ptr(self.mvLevelSigma2), ptr(self.mvInvLevelSigma2),
                                            ptr(self.mnFeaturesPerLevel)))
        self.max_keypoints = self._L.orbx_extractor_max_keypoints(self._h)
        self.mvDynamicArea = []   # list of (x, y, w, h); ORBextractor.h:95-96

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # getters of the reference class
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return self.scaleFactor
    def GetScaleFactors(self): return self.mvScaleFactor.copy()
    def GetInverseScaleFactors(self): return self.mvInvScaleFactor.copy()
    def GetScaleSigmaSquares(self): return self.mvLevelSigma2.copy()
    def GetInverseScaleSigmaSquares(self): return self.mvInvLevelSigma2.copy()

    def __call__(self, image: np.ndarray, vLappingArea=(0, 0)):
        """operator()(image, mask, keypoints, descriptors, vLappingArea) (ORBextractor.cc:1271).
        Returns (keypoints[KP_DTYPE], descriptors[n, 32] uint8, monoIndex); monoIndex is -1 for
        an empty image."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8), -1
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 image expected"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        n, mono = C.c_int(), C.c_int()
        rects = None
        if self.mvDynamicArea:
            rects = np.array([tuple(r) for r in self.mvDynamicArea], RECT_DTYPE)
        while True:
            cap = self.max_keypoints
            kps = np.zeros(cap, KP_DTYPE)
            desc = np.zeros((cap, 32), np.uint8)
            try:
                check(self._L.orbx_extract(self._h, image.ctypes.data, w, h, image.strides[0], int(vLappingArea[0]),
                                           int(vLappingArea[1]), ptr(rects), 0 if rects is None else len(rects), ptr(kps),
                                           ptr(desc), cap, C.byref(n), C.byref(mono)))
                break
            except OrbxError:
                if not self._capacity_grew():
                    raise
        return kps[:n.value].copy(), desc[:n.value].copy(), mono.value

    def _capacity_grew(self):
        """True if the geometry of the last image holds more keypoints than the arrays offered so far (an image of
        another aspect ratio than the handle was created for): the caller offers larger arrays and calls again."""
        now = self._L.orbx_extractor_max_keypoints(self._h)
        if now <= self.max_keypoints:
            return False
        self.max_keypoints = now
        return True

    def extract_batch(self, images, vLappingArea=(0, 0), copy=True):
        """One pass of every kernel over a list of equally sized host images (orbx_extract_batch: a batch of three
        upload chunks or more is pipelined -- staging, upload, kernels and download overlap).
        Returns a list of (keypoints, descriptors, monoIndex), views into two arrays that hold the whole batch;
        copy=False returns views into the arrays that the extractor keeps and overwrites in its next batch call."""
        nf = len(images)
        assert 1 <= nf <= self.max_batch
        h, w = images[0].shape
        # images that share one row stride (views into wider buffers, as cv::Mat ROIs are) go in as they are
        stride = images[0].strides[0]
        if all(im.strides == (stride, 1) for im in images) and stride >= w:
            imgs = list(images)
        else:
            imgs = [im if im.flags.c_contiguous else np.ascontiguousarray(im) for im in images]
            stride = w
        assert all(im.shape == (h, w) and im.dtype == np.uint8 for im in imgs)
        ptrs = (C.c_void_p * nf)(*[im.__array_interface__["data"][0] for im in imgs])
        n = np.zeros(nf, np.int32)
        mono = np.zeros(nf, np.int32)
        while True:
            cap = self.max_keypoints
            # The library writes into two arrays that belong to the extractor and are reused by the next call of the
            # same shape, as a C++ caller reuses its vectors (fresh arrays cost a page fault per 4 KB the library
            # writes); copy=True then copies each frame's used rows out of them.
            held = getattr(self, "_batch_out", None)
            if held is not None and held[0].shape == (nf, cap):
                kps, desc = held
            else:
                kps = np.empty((nf, cap), KP_DTYPE)
                desc = np.empty((nf, cap, 32), np.uint8)
                self._batch_out = (kps, desc)
            try:
                check(self._L.orbx_extract_batch(self._h, ptrs, nf, w, h, stride, int(vLappingArea[0]), int(vLappingArea[1]),
                                                 ptr(kps), ptr(desc), cap, ptr(n), ptr(mono)))
                break
            except OrbxError:
                if not self._capacity_grew():
                    raise
        if copy:
            # one copy of the used part of both arrays (the per-frame results are views into it): two allocations per
            # call instead of two per frame
            nmax = int(n.max()) if nf else 0
            kps, desc = kps[:, :nmax].copy(), desc[:, :nmax].copy()
        return [(kps[f, :n[f]], desc[f, :n[f]], int(mono[f])) for f in range(nf)]

    def extract_batch_arrays(self, stack, vLappingArea=(0, 0)):
        """orbx_extract_batch on a stack of host images [n, H, W] (uint8, unit pixel stride; any row and frame
        stride), results as whole arrays: (keypoints [n, cap] KP_DTYPE, descriptors [n, cap, 32], n_keypoints [n],
        monoIndex [n]); frame f's rows are [f, :n_keypoints[f]].  The arrays belong to the extractor and are
        overwritten by its next call of the same shape.  No per-frame Python objects are made: this is the mirror's
        counterpart of a C++ caller that passes plain buffers."""
        assert stack.ndim == 3 and stack.dtype == np.uint8 and stack.strides[2] == 1
        nf, h, w = stack.shape
        assert 1 <= nf <= self.max_batch and stack.strides[1] >= w
        base = stack.__array_interface__["data"][0]
        ptrs = (base + np.arange(nf, dtype=np.uint64) * np.uint64(stack.strides[0])).astype(np.uint64)
        held = getattr(self, "_arrays_out", None)
        while True:
            cap = self.max_keypoints
            if held is None or held[0].shape != (nf, cap):
                held = (np.empty((nf, cap), KP_DTYPE), np.empty((nf, cap, 32), np.uint8), np.zeros(nf, np.int32),
                        np.zeros(nf, np.int32))
                self._arrays_out = held
            kps, desc, n, mono = held
            try:
                check(self._L.orbx_extract_batch(self._h, ptr(ptrs), nf, w, h, stack.strides[1], int(vLappingArea[0]),
                                                 int(vLappingArea[1]), ptr(kps), ptr(desc), cap, ptr(n), ptr(mono)))
                return kps, desc, n, mono
            except OrbxError:
                if not self._capacity_grew():
                    raise
                held = None

    def extract_batch_device(self, d_images, vLappingArea=(0, 0), out=None, stream=None):
        """d_images: torch uint8 CUDA tensor [n, H, W] (row stride W, or any row/frame stride).
        Enqueues on the current torch stream and returns (kps, desc, counts) CUDA tensors:
        kps uint8 [n, cap, 28] (view as KP_DTYPE on the host), desc uint8 [n, cap, 32],
        counts int32 [n, 2] = (n_keypoints, monoIndex).  Call status() after synchronising."""
        import torch
        assert d_images.is_cuda and d_images.dtype == torch.uint8 and d_images.dim() == 3
        assert d_images.stride(2) == 1
        nf, h, w = d_images.shape
        cap = self.max_keypoints
        if out is None:
            kps = torch.empty((nf, cap, 28), dtype=torch.uint8, device=d_images.device)
            desc = torch.empty((nf, cap, 32), dtype=torch.uint8, device=d_images.device)
            counts = torch.empty((nf, 2), dtype=torch.int32, device=d_images.device)
        else:
            kps, desc, counts = out
        st = torch.cuda.current_stream(d_images.device).cuda_stream if stream is None else stream
        rects = None
        if self.mvDynamicArea:
            rects = np.array([tuple(r) for r in self.mvDynamicArea], RECT_DTYPE)
        check(self._L.orbx_extract_batch_device(self._h, d_images.data_ptr(), nf, w, h, d_images.stride(1),
                                                d_images.stride(0), int(vLappingArea[0]), int(vLappingArea[1]),
                                                ptr(rects), 0 if rects is None else len(rects), kps.data_ptr(),
                                                desc.data_ptr(), cap, counts.data_ptr(), st))
        return kps, desc, counts

    def extract_batch_staged_device(self, stager, d_src, vLappingArea=(0, 0), out=None, stream=None):
        """extract_batch_device on images as the camera delivers them: d_src is a torch uint8 CUDA tensor
        [n, H, W] or [n, H, W, C]; `stager` (InputStager) rectifies / resizes / converts to grey straight
        into level 0 of the pyramid.  The extractor must have been created for the stager's output size."""
        import torch
        assert d_src.is_cuda and d_src.dtype == torch.uint8 and d_src.dim() in (3, 4)
        nf = d_src.shape[0]
        cn = 1 if d_src.dim() == 3 else d_src.shape[3]
        assert cn == stager.channels and tuple(d_src.shape[1:3]) == (stager.src_height, stager.src_width)
        assert d_src.stride(2) == cn and (d_src.dim() == 3 or d_src.stride(3) == 1)
        cap = self.max_keypoints
        if out is None:
            kps = torch.empty((nf, cap, 28), dtype=torch.uint8, device=d_src.device)
            desc = torch.empty((nf, cap, 32), dtype=torch.uint8, device=d_src.device)
            counts = torch.empty((nf, 2), dtype=torch.int32, device=d_src.device)
        else:
            kps, desc, counts = out
        st = torch.cuda.current_stream(d_src.device).cuda_stream if stream is None else stream
        rects = None
        if self.mvDynamicArea:
            rects = np.array([tuple(r) for r in self.mvDynamicArea], RECT_DTYPE)
        check(self._L.orbx_extract_batch_staged_device(self._h, stager.handle, d_src.data_ptr(), nf, d_src.stride(1),
                                                       d_src.stride(0), int(vLappingArea[0]), int(vLappingArea[1]),
                                                       ptr(rects), 0 if rects is None else len(rects), kps.data_ptr(),
                                                       desc.data_ptr(), cap, counts.data_ptr(), st))
        return kps, desc, counts

    def status(self):
        """Raises if the last asynchronous batch overflowed an internal buffer."""
        check(self._L.orbx_extract_status(self._h))

    STAGES = ("copy_in", "pyramid", "fast", "quadtree", "layout", "describe")

    def set_timing(self, enable: bool):
        check(self._L.orbx_extractor_set_timing(self._h, 1 if enable else 0))

    def get_timing(self):
        """Per-stage milliseconds summed over the batches since set_timing(True), and the
        number of batches: ({stage: ms}, n_calls)."""
        ms = np.zeros(8, np.float64)
        n = C.c_long()
        check(self._L.orbx_extractor_get_timing(self._h, ptr(ms), 8, C.byref(n)))
        return dict(zip(self.STAGES, ms[:6].tolist())), n.value

    def pyramid_level(self, level: int, frame: int = 0) -> np.ndarray:
        """mvImagePyramid[level] of the last call (borderless copy)."""
        w, h = C.c_int(), C.c_int()
        check(self._L.orbx_pyramid_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(self._L.orbx_pyramid_level(self._h, frame, level, ptr(out), out.size, C.byref(w), C.byref(h)))
        return out

    def debug_blurred_level(self, level: int, frame: int = 0) -> np.ndarray:
        """GaussianBlur(mvImagePyramid[level]) of the last call, for an extractor that blurs whole levels (raises
        otherwise)."""
        w, h = C.c_int(), C.c_int()
        check(self._L.orbx_debug_blurred_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(self._L.orbx_debug_blurred_level(self._h, frame, level, ptr(out), out.size, C.byref(w), C.byref(h)))
        return out

    def debug_level_candidates(self, level: int, frame: int = 0):
        cap = 1 << 22
        xyr = np.zeros((cap, 3), np.float32)
        order = np.zeros(cap, np.uint32)
        n = C.c_int()
        check(self._L.orbx_debug_level_candidates(self._h, frame, level, ptr(xyr), ptr(order), cap, C.byref(n)))
        xyr, order = xyr[:n.value], order[:n.value]
        o = np.argsort(order, kind="stable")
        return xyr[o].copy(), order[o].copy()

    def debug_level_keypoints(self, level: int, frame: int = 0):
        cap = self.max_keypoints
        xyr = np.zeros((cap, 3), np.float32)
        n = C.c_int()
        check(self._L.orbx_debug_level_keypoints(self._h, frame, level, ptr(xyr), cap, C.byref(n)))
        return xyr[:n.value].copy()


def keypoints_from_device(kps_u8, counts, frame: int) -> np.ndarray:
    """Host KP_DTYPE view of frame `frame` of the tensors returned by extract_batch_device."""
    n = int(counts[frame, 0])
    raw = kps_u8[frame, :n].cpu().numpy()
    return np.frombuffer(raw.tobytes(), dtype=KP_DTYPE)
