"""Input staging in front of the extractor over the C ABI: the cv::remap rectification or cv::resize that
System::TrackStereo / TrackMonocular apply (System.cc:355-376) and the grey conversion of
Tracking::GrabImage* (Tracking.cc:1591-1619), as one kernel."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr

STAGE_NONE, STAGE_REMAP, STAGE_RESIZE = 0, 1, 2


class StageParams(C.Structure):
    _fields_ = [("src_width", C.c_int), ("src_height", C.c_int), ("channels", C.c_int), ("rgb_order", C.c_int),
                ("mode", C.c_int), ("dst_width", C.c_int), ("dst_height", C.c_int), ("map_x", C.c_void_p),
                ("map_y", C.c_void_p), ("device", C.c_int)]


class InputStager:
    """InputStager(width, height, channels, rgb=False, maps=(M1, M2) | size=(w, h)):
    maps: float32 [H', W'] rectification maps (cv::initUndistortRectifyMap, CV_32FC1) -> cv::remap;
    size: (width, height) -> cv::resize; neither: grey conversion only.  rgb = Tracking::mbRGB."""

    def __init__(self, width: int, height: int, channels: int = 1, rgb: bool = False, maps=None, size=None,
                 device: int = 0):
        assert maps is None or size is None
        p = StageParams()
        p.src_width, p.src_height, p.channels, p.rgb_order, p.device = width, height, channels, int(bool(rgb)), device
        self._keep = None
        if maps is not None:
            mx = np.ascontiguousarray(maps[0], np.float32)
            my = np.ascontiguousarray(maps[1], np.float32)
            assert mx.ndim == 2 and mx.shape == my.shape
            p.mode, p.dst_height, p.dst_width = STAGE_REMAP, mx.shape[0], mx.shape[1]
            p.map_x, p.map_y = mx.ctypes.data, my.ctypes.data
            self._keep = (mx, my)
        elif size is not None:
            p.mode, p.dst_width, p.dst_height = STAGE_RESIZE, int(size[0]), int(size[1])
        else:
            p.mode = STAGE_NONE
        h = C.c_void_p()
        check(lib().orbx_stager_create(C.byref(p), C.byref(h)))
        self._h = h
        self._keep = None
        self.src_width, self.src_height, self.channels = width, height, channels
        w, hh = C.c_int(), C.c_int()
        check(lib().orbx_stager_output_size(h, C.byref(w), C.byref(hh)))
        self.width, self.height = w.value, hh.value

    @property
    def handle(self):
        return self._h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().orbx_stager_destroy(h)
            except Exception:
                pass

    def __call__(self, image: np.ndarray) -> np.ndarray:
        """One host image (H x W [x C] uint8) -> the grey image the extractor receives."""
        img = np.ascontiguousarray(image, np.uint8)
        cn = 1 if img.ndim == 2 else img.shape[2]
        assert img.shape[:2] == (self.src_height, self.src_width) and cn == self.channels
        out = np.zeros((self.height, self.width), np.uint8)
        check(lib().orbx_stage(self._h, ptr(img), self.src_width * cn, ptr(out), self.width))
        return out

    def stage_batch_device(self, d_src, out=None, stream=None):
        """torch uint8 CUDA tensor [n, H, W] or [n, H, W, C] -> [n, H', W'] grey, on the current torch stream."""
        import torch
        assert d_src.is_cuda and d_src.dtype == torch.uint8 and d_src.dim() in (3, 4)
        cn = 1 if d_src.dim() == 3 else d_src.shape[3]
        assert cn == self.channels and tuple(d_src.shape[1:3]) == (self.src_height, self.src_width)
        assert d_src.stride(2) == cn
        nf = d_src.shape[0]
        if out is None:
            out = torch.empty((nf, self.height, self.width), dtype=torch.uint8, device=d_src.device)
        st = torch.cuda.current_stream(d_src.device).cuda_stream if stream is None else stream
        check(lib().orbx_stage_batch_device(self._h, d_src.data_ptr(), nf, d_src.stride(1), d_src.stride(0),
                                            out.data_ptr(), out.stride(1), out.stride(0), st))
        return out
