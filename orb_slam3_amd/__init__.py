"""orb_slam3_amd: MI355X (gfx950) ORB front-end -- ORB extraction and Hamming descriptor
matching as HIP kernels behind a C ABI (include/orbx.h), with Python mirrors of the
reference's ORBextractor / ORBmatcher interfaces.  No CPU fallback: importing the
kernels' entry points fails if lib/liborbx.so is missing."""
from ._lib import KP_DTYPE, MATCH2_DTYPE, QUERY_DTYPE, RECT_DTYPE, OrbxError, lib  # noqa: F401
from .extractor import ORBextractor, keypoints_from_device  # noqa: F401
from .matcher import (Camera, FrameFeatures, ORBmatcher, bucket_allpairs, bucket_best, compute_stereo_matches, compute_stereo_matches_device,  # noqa: F401
                      distinctive_descriptors,
                      fisheye_ratio_matches, knn2_merge_device, knn_match2, knn_match2_device, window_candidates)
from .vocabulary import Vocabulary, shared_nodes  # noqa: F401,E402
from .staging import InputStager  # noqa: F401,E402


def settle_interpreter() -> None:
    """For latency-sensitive callers of the Python mirror: run the interpreter's full cyclic garbage collection now and
    exempt everything alive from later ones (gc.collect() + gc.freeze()).  Importing this package, numpy and torch leaves
    about 170 000 collector-tracked objects behind; CPython's first generation-2 collection over them takes 35-45 ms and
    otherwise lands inside some later call (measured: always around the 120th-135th orbx_extract of a process).  The
    library itself is not involved -- a C or C++ caller never sees that pause."""
    import gc
    gc.collect()
    gc.freeze()
