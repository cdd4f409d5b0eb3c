"""ctypes binding of the CPU oracle (oracle/libsift_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg as the checker / timed baseline.  The product
package (sift_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsift_oracle.so")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
CANDIDATE_DTYPE = np.dtype([("oct", "<i4"), ("img", "<i4"), ("i", "<i4"), ("j", "<i4")])
STATS_FIELDS = ("candidates", "out_of_bounds", "no_convergence", "low_contrast", "det_nonpositive",
                "edge_like", "singular", "localized", "oriented")
STATS_DTYPE = np.dtype([(f, "<i8") for f in STATS_FIELDS])


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("sift_oracle.c", "homography_oracle.c", "stitch_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libsift_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, ci, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
        L.orc_bgr_to_gray.argtypes = [vp, ci, ci, vp]
        L.orc_create_gray.restype = vp
        L.orc_create_gray.argtypes = [vp, ci, ci, ci]
        L.orc_destroy.argtypes = [vp]
        L.orc_gen_gaussian_images.argtypes = [vp, ci]
        L.orc_gen_dog_images.argtypes = [vp]
        L.orc_gen_scale_space_extrema.argtypes = [vp, ci]
        L.orc_clean_keypoints.argtypes = [vp]
        L.orc_get_descriptors.argtypes = [vp, ci]
        L.orc_exec.argtypes = [vp, ci, ci, dp]
        L.orc_octaves.argtypes = [vp]
        L.orc_level_dims.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
        for name in ("orc_onex", "orc_base"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        for name in ("orc_gauss", "orc_dog"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp, ci, ci]
        L.orc_num_candidates.argtypes = [vp]
        L.orc_candidates.restype = vp
        L.orc_candidates.argtypes = [vp]
        L.orc_stats_table.restype = vp
        L.orc_stats_table.argtypes = [vp]
        L.orc_num_keypoints.argtypes = [vp]
        L.orc_keypoints.restype = vp
        L.orc_keypoints.argtypes = [vp]
        L.orc_descriptors.restype = vp
        L.orc_descriptors.argtypes = [vp]
        L.orc_get.argtypes = [vp, vp]
        L.orc_gaussian_kernel.argtypes = [C.c_double, dp]
        L.orc_knn2_f32.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, vp, vp, ci]
        L.orc_knn2_u8.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, vp, vp, ci]
        L.orc_min_dist_filter.argtypes = [vp, ci, vp]
        _lib = L
    return _lib


def _from_ptr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


def bgr_to_gray(bgr):
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr_to_gray(bgr.ctypes.data, w, h, out.ctypes.data)
    return out


def gaussian_kernel(sigma):
    buf = (C.c_double * 64)()
    n = lib().orc_gaussian_kernel(float(sigma), buf)
    return np.array(buf[:n])


class OracleSift:
    """sift::sift_handler restated on the CPU (sift.hpp:20-118)."""

    def __init__(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        assert gray.ndim == 2
        self.h, self.w = gray.shape
        self._L = lib()
        self._h = self._L.orc_create_gray(gray.ctypes.data, self.w, self.h, self.w)
        self.timings = None

    def close(self):
        if self._h:
            self._L.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # the five TIMEIT stages of exec()
    def gen_gaussian_images(self, blur_threads=1):
        self._L.orc_gen_gaussian_images(self._h, blur_threads)

    def gen_dog_images(self):
        self._L.orc_gen_dog_images(self._h)

    def gen_scale_space_extrema(self, threads=1):
        self._L.orc_gen_scale_space_extrema(self._h, threads)

    def clean_keypoints(self):
        self._L.orc_clean_keypoints(self._h)

    def get_descriptors(self, threads=1):
        self._L.orc_get_descriptors(self._h, threads)

    def exec(self, threads=1, blur_threads=1):
        t = (C.c_double * 5)()
        self._L.orc_exec(self._h, threads, blur_threads, t)
        self.timings = dict(zip(
            ("gen_gaussian_images", "gen_dog_images", "gen_scale_space_extrema", "clean_keypoints",
             "get_descriptors"), list(t)))
        return self

    @property
    def octaves(self):
        return self._L.orc_octaves(self._h)

    def level_dims(self, oct):
        w, h = C.c_int(), C.c_int()
        self._L.orc_level_dims(self._h, oct, C.byref(w), C.byref(h))
        return w.value, h.value

    def onex(self):
        return _from_ptr(self._L.orc_onex(self._h), (self.h, self.w), np.float64)

    def base(self):
        return _from_ptr(self._L.orc_base(self._h), (2 * self.h, 2 * self.w), np.float64)

    def gauss(self, oct, lvl):
        w, h = self.level_dims(oct)
        return _from_ptr(self._L.orc_gauss(self._h, oct, lvl), (h, w), np.float64)

    def dog(self, oct, lvl):
        w, h = self.level_dims(oct)
        return _from_ptr(self._L.orc_dog(self._h, oct, lvl), (h, w), np.float64)

    def candidates(self):
        n = self._L.orc_num_candidates(self._h)
        return _from_ptr(self._L.orc_candidates(self._h), (n,), CANDIDATE_DTYPE)

    def stats(self):
        return _from_ptr(self._L.orc_stats_table(self._h), (self.octaves, 3), STATS_DTYPE)

    def keypoints(self):
        n = self._L.orc_num_keypoints(self._h)
        return _from_ptr(self._L.orc_keypoints(self._h), (n,), KEYPOINT_DTYPE)

    def descriptors(self):
        n = self._L.orc_num_keypoints(self._h)
        return _from_ptr(self._L.orc_descriptors(self._h), (n, 128), np.float64)

    def get(self):
        n = self._L.orc_num_keypoints(self._h)
        out = np.empty((n, 128), np.float32)
        if n:
            self._L.orc_get(self._h, out.ctypes.data)
        return out


def knn2_f32(q, t, threads=1):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    nq, dim = q.shape
    nt = t.shape[0]
    i1 = np.empty(nq, np.int32); i2 = np.empty(nq, np.int32)
    d1 = np.empty(nq, np.float32); d2 = np.empty(nq, np.float32)
    acc = np.empty(nq, np.uint8)
    lib().orc_knn2_f32(q.ctypes.data, nq, t.ctypes.data, nt, dim, i1.ctypes.data, d1.ctypes.data,
                       i2.ctypes.data, d2.ctypes.data, acc.ctypes.data, threads)
    return i1, d1, i2, d2, acc


def knn2_u8(q, t, threads=1):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nq, dim = q.shape
    nt = t.shape[0]
    i1 = np.empty(nq, np.int32); i2 = np.empty(nq, np.int32)
    d1 = np.empty(nq, np.int32); d2 = np.empty(nq, np.int32)
    acc = np.empty(nq, np.uint8)
    lib().orc_knn2_u8(q.ctypes.data, nq, t.ctypes.data, nt, dim, i1.ctypes.data, d1.ctypes.data,
                      i2.ctypes.data, d2.ctypes.data, acc.ctypes.data, threads)
    return i1, d1, i2, d2, acc


def min_dist_filter(dist):
    dist = np.ascontiguousarray(dist, np.float32)
    good = np.empty(len(dist), np.int32)
    n = lib().orc_min_dist_filter(dist.ctypes.data, len(dist), good.ctypes.data)
    return good[:n].copy()


# ---- findHomography(RANSAC) restatement (homography_oracle.c; parity unpinned) ----

class HomographyInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("found", "iterations", "best_iteration", "inliers", "lm_iterations")]


def _pts(a):
    return np.ascontiguousarray(a, np.float32).reshape(-1, 2)


def find_homography(src, dst, thresh=3.0, max_iters=2000, confidence=0.995):
    """Returns (H 3x3 or None, mask uint8[N], info dict)."""
    src, dst = _pts(src), _pts(dst)
    n = src.shape[0]
    H = np.zeros(9)
    mask = np.zeros(max(n, 1), np.uint8)
    info = HomographyInfo()
    L = lib()
    L.orc_find_homography(C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data), n, C.c_double(thresh), max_iters,
                          C.c_double(confidence), C.c_void_p(H.ctypes.data), C.c_void_p(mask.ctypes.data), C.byref(info))
    d = {k: getattr(info, k) for k, _ in HomographyInfo._fields_}
    return (H.reshape(3, 3) if info.found else None), mask[:n], d


def homography_rng_sequence(state, n):
    out = np.zeros(n, np.uint32)
    lib().orc_h_rng_sequence(C.c_uint64(state), n, C.c_void_p(out.ctypes.data))
    return out


def homography_subsets(src, dst, n_iters):
    src, dst = _pts(src), _pts(dst)
    out = np.zeros((n_iters, 4), np.int32)
    k = lib().orc_h_subsets(C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data), src.shape[0], n_iters,
                            C.c_void_p(out.ctypes.data))
    return out[:k]


def homography_kernel(src, dst):
    src, dst = _pts(src), _pts(dst)
    H = np.zeros(9)
    ok = lib().orc_h_kernel(C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data), src.shape[0], C.c_void_p(H.ctypes.data))
    return H.reshape(3, 3) if ok else None


def homography_inliers(src, dst, H, thresh=3.0):
    src, dst = _pts(src), _pts(dst)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    mask = np.zeros(max(src.shape[0], 1), np.uint8)
    n = lib().orc_h_inliers(C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data), src.shape[0], C.c_void_p(H.ctypes.data),
                            C.c_double(thresh), C.c_void_p(mask.ctypes.data))
    return n, mask[:src.shape[0]]


# ---- panorama composition restatement (stitch_oracle.c; parity unpinned) ----

def warp_perspective(src_bgr, H, dw, dh):
    src = np.ascontiguousarray(src_bgr, np.uint8)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    dst = np.zeros((dh, dw, 3), np.uint8)
    ok = lib().orc_warp_perspective(C.c_void_p(src.ctypes.data), src.shape[1], src.shape[0], C.c_void_p(H.ctypes.data), dw, dh,
                                    C.c_void_p(dst.ctypes.data))
    return dst if ok else None


def stitch_pair(ref_bgr, img_bgr, H):
    """main.cpp:115-145: returns the cropped panorama (rows of ref, up to ref+img columns) or None."""
    ref = np.ascontiguousarray(ref_bgr, np.uint8)
    img = np.ascontiguousarray(img_bgr, np.uint8)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    out = np.zeros(ref.shape[0] * (ref.shape[1] + img.shape[1]) * 3, np.uint8)
    w = C.c_int()
    ok = lib().orc_stitch_pair(C.c_void_p(ref.ctypes.data), ref.shape[1], ref.shape[0], C.c_void_p(img.ctypes.data), img.shape[1],
                               img.shape[0], C.c_void_p(H.ctypes.data), C.c_void_p(out.ctypes.data), C.byref(w))
    if not ok:
        return None
    return out[: ref.shape[0] * w.value * 3].reshape(ref.shape[0], w.value, 3).copy()
