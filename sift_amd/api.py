"""ctypes binding of the C ABI in include/sift_amd.h (sift_amd/libsift_amd.so).

This module is plumbing: every compute call goes straight to the HIP library.
There is no CPU fallback; if the library is missing or no GPU is present the
calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsift_amd.so")

SIFT_OK = 0
SIFT_ERR_INVALID = -1
SIFT_ERR_HIP = -2
SIFT_ERR_OVERFLOW = -3
SIFT_ERR_NOT_INTEGRAL = -4

LEVEL_GAUSS = 0
LEVEL_DOG = 1
LEVEL_GRAD_MAG = 2
LEVEL_GRAD_DIR = 3

STAGE_NAMES = ("base", "blur_l1", "blur_l2", "blur_l3", "blur_l4", "blur_l5", "extrema", "localize_orient", "clean",
               "descriptor", "gradient", "blur_l12")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
KNN2_DTYPE = np.dtype([("d1", "<i4"), ("i1", "<i4"), ("d2", "<i4"), ("i2", "<i4")])
COUNTS_DTYPE = np.dtype([("candidates", "<i4"), ("oriented", "<i4"), ("keypoints", "<i4"), ("overflow", "<i4")])

# every symbol include/sift_amd.h declares
EXPORTED_SYMBOLS = (
    "sift_last_error", "sift_device_count", "sift_ctx_create", "sift_ctx_destroy", "sift_ctx_stream", "sift_ctx_pipeline_runs",
    "sift_ctx_synchronize", "sift_ctx_set_single_image_capacity", "sift_num_octaves", "sift_octave_dims", "sift_extract_gray8", "sift_extract_bgr8", "sift_extract_gray8_stages", "sift_base_image_gray8",
    "sift_result_free", "sift_result_get_f32", "sift_batch_create", "sift_batch_destroy",
    "sift_batch_frames_device_ptr", "sift_batch_set_external_frames", "sift_batch_upload_gray8", "sift_batch_upload_bgr8", "sift_batch_run",
    "sift_batch_counts", "sift_batch_download", "sift_batch_result_device_ptrs", "sift_batch_get_level",
    "sift_batch_get_candidates", "sift_batch_get_oriented", "sift_batch_run_profiled", "sift_batch_descriptor_work", "sift_knn2_u8",
    "sift_knn2_f32", "sift_knn2_device", "sift_knn2_device_noalloc", "sift_knn2_merge_device", "sift_match_min_dist_filter",
    "sift_find_homography", "sift_warp_perspective_bgr8", "sift_stitch_pair_bgr8",
    "sift_synth_dead_leaves", "sift_synth_descriptors", "sift_synth_queries",
    "sift_group_create", "sift_group_destroy", "sift_group_size", "sift_shard_bounds", "sift_knn2_exchange_plan",
    "sift_extract_batch_u8", "sift_frame_results_free", "sift_knn2_sharded",
)


class SiftError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("sift_amd error %d: %s" % (code, message))
        self.code = code


class _FrameCounts(C.Structure):
    _fields_ = [("candidates", C.c_int32), ("oriented", C.c_int32), ("keypoints", C.c_int32),
                ("overflow", C.c_int32)]


class _Result(C.Structure):
    _fields_ = [("n", C.c_int32), ("octaves", C.c_int32), ("keypoints", C.c_void_p),
                ("descriptors", C.c_void_p), ("counts", _FrameCounts)]


class _HomographyInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("found", "iterations", "best_iteration", "inliers", "lm_iterations", "hypotheses_scored")]


class _FrameResult(C.Structure):
    _fields_ = [("n", C.c_int32), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("counts", _FrameCounts)]


class _StageTimes(C.Structure):
    _fields_ = [("ms", C.c_double * len(STAGE_NAMES)), ("launches", C.c_int32 * len(STAGE_NAMES))]


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own copy of the HIP runtime under the same
    soname (libamdhip64.so.7) as the system one that libsift_amd.so links.  A
    process that initialises both copies loses the GPU in the second one, so
    when torch is installed its copy is loaded first and the dynamic linker
    then binds libsift_amd.so to it: one runtime per process whatever the
    import order.  Without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """Load libsift_amd.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SiftError(SIFT_ERR_INVALID, "%s is missing: build it with `make -C sift_amd/csrc`" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, ci, sz, u64, i64 = C.c_void_p, C.c_int, C.c_size_t, C.c_uint64, C.c_int64
    pvp = C.POINTER(C.c_void_p)
    L.sift_last_error.restype = C.c_char_p
    L.sift_last_error.argtypes = []
    L.sift_device_count.argtypes = [C.POINTER(ci)]
    L.sift_ctx_create.argtypes = [ci, pvp]
    L.sift_ctx_destroy.argtypes = [vp]
    L.sift_ctx_stream.argtypes = [vp, pvp]
    L.sift_ctx_synchronize.argtypes = [vp]
    L.sift_ctx_pipeline_runs.argtypes = [vp, C.POINTER(C.c_int64)]
    L.sift_ctx_set_single_image_capacity.argtypes = [vp, ci]
    L.sift_num_octaves.argtypes = [ci, ci]
    L.sift_octave_dims.argtypes = [ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.sift_extract_gray8.argtypes = [vp, vp, ci, ci, sz, C.POINTER(_Result)]
    L.sift_extract_bgr8.argtypes = [vp, vp, ci, ci, sz, C.POINTER(_Result)]
    L.sift_extract_gray8_stages.argtypes = [vp, vp, ci, ci, sz, C.POINTER(_Result), C.POINTER(C.c_double)]
    L.sift_base_image_gray8.argtypes = [vp, vp, ci, ci, sz, vp]
    L.sift_result_free.argtypes = [C.POINTER(_Result)]
    L.sift_result_get_f32.argtypes = [C.POINTER(_Result), vp]
    L.sift_batch_create.argtypes = [vp, ci, ci, ci, ci, pvp]
    L.sift_batch_destroy.argtypes = [vp]
    L.sift_batch_frames_device_ptr.argtypes = [vp, pvp]
    L.sift_batch_set_external_frames.argtypes = [vp, vp]
    L.sift_batch_upload_gray8.argtypes = [vp, ci, vp, sz]
    L.sift_batch_upload_bgr8.argtypes = [vp, ci, vp, sz]
    L.sift_batch_run.argtypes = [vp, ci]
    L.sift_batch_counts.argtypes = [vp, ci, vp]
    L.sift_batch_download.argtypes = [vp, ci, ci, vp, vp, vp, C.POINTER(C.c_int32)]
    L.sift_batch_result_device_ptrs.argtypes = [vp, ci, pvp, pvp]
    L.sift_batch_get_level.argtypes = [vp, ci, ci, ci, ci, vp]
    L.sift_batch_get_candidates.argtypes = [vp, ci, ci, vp, C.POINTER(C.c_int32)]
    L.sift_batch_get_oriented.argtypes = [vp, ci, ci, vp, C.POINTER(C.c_int32)]
    L.sift_batch_run_profiled.argtypes = [vp, ci, C.POINTER(_StageTimes)]
    L.sift_batch_descriptor_work.argtypes = [vp, ci, vp]
    L.sift_knn2_u8.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    L.sift_knn2_f32.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    L.sift_knn2_device.argtypes = [vp, vp, ci, vp, ci, ci, vp]
    L.sift_knn2_device_noalloc.argtypes = [vp, vp, ci, vp, ci, ci, vp]
    L.sift_knn2_merge_device.argtypes = [vp, vp, ci, ci, vp, vp]
    L.sift_match_min_dist_filter.argtypes = [vp, ci, vp, C.POINTER(C.c_int32)]
    L.sift_warp_perspective_bgr8.argtypes = [vp, vp, ci, ci, sz, vp, ci, ci, vp]
    L.sift_stitch_pair_bgr8.argtypes = [vp, vp, ci, ci, sz, vp, ci, ci, sz, vp, vp, C.POINTER(ci)]
    L.sift_find_homography.argtypes = [vp, vp, vp, ci, C.c_double, ci, C.c_double, vp, vp, C.POINTER(_HomographyInfo)]
    L.sift_group_create.argtypes = [pvp, ci, pvp]
    L.sift_group_destroy.argtypes = [vp]
    L.sift_group_size.argtypes = [vp]
    L.sift_shard_bounds.argtypes = [i64, ci, vp]
    L.sift_knn2_exchange_plan.argtypes = [ci, ci, ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sift_extract_batch_u8.argtypes = [vp, vp, ci, ci, ci, sz, sz, ci, C.POINTER(_FrameResult)]
    L.sift_frame_results_free.argtypes = [C.POINTER(_FrameResult), ci]
    L.sift_knn2_sharded.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    L.sift_synth_dead_leaves.argtypes = [u64, u64, ci, ci, vp]
    L.sift_synth_descriptors.argtypes = [u64, i64, vp]
    L.sift_synth_queries.argtypes = [u64, vp, i64, i64, vp]
    _lib = L
    return L


def _check(rc):
    if rc != SIFT_OK:
        raise SiftError(rc, lib().sift_last_error().decode("utf-8", "replace"))


# ---- host-only helpers ------------------------------------------------------

def device_count():
    n = C.c_int(0)
    _check(lib().sift_device_count(C.byref(n)))
    return n.value


def num_octaves(w, h):
    return lib().sift_num_octaves(w, h)


def octave_dims(w, h, oct):
    ow, oh = C.c_int(), C.c_int()
    _check(lib().sift_octave_dims(w, h, oct, C.byref(ow), C.byref(oh)))
    return ow.value, oh.value


def synth_dead_leaves(seed, frame_index, w, h):
    out = np.empty((h, w), np.uint8)
    _check(lib().sift_synth_dead_leaves(seed, frame_index, w, h, out.ctypes.data))
    return out


def synth_descriptors(seed, n):
    out = np.empty((n, 128), np.uint8)
    _check(lib().sift_synth_descriptors(seed, n, out.ctypes.data))
    return out


def synth_queries(seed, train, nq):
    train = np.ascontiguousarray(train, np.uint8)
    out = np.empty((nq, 128), np.uint8)
    _check(lib().sift_synth_queries(seed, train.ctypes.data, train.shape[0], nq, out.ctypes.data))
    return out


def shard_bounds(n, parts):
    """Contiguous near-equal split of range(n): part r = [b[r], b[r+1])."""
    out = np.zeros(parts + 1, np.int64)
    _check(lib().sift_shard_bounds(n, parts, out.ctypes.data))
    return out.tolist()


def knn2_exchange_plan(nq, parts, rank):
    """(slice_len, own_lo, own_hi) of the sharded matcher's query slices."""
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib().sift_knn2_exchange_plan(nq, parts, rank, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def min_dist_filter(knn):
    knn = np.ascontiguousarray(knn, KNN2_DTYPE)
    good = np.empty(len(knn), np.int32)
    n = C.c_int32()
    _check(lib().sift_match_min_dist_filter(knn.ctypes.data, len(knn), good.ctypes.data, C.byref(n)))
    return good[: n.value].copy()


# ---- GPU objects ----------------------------------------------------------

class Context:
    """One per GPU: owns the HIP stream every call enqueues on."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().sift_ctx_create(device, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().sift_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().sift_ctx_synchronize(self._h))

    def pipeline_runs(self):
        n = C.c_int64(0)
        _check(lib().sift_ctx_pipeline_runs(self._h, C.byref(n)))
        return n.value

    def set_single_image_capacity(self, max_keypoints):
        """Initial list capacity of extract() (0: default); extract() grows it when a frame needs more."""
        _check(lib().sift_ctx_set_single_image_capacity(self._h, max_keypoints))

    @property
    def stream(self):
        s = C.c_void_p()
        _check(lib().sift_ctx_stream(self._h, C.byref(s)))
        return s.value

    # sift_handler(name, Mat&&) + exec() + get()
    def extract(self, image):
        """image: HxW uint8 gray or HxWx3 uint8 BGR.  Returns (keypoints, descriptors float64 Nx128, info)."""
        image = np.ascontiguousarray(image, np.uint8)
        res = _Result()
        if image.ndim == 2:
            h, w = image.shape
            _check(lib().sift_extract_gray8(self._h, image.ctypes.data, w, h, image.strides[0], C.byref(res)))
        elif image.ndim == 3 and image.shape[2] == 3:
            h, w, _ = image.shape
            _check(lib().sift_extract_bgr8(self._h, image.ctypes.data, w, h, image.strides[0], C.byref(res)))
        else:
            raise SiftError(SIFT_ERR_INVALID, "image must be HxW or HxWx3 uint8")
        try:
            n = res.n
            kps = np.zeros(n, KEYPOINT_DTYPE)
            desc = np.zeros((n, 128), np.float64)
            if n:
                C.memmove(kps.ctypes.data, res.keypoints, n * KEYPOINT_DTYPE.itemsize)
                C.memmove(desc.ctypes.data, res.descriptors, n * 128 * 8)
            info = {"octaves": res.octaves, "candidates": res.counts.candidates,
                    "oriented": res.counts.oriented, "keypoints": res.counts.keypoints}
        finally:
            lib().sift_result_free(C.byref(res))
        return kps, desc, info

    def base_image(self, gray):
        """The constructor's base image (sift.cpp:40-45) of an HxW uint8 frame: 2H x 2W float64."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        out = np.empty((2 * h, 2 * w), np.float64)
        _check(lib().sift_base_image_gray8(self._h, gray.ctypes.data, w, h, gray.strides[0], out.ctypes.data))
        return out

    def extract_stages(self, gray):
        """extract() of a gray frame plus the device milliseconds of the reference's five TIMEIT stages."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        res = _Result()
        ms = (C.c_double * 5)()
        _check(lib().sift_extract_gray8_stages(self._h, gray.ctypes.data, w, h, gray.strides[0], C.byref(res), ms))
        try:
            n = res.n
            kps = np.zeros(n, KEYPOINT_DTYPE)
            desc = np.zeros((n, 128), np.float64)
            if n:
                C.memmove(kps.ctypes.data, res.keypoints, n * KEYPOINT_DTYPE.itemsize)
                C.memmove(desc.ctypes.data, res.descriptors, n * 128 * 8)
        finally:
            lib().sift_result_free(C.byref(res))
        return kps, desc, list(ms)

    def batch(self, w, h, max_frames, max_keypoints_per_frame=0):
        return Batch(self, w, h, max_frames, max_keypoints_per_frame)

    # matcher
    def knn2(self, q, t):
        """Exact top-2 + ratio test.  q, t: Nx128 uint8 (values <= 127) or float32 integer-valued."""
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        nq, nt = q.shape[0], t.shape[0]
        out = np.zeros(nq, KNN2_DTYPE)
        acc = np.zeros(nq, np.uint8)
        if q.dtype == np.uint8 and t.dtype == np.uint8:
            _check(lib().sift_knn2_u8(self._h, q.ctypes.data, nq, t.ctypes.data, nt, out.ctypes.data, acc.ctypes.data))
        else:
            q = np.ascontiguousarray(q, np.float32)
            t = np.ascontiguousarray(t, np.float32)
            _check(lib().sift_knn2_f32(self._h, q.ctypes.data, nq, t.ctypes.data, nt, out.ctypes.data, acc.ctypes.data))
        return out, acc

    def find_homography(self, src, dst, reproj_threshold=3.0, max_iters=2000, confidence=0.995):
        """findHomography(src, dst, RANSAC, reproj_threshold, mask) (main.cpp:105).  src, dst: Nx2 float32.
        Returns (H or None, mask uint8[N], info dict)."""
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
        dst = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
        if src.shape != dst.shape:
            raise ValueError("src and dst must hold the same number of points")
        n = src.shape[0]
        H = np.zeros(9, np.float64)
        mask = np.zeros(max(n, 1), np.uint8)
        info = _HomographyInfo()
        _check(lib().sift_find_homography(self._h, src.ctypes.data, dst.ctypes.data, n, reproj_threshold, max_iters,
                                          confidence, H.ctypes.data, mask.ctypes.data, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in _HomographyInfo._fields_}
        return (H.reshape(3, 3) if info.found else None), mask[:n], d

    def warp_perspective(self, src_bgr, H, dw, dh):
        """warpPerspective(src, dst, H, Size(dw, dh)) (main.cpp:115): HxWx3 uint8 -> dh x dw x 3."""
        src = np.ascontiguousarray(src_bgr, np.uint8)
        H = np.ascontiguousarray(H, np.float64).reshape(9)
        dst = np.empty((dh, dw, 3), np.uint8)
        _check(lib().sift_warp_perspective_bgr8(self._h, src.ctypes.data, src.shape[1], src.shape[0], src.strides[0],
                                                H.ctypes.data, dw, dh, dst.ctypes.data))
        return dst

    def stitch_pair(self, ref_bgr, img_bgr, H):
        """main.cpp:115-145: the cropped panorama of `ref` warped by H and `img`."""
        ref = np.ascontiguousarray(ref_bgr, np.uint8)
        img = np.ascontiguousarray(img_bgr, np.uint8)
        H = np.ascontiguousarray(H, np.float64).reshape(9)
        out = np.empty(ref.shape[0] * (ref.shape[1] + img.shape[1]) * 3, np.uint8)
        w = C.c_int()
        _check(lib().sift_stitch_pair_bgr8(self._h, ref.ctypes.data, ref.shape[1], ref.shape[0], ref.strides[0], img.ctypes.data,
                                           img.shape[1], img.shape[0], img.strides[0], H.ctypes.data, out.ctypes.data, C.byref(w)))
        return out[: ref.shape[0] * w.value * 3].reshape(ref.shape[0], w.value, 3).copy()

    def knn2_device(self, d_q, nq, d_t, nt, train_base, d_out):
        _check(lib().sift_knn2_device(self._h, d_q, nq, d_t, nt, train_base, d_out))

    def knn2_device_noalloc(self, d_q, nq, d_t, nt, train_base, d_out):
        """sift_knn2_device on the library's static scratch: no device allocation; nt <= 2,097,152."""
        _check(lib().sift_knn2_device_noalloc(self._h, d_q, nq, d_t, nt, train_base, d_out))

    def knn2_merge_device(self, d_parts, nparts, nq, d_out, d_accept):
        _check(lib().sift_knn2_merge_device(self._h, d_parts, nparts, nq, d_out, d_accept))


class Group:
    """Several GPUs of a node driven from this process: frame-sharded extraction and the train-sharded matcher."""

    def __init__(self, contexts):
        self.contexts = list(contexts)
        arr = (C.c_void_p * len(self.contexts))(*[c._h for c in self.contexts])
        self._h = C.c_void_p()
        _check(lib().sift_group_create(arr, len(self.contexts), C.byref(self._h)))

    def close(self):
        if self._h:
            lib().sift_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extract_batch(self, frames, frames_per_launch=0):
        """frames: [n, h, w] uint8 host array.  Returns a list of (keypoints, descriptors uint8 [k, 128], counts)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        res = (_FrameResult * n)()
        _check(lib().sift_extract_batch_u8(self._h, frames.ctypes.data, n, w, h, frames.strides[1], frames.strides[0],
                                           frames_per_launch, res))
        try:
            out = []
            for r in res:
                kps = np.zeros(r.n, KEYPOINT_DTYPE)
                desc = np.zeros((r.n, 128), np.uint8)
                if r.n:
                    C.memmove(kps.ctypes.data, r.keypoints, r.n * KEYPOINT_DTYPE.itemsize)
                    C.memmove(desc.ctypes.data, r.descriptors, r.n * 128)
                out.append((kps, desc, {"candidates": r.counts.candidates, "oriented": r.counts.oriented,
                                        "keypoints": r.counts.keypoints}))
        finally:
            lib().sift_frame_results_free(res, n)
        return out

    def extract_batch_raw(self, frames, frames_per_launch=0):
        """Same call without the conversion to numpy: returns (total keypoints, free()) for throughput timing."""
        n, h, w = frames.shape
        res = (_FrameResult * n)()
        _check(lib().sift_extract_batch_u8(self._h, frames.ctypes.data, n, w, h, frames.strides[1], frames.strides[0],
                                           frames_per_launch, res))
        total = sum(r.n for r in res)
        lib().sift_frame_results_free(res, n)
        return total

    def knn2(self, q, t):
        """Exact top-2 + ratio test, train set sharded over the group's GPUs.  q, t: Nx128 uint8 (values <= 127)."""
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros(q.shape[0], KNN2_DTYPE)
        acc = np.zeros(q.shape[0], np.uint8)
        _check(lib().sift_knn2_sharded(self._h, q.ctypes.data, q.shape[0], t.ctypes.data, t.shape[0], out.ctypes.data,
                                       acc.ctypes.data))
        return out, acc


class Batch:
    """Pyramid workspace for up to max_frames same-sized frames resident in HBM."""

    def __init__(self, ctx, w, h, max_frames, max_keypoints_per_frame=0):
        self.ctx = ctx
        self.w, self.h, self.max_frames = w, h, max_frames
        self._h = C.c_void_p()
        _check(lib().sift_batch_create(ctx._h, w, h, max_frames, max_keypoints_per_frame, C.byref(self._h)))
        self.octaves = num_octaves(w, h)

    def close(self):
        if self._h:
            lib().sift_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def frames_device_ptr(self):
        p = C.c_void_p()
        _check(lib().sift_batch_frames_device_ptr(self._h, C.byref(p)))
        return p.value

    def set_external_frames(self, device_ptr):
        """Read frames from a caller-owned device buffer [n][h][w] uint8 (None restores the internal one)."""
        _check(lib().sift_batch_set_external_frames(self._h, device_ptr))

    def upload(self, frame, image):
        image = np.ascontiguousarray(image, np.uint8)
        if image.ndim == 2:
            assert image.shape == (self.h, self.w)
            _check(lib().sift_batch_upload_gray8(self._h, frame, image.ctypes.data, image.strides[0]))
        else:
            assert image.shape == (self.h, self.w, 3)
            _check(lib().sift_batch_upload_bgr8(self._h, frame, image.ctypes.data, image.strides[0]))

    def run(self, n_frames):
        _check(lib().sift_batch_run(self._h, n_frames))

    def run_profiled(self, n_frames):
        t = _StageTimes()
        _check(lib().sift_batch_run_profiled(self._h, n_frames, C.byref(t)))
        return {STAGE_NAMES[i]: {"ms": t.ms[i], "launches": t.launches[i]} for i in range(len(STAGE_NAMES))}

    def descriptor_work(self, n_frames):
        """Per frame (contributing window samples, histogram updates) of the descriptor stage of the last run."""
        out = np.zeros((n_frames, 2), np.int64)
        _check(lib().sift_batch_descriptor_work(self._h, n_frames, out.ctypes.data))
        return out

    def counts(self, n_frames):
        out = np.zeros(n_frames, COUNTS_DTYPE)
        _check(lib().sift_batch_counts(self._h, n_frames, out.ctypes.data))
        return out

    def download(self, frame):
        n = C.c_int32()
        cnt = self.counts(frame + 1)[frame]
        cap = int(cnt["keypoints"])
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 128), np.uint8)
        _check(lib().sift_batch_download(self._h, frame, cap, kps.ctypes.data, desc.ctypes.data, None, C.byref(n)))
        return kps[: n.value], desc[: n.value]

    def result_device_ptrs(self, frame):
        k, d = C.c_void_p(), C.c_void_p()
        _check(lib().sift_batch_result_device_ptrs(self._h, frame, C.byref(k), C.byref(d)))
        return k.value, d.value

    def level(self, frame, oct, lvl, kind=LEVEL_GAUSS):
        ow, oh = octave_dims(self.w, self.h, oct)
        out = np.empty((oh, ow), np.float64)
        _check(lib().sift_batch_get_level(self._h, frame, oct, lvl, kind, out.ctypes.data))
        return out

    def candidates(self, frame):
        n = C.c_int32()
        _check(lib().sift_batch_get_candidates(self._h, frame, 0, None, C.byref(n)))
        out = np.zeros((n.value, 4), np.int32)
        if n.value:
            _check(lib().sift_batch_get_candidates(self._h, frame, n.value, out.ctypes.data, C.byref(n)))
        return out

    def oriented(self, frame):
        n = C.c_int32()
        _check(lib().sift_batch_get_oriented(self._h, frame, 0, None, C.byref(n)))
        out = np.zeros(n.value, KEYPOINT_DTYPE)
        if n.value:
            _check(lib().sift_batch_get_oriented(self._h, frame, n.value, out.ctypes.data, C.byref(n)))
        return out
