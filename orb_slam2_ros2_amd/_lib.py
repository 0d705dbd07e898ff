"""ctypes binding of the C-ABI in include/orbfe.h (liborbfe_hip.so).

There is no fallback: if the HIP library is missing or no device is usable this module raises.
"""
from __future__ import annotations

import ctypes as C
import threading
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbfe_hip.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28

ORBFE_OK = 0
STATUS_NAMES = {0: "OK", 1: "EBADARG", 2: "EBADSIZE", 3: "EDEVICE", 4: "ECAPACITY", 5: "ENOMEM"}
STAGE_COUNT = 8


class OrbfeError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"orbfe {STATUS_NAMES.get(status, status)}: {msg}")
        self.status = status


class ImageSizeError(OrbfeError):
    """Mirrors the reference's ImageSizeError (include/ORB_SLAM2/Error.h, thrown at ORBExtractor.cc:310-314)."""


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("n_features", C.c_int32), ("n_levels", C.c_int32),
        ("scale_factor", C.c_float), ("fast_hi", C.c_int32), ("fast_lo", C.c_int32), ("brief_pairs", C.c_void_p),
        ("blur_variant", C.c_int32), ("gray_variant", C.c_int32), ("device_id", C.c_int32), ("max_images", C.c_int32), ("stream", C.c_void_p),
    ]


class LevelInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("scale", C.c_float), ("quota", C.c_int32),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("cell_w", C.c_int32), ("cell_h", C.c_int32)]


class BaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32), ("poses", C.c_void_p),
                ("points", C.c_void_p), ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p), ("meas", C.c_void_p),
                ("is_stereo", C.c_void_p), ("info", C.c_void_p), ("huber_delta", C.c_void_p), ("fx", C.c_double),
                ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double)]


class BaSystemOut(C.Structure):
    _fields_ = [("Hpp", C.c_void_p), ("bp", C.c_void_p), ("Hll", C.c_void_p), ("bl", C.c_void_p), ("Hpl", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("k1", C.c_float), ("k2", C.c_float),
                ("p1", C.c_float), ("p2", C.c_float), ("k3", C.c_float), ("bf", C.c_float)]


class BaOptimizeOut(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("level", C.c_void_p), ("chi2", C.c_void_p), ("bad", C.c_void_p),
                ("iterations", C.c_void_p)]


class FramePose(C.Structure):
    _fields_ = [("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("min_u", C.c_float), ("max_u", C.c_float), ("min_v", C.c_float),
                ("max_v", C.c_float)]


class TrackInput(C.Structure):
    _fields_ = [("n_mp", C.c_int32), ("pos", C.c_void_p), ("view_dir", C.c_void_p), ("max_dist", C.c_void_p), ("min_dist", C.c_void_p),
                ("desc", C.c_void_p), ("flags", C.c_void_p), ("held", C.c_void_p), ("right_u", C.c_void_p), ("level_sigma2", C.c_void_p),
                ("level_inv_sigma2", C.c_void_p), ("pose_se3", C.c_void_p), ("th", C.c_float), ("ratio", C.c_float),
                ("min_threshold", C.c_int32), ("min_matches", C.c_int32)]


class MotionInput(C.Structure):
    _fields_ = [("n", C.c_int32), ("qxy", C.c_void_p), ("q_octave", C.c_void_p), ("q_min_level", C.c_void_p), ("q_max_level", C.c_void_p), ("desc", C.c_void_p),
                ("pos", C.c_void_p), ("held", C.c_void_p), ("right_u", C.c_void_p), ("level_sigma2", C.c_void_p), ("level_inv_sigma2", C.c_void_p),
                ("pose_se3", C.c_void_p), ("th", C.c_float), ("th_second", C.c_float), ("ratio", C.c_float), ("min_threshold", C.c_int32),
                ("min_matches", C.c_int32)]


class TrackOutput(C.Structure):
    _fields_ = [("assigned", C.c_void_p), ("edge_of", C.c_void_p), ("inlier", C.c_void_p), ("n_matches", C.c_void_p), ("n_edges", C.c_void_p),
                ("n_good", C.c_void_p), ("pose_out", C.c_void_p)]


class MapSummary(C.Structure):
    _fields_ = [("next_id", C.c_uint64), ("n_scale_factors", C.c_int32), ("n_keyframes", C.c_int32), ("n_mappoints", C.c_int32),
                ("n_keypoints", C.c_int64), ("n_observations", C.c_int64)]


class MapGraph(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("pose_kf_id", "pose_fixed", "poses", "point_id", "points", "edge_pose", "edge_point",
                                          "edge_feat", "meas", "is_stereo", "info", "huber_delta")]


class MapBaReport(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_poses", "n_group", "n_points", "n_edges", "n_outlier_edges", "n_keyframes_hit",
                                         "n_bad_keyframes", "written")] + [("iterations", C.c_int32 * 2),
                                                                           ("chi2_before", C.c_double), ("chi2_after", C.c_double)]


class BatchResults(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("kps", "desc", "counts", "right_u", "depth", "n_matches")]


class BaEdgeOut(C.Structure):
    _fields_ = [("error", C.c_void_p), ("chi2", C.c_void_p), ("rho", C.c_void_p), ("j_point", C.c_void_p),
                ("j_pose", C.c_void_p), ("depth_positive", C.c_void_p)]


EXPORTS = [
    "orbfe_abi_version", "orbfe_create", "orbfe_destroy", "orbfe_last_error", "orbfe_get_level_info", "orbfe_get_scale_factors",
    "orbfe_get_capacity",
    "orbfe_extract", "orbfe_extract_batch", "orbfe_extract_slot", "orbfe_extract_slot_begin", "orbfe_extract_slot_end", "orbfe_extract_slots", "orbfe_frame_stereo", "orbfe_frame_stereo_slots", "orbfe_frame_rgbd_image", "orbfe_track_motion_model", "orbfe_fetch_batch", "orbfe_fetch_stereo_batch", "orbfe_get_pyramid", "orbfe_stereo_match", "orbfe_stereo_batch_device", "orbfe_sync",
    "orbfe_host_alloc", "orbfe_host_alloc_on", "orbfe_host_free", "orbfe_recommended_hw_queues", "orbfe_stream_submit", "orbfe_stream_wait", "orbfe_stream_device_results", "orbfe_record_bytes", "orbfe_stream_pack_records",
    "orbfe_fetch_features", "orbfe_fetch_stereo", "orbfe_device_results", "orbfe_match_bruteforce", "orbfe_ba_eval_edges", "orbfe_ba_build_system", "orbfe_ba_local_optimize", "orbfe_pose_only_optimize", "orbfe_search_in_area", "orbfe_search_in_area_features", "orbfe_search_in_area_features_ex", "orbfe_extract_color", "orbfe_frame_rgbd", "orbfe_project_map_points", "orbfe_track_local_map",
    "orbfe_map_pb_summary", "orbfe_map_pb_reencode", "orbfe_map_pb_to_txt", "orbfe_map_txt_to_pb", "orbfe_map_local_graph", "orbfe_map_local_ba",
    "orbfe_profile_enable", "orbfe_profile_read", "orbfe_stage_name", "orbfe_debug_candidates",
]

_lib = None


def load() -> C.CDLL:
    """Load liborbfe_hip.so (built by orb_slam2_ros2_amd/csrc/Makefile); raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: build it with `make -C orb_slam2_ros2_amd/csrc` "
                      "(there is no CPU fallback for the front end)")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7, and a process that
    # loads the system runtime first and torch's second ends up with two runtimes, the second of which sees no
    # GPU.  If torch is installed, let it load its runtime first; liborbfe_hip.so then binds to that same copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    L.orbfe_abi_version.restype = C.c_int
    L.orbfe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.orbfe_destroy.argtypes = [vp]
    L.orbfe_destroy.restype = None
    L.orbfe_last_error.argtypes = [vp]
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_get_level_info.argtypes = [vp, i32, C.POINTER(LevelInfo)]
    L.orbfe_get_scale_factors.argtypes = [vp, vp, i32]
    L.orbfe_get_capacity.argtypes = [vp]
    L.orbfe_get_capacity.restype = i32
    L.orbfe_extract.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    L.orbfe_extract_batch.argtypes = [vp, i32, vp, C.c_size_t, vp, vp, vp]
    if hasattr(L, "orbfe_extract_slot_begin"):   # (ABI 4; tools/exp compares against libraries of earlier rounds)
        L.orbfe_extract_slot_begin.argtypes = [vp, i32, vp, C.c_size_t]
        L.orbfe_extract_slot_end.argtypes = [vp, i32, vp, vp, vp]
    L.orbfe_extract_slot.argtypes = [vp, i32, vp, C.c_size_t, vp, vp, vp]
    L.orbfe_extract_slots.argtypes = [vp, i32, i32, vp, C.c_size_t, vp, vp, vp]
    L.orbfe_frame_rgbd_image.argtypes = [vp, i32, vp, C.c_size_t, i32, vp, vp, i32, C.c_size_t, C.c_float, vp, vp, vp, vp, vp]
    L.orbfe_track_motion_model.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_frame_stereo.argtypes = [vp, vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp]
    L.orbfe_frame_stereo_slots.argtypes = [vp, i32, vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp]
    L.orbfe_fetch_batch.argtypes = [vp, i32, i32, vp, vp, vp]
    L.orbfe_fetch_stereo_batch.argtypes = [vp, i32, i32, vp, vp, vp]
    L.orbfe_get_pyramid.argtypes = [vp, i32, i32, i32, vp]
    L.orbfe_stereo_match.argtypes = [vp, i32, i32, f32, f32, vp, vp, vp, vp, vp]
    L.orbfe_stereo_batch_device.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, i32, f32, f32]
    L.orbfe_sync.argtypes = [vp]
    L.orbfe_host_alloc.argtypes = [C.c_size_t]
    L.orbfe_host_alloc.restype = vp
    L.orbfe_host_alloc_on.argtypes = [i32, C.c_size_t]
    L.orbfe_host_alloc_on.restype = vp
    L.orbfe_host_free.argtypes = [vp]
    L.orbfe_host_free.restype = None
    L.orbfe_stream_submit.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, i32, f32, f32, C.POINTER(BatchResults), C.POINTER(C.c_int64)]
    L.orbfe_stream_wait.argtypes = [vp, C.c_int64]
    L.orbfe_stream_device_results.argtypes = [vp, C.c_int64, i32] + [C.POINTER(vp)] * 6
    L.orbfe_record_bytes.argtypes = [vp]
    L.orbfe_record_bytes.restype = C.c_size_t
    L.orbfe_stream_pack_records.argtypes = [vp, C.c_int64, i32, vp]
    L.orbfe_fetch_features.argtypes = [vp, i32, vp, vp, vp]
    L.orbfe_fetch_stereo.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.orbfe_device_results.argtypes = [vp] + [C.POINTER(vp)] * 6
    L.orbfe_match_bruteforce.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.orbfe_ba_eval_edges.argtypes = [vp, C.POINTER(BaProblem), C.POINTER(BaEdgeOut)]
    L.orbfe_ba_build_system.argtypes = [vp, C.POINTER(BaProblem), vp, C.POINTER(BaSystemOut)]
    L.orbfe_ba_local_optimize.argtypes = [vp, C.POINTER(BaProblem), vp, i32, i32, vp, C.POINTER(BaOptimizeOut)]
    L.orbfe_pose_only_optimize.argtypes = [vp, i32, vp, vp, vp, vp, vp] + [C.c_double] * 5 + [vp, vp, vp]
    L.orbfe_search_in_area.argtypes = [vp, i32, i32] + [vp] * 10
    L.orbfe_search_in_area_features.argtypes = [vp, i32, vp, vp, i32] + [vp] * 10
    L.orbfe_search_in_area_features_ex.argtypes = [vp, i32, vp, vp, vp, i32] + [vp] * 11
    L.orbfe_extract_color.argtypes = [vp, vp, C.c_size_t, i32, vp, vp, vp]
    L.orbfe_frame_rgbd.argtypes = [vp, i32, C.POINTER(Camera), vp, i32, C.c_size_t, f32, vp, vp, vp]
    L.orbfe_project_map_points.argtypes = [vp, i32, vp, vp, vp, vp, C.POINTER(FramePose), C.POINTER(Camera), vp, vp, vp, vp, vp]
    L.orbfe_track_local_map.argtypes = [vp, i32, C.POINTER(FramePose), C.POINTER(Camera), C.POINTER(TrackInput), C.POINTER(TrackOutput)]
    L.orbfe_map_pb_summary.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(MapSummary)]
    L.orbfe_map_pb_reencode.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orbfe_map_pb_to_txt.argtypes = [C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orbfe_map_txt_to_pb.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orbfe_map_local_graph.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.POINTER(i32 * 4), C.POINTER(MapGraph)]
    L.orbfe_map_local_ba.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint64, C.POINTER(Camera), vp, vp, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(MapBaReport)]
    L.orbfe_profile_enable.argtypes = [vp, i32]
    L.orbfe_profile_read.argtypes = [vp, vp, vp, i32]
    L.orbfe_stage_name.argtypes = [i32]
    L.orbfe_stage_name.restype = C.c_char_p
    L.orbfe_debug_candidates.argtypes = [vp, i32, i32, vp, i32, vp]
    _lib = L
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class PinnedArray:
    """numpy view of page-locked host memory from orbfe_host_alloc (freed with the object)"""

    def __init__(self, shape, dtype, device_id=-1):
        """device_id: the HIP device whose NUMA node the pages go to (-1: the calling thread's current device)"""
        self.lib = load()
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self.p = self.lib.orbfe_host_alloc_on(int(device_id), max(n, 1))
        if not self.p:
            raise MemoryError(f"orbfe_host_alloc({n}) failed")
        self.array = np.frombuffer((C.c_uint8 * max(n, 1)).from_address(self.p), dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if getattr(self, "p", None):
            self.array = None
            self.lib.orbfe_host_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- map.pb (host-only entry points: no context, no device) ------------------------------------------
def _status(st, what):
    if st != 0:
        raise RuntimeError(f"{what}: orbfe status {st}")


def map_pb_summary(pb: bytes) -> dict:
    """Counts of an `orbslam2.MapData` file as Map::loadFromProtobuf would read it (src/Map.cc:252-313)."""
    s = MapSummary()
    _status(load().orbfe_map_pb_summary(pb, len(pb), C.byref(s)), "map_pb_summary")
    return {k: getattr(s, k) for k, _ in MapSummary._fields_}


def map_pb_reencode(pb: bytes) -> bytes:
    """Parse + serialise: the canonical libprotobuf encoding of the same content."""
    L, n = load(), C.c_size_t(0)
    _status(L.orbfe_map_pb_reencode(pb, len(pb), None, 0, C.byref(n)), "map_pb_reencode")
    out = np.zeros(max(n.value, 1), np.uint8)
    _status(L.orbfe_map_pb_reencode(pb, len(pb), ptr(out), out.size, C.byref(n)), "map_pb_reencode")
    return out[:n.value].tobytes()


def map_pb_to_txt(pb: bytes):
    """map.pb -> (KeyFrames.txt, MapPoints.txt) as Map::saveToTxtFile writes them (src/Map.cc:82-110)"""
    L, nk, nm = load(), C.c_size_t(0), C.c_size_t(0)
    _status(L.orbfe_map_pb_to_txt(pb, len(pb), None, 0, C.byref(nk), None, 0, C.byref(nm)), "map_pb_to_txt")
    kf, mp = np.zeros(max(nk.value, 1), np.uint8), np.zeros(max(nm.value, 1), np.uint8)
    _status(L.orbfe_map_pb_to_txt(pb, len(pb), ptr(kf), kf.size, C.byref(nk), ptr(mp), mp.size, C.byref(nm)), "map_pb_to_txt")
    return kf[:nk.value].tobytes().decode(), mp[:nm.value].tobytes().decode()


def map_txt_to_pb(keyframes_txt: str, mappoints_txt: str) -> bytes:
    """(KeyFrames.txt, MapPoints.txt) -> map.pb: Map::loadFromTxtFile's readers (src/Map.cc:117-165), canonical protobuf encoding"""
    L, n = load(), C.c_size_t(0)
    k, m = keyframes_txt.encode(), mappoints_txt.encode()
    _status(L.orbfe_map_txt_to_pb(k, len(k), m, len(m), None, 0, C.byref(n)), "map_txt_to_pb")
    out = np.zeros(max(n.value, 1), np.uint8)
    _status(L.orbfe_map_txt_to_pb(k, len(k), m, len(m), ptr(out), out.size, C.byref(n)), "map_txt_to_pb")
    return out[:n.value].tobytes()


def map_local_graph(pb: bytes, kf_id: int) -> dict:
    """The graph Optimizer::OptimizeLocalMap builds around keyframe kf_id (src/Optimizer.cc:232-330)."""
    L, sizes = load(), (C.c_int32 * 4)()
    _status(L.orbfe_map_local_graph(pb, len(pb), kf_id, C.byref(sizes), None), "map_local_graph")
    n_poses, n_group, n_points, n_edges = list(sizes)
    g = dict(pose_kf_id=np.zeros(n_poses, np.uint64), pose_fixed=np.zeros(n_poses, np.uint8), poses=np.zeros((n_poses, 7)),
             point_id=np.zeros(n_points, np.uint64), points=np.zeros((n_points, 3)), edge_pose=np.zeros(n_edges, np.int32),
             edge_point=np.zeros(n_edges, np.int32), edge_feat=np.zeros(n_edges, np.int32), meas=np.zeros((n_edges, 3)),
             is_stereo=np.zeros(n_edges, np.uint8), info=np.zeros(n_edges), huber_delta=np.zeros(n_edges))
    mg = MapGraph(*[ptr(g[k]).value if g[k].size else None for k, _ in MapGraph._fields_])
    _status(L.orbfe_map_local_graph(pb, len(pb), kf_id, C.byref(sizes), C.byref(mg)), "map_local_graph")
    g["n_group"] = n_group
    return g


class Context:
    """Owns one orbfe_ctx (device buffers for `max_images` image slots of one fixed geometry)."""

    def __init__(self, width, height, n_features=2000, n_levels=8, scale_factor=1.2, fast_hi=20, fast_lo=7, brief_pairs=None,
                 blur_variant=0, device_id=0, max_images=2, stream=None, gray_variant=0):
        self.lib = load()
        self._pairs = None
        if brief_pairs is not None:
            self._pairs = np.ascontiguousarray(brief_pairs, np.int8).reshape(256, 4)
        cfg = Config(width, height, n_features, n_levels, scale_factor, fast_hi, fast_lo, ptr(self._pairs), blur_variant,
                     gray_variant, device_id, max_images, stream)
        self.cfg = cfg
        h = C.c_void_p(None)
        st = self.lib.orbfe_create(C.byref(cfg), C.byref(h))
        if st != ORBFE_OK:
            msg = self.lib.orbfe_last_error(None).decode()
            raise (ImageSizeError if st == 2 else OrbfeError)(st, msg)
        self.h = h
        self.width, self.height, self.n_levels, self.max_images = width, height, n_levels, max_images
        self.requested_features = n_features
        self.n_features = int(self.lib.orbfe_get_capacity(h))   # stride of every per-image array (>= the requested count)

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != ORBFE_OK:
            msg = self.lib.orbfe_last_error(self.h).decode()
            raise (ImageSizeError if st == 2 else OrbfeError)(st, msg)

    # ---- geometry ---------------------------------------------------------------------------
    def level_info(self, level) -> LevelInfo:
        li = LevelInfo()
        self._check(self.lib.orbfe_get_level_info(self.h, level, C.byref(li)))
        return li

    def scale_factors(self) -> np.ndarray:
        out = np.zeros(self.n_levels, np.float32)
        self._check(self.lib.orbfe_get_scale_factors(self.h, ptr(out), self.n_levels))
        return out

    # ---- extraction -------------------------------------------------------------------------
    def _host_images(self, imgs):
        """the images as the library reads them (uint8 rows of one stride >= width) and that stride"""
        imgs = [np.asarray(im) for im in imgs]
        for im in imgs:
            if im.shape != (self.height, self.width):
                raise ValueError(f"image shape {im.shape} != context geometry {(self.height, self.width)}")
        # views with padded rows (a cv::Mat ROI / a row-padded camera buffer) go through as they are if they share one row stride
        strides = {im.strides[0] for im in imgs}
        if not (len(strides) == 1 and all(im.dtype == np.uint8 and im.strides[1] == 1 and im.strides[0] >= self.width for im in imgs)):
            imgs = [np.ascontiguousarray(im, np.uint8) for im in imgs]
            strides = {self.width}
        return imgs, (strides.pop() if imgs else self.width)

    def _result_arrays(self, n):
        """the [n][n_features] arrays a call's results land in: kept per context and image count (the first touch of a quarter of a
        megabyte of fresh pages was a fifth of a one-pair call); what the caller gets are copies of the filled parts"""
        cache = self.__dict__.setdefault("_res_cache", {})
        n = (threading.get_ident(), n)  # (a set per calling thread: the copies are taken after the library call has returned)
        if n not in cache:
            nf = max(self.n_features, 1)
            n_img = n[1]
            cache[n] = (np.zeros((max(n_img, 1), nf), KP_DTYPE), np.zeros((max(n_img, 1), nf, 32), np.uint8), np.zeros(max(n_img, 1), np.int32),
                        np.zeros(nf, np.float64), np.zeros(nf, np.float64))
        return cache[n]

    def extract_batch(self, imgs):
        imgs, stride = self._host_images(imgs)
        n = len(imgs)
        arr = (C.c_void_p * max(n, 1))(*[im.ctypes.data for im in imgs])
        if n <= 2:
            kps, desc, cnt = self._result_arrays(n)[:3]
        else:
            kps = np.zeros((n, max(self.n_features, 1)), KP_DTYPE)
            desc = np.zeros((n, max(self.n_features, 1), 32), np.uint8)
            cnt = np.zeros(n, np.int32)
        self._check(self.lib.orbfe_extract_batch(self.h, n, arr, stride, ptr(kps), ptr(desc), ptr(cnt)))
        return [(kps[i, :cnt[i]].copy(), desc[i, :cnt[i]].copy()) for i in range(n)]

    def frame_stereo(self, left, right, fx, bf, slot_left=None):
        """orbfe_frame_stereo: both extractions and the stereo match of one frame as one call (Frame::createStereo's device work).
        -> ((kpsL, descL), (kpsR, descR), n_matches, right_u, depth); slot_left (even): orbfe_frame_stereo_slots into that slot pair"""
        (left, right), stride = self._host_images([left, right])
        kps, desc, cnt, ru, dp = self._result_arrays(2)
        nm = C.c_int32(0)
        if slot_left is None:
            self._check(self.lib.orbfe_frame_stereo(self.h, left.ctypes.data, right.ctypes.data, stride, fx, bf, ptr(kps), ptr(desc), ptr(cnt),
                                                    ptr(ru), ptr(dp), C.byref(nm)))
        else:
            self._check(self.lib.orbfe_frame_stereo_slots(self.h, slot_left, left.ctypes.data, right.ctypes.data, stride, fx, bf, ptr(kps),
                                                          ptr(desc), ptr(cnt), ptr(ru), ptr(dp), C.byref(nm)))
        return ((kps[0, :cnt[0]].copy(), desc[0, :cnt[0]].copy()), (kps[1, :cnt[1]].copy(), desc[1, :cnt[1]].copy()), nm.value, ru.copy(), dp.copy())

    def extract(self, img):
        return self.extract_batch([img])[0]

    def extract_slot(self, slot, img):
        """one image -> slot `slot` on the slot's own stream (calls on different slots may run on different threads at once)"""
        img = np.asarray(img)
        if img.shape != (self.height, self.width):
            raise ValueError(f"image shape {img.shape} != context geometry {(self.height, self.width)}")
        if not (img.dtype == np.uint8 and img.strides[1] == 1 and img.strides[0] >= self.width):
            img = np.ascontiguousarray(img, np.uint8)
        nf = max(self.n_features, 1)
        kps = np.zeros(nf, KP_DTYPE)
        desc = np.zeros((nf, 32), np.uint8)
        n = C.c_int32(0)
        self._check(self.lib.orbfe_extract_slot(self.h, slot, img.ctypes.data, img.strides[0], ptr(kps), ptr(desc), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_slot_begin(self, slot, img):
        """orbfe_extract_slot in two halves: stage the image and enqueue the extraction on the slot's lane, return at once"""
        img = np.asarray(img)
        if img.shape != (self.height, self.width):
            raise ValueError(f"image shape {img.shape} != context geometry {(self.height, self.width)}")
        if not (img.dtype == np.uint8 and img.strides[1] == 1 and img.strides[0] >= self.width):
            img = np.ascontiguousarray(img, np.uint8)
        self._check(self.lib.orbfe_extract_slot_begin(self.h, slot, img.ctypes.data, img.strides[0]))

    def extract_slot_end(self, slot):
        """... wait for it and deliver what extract_slot delivers"""
        nf = max(self.n_features, 1)
        kps = np.zeros(nf, KP_DTYPE)
        desc = np.zeros((nf, 32), np.uint8)
        n = C.c_int32(0)
        self._check(self.lib.orbfe_extract_slot_end(self.h, slot, ptr(kps), ptr(desc), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_slots(self, slot0, imgs):
        """len(imgs) images -> the consecutive slots slot0 .. as ONE launch sequence on slot0's lane (orbfe_extract_slots)"""
        n_img = len(imgs)
        arrs = []
        for img in imgs:
            img = np.asarray(img)
            if img.shape != (self.height, self.width):
                raise ValueError(f"image shape {img.shape} != context geometry {(self.height, self.width)}")
            arrs.append(np.ascontiguousarray(img, np.uint8))
        nf = max(self.n_features, 1)
        ptrs = (C.c_void_p * n_img)(*[a.ctypes.data for a in arrs])
        kps = np.zeros((n_img, nf), KP_DTYPE)
        desc = np.zeros((n_img, nf, 32), np.uint8)
        n = np.zeros(n_img, np.int32)
        self._check(self.lib.orbfe_extract_slots(self.h, slot0, n_img, ptrs, self.width, ptr(kps), ptr(desc), ptr(n)))
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(n_img)]

    def fetch_batch(self, slot0, n_slots):
        """packed results of slots [slot0, slot0 + n_slots): (kps [n][NF], desc [n][NF][32], counts [n])"""
        nf = max(self.n_features, 1)
        kps = np.zeros((n_slots, nf), KP_DTYPE)
        desc = np.zeros((n_slots, nf, 32), np.uint8)
        cnt = np.zeros(n_slots, np.int32)
        self._check(self.lib.orbfe_fetch_batch(self.h, slot0, n_slots, ptr(kps), ptr(desc), ptr(cnt)))
        return kps, desc, cnt

    def fetch_stereo_batch(self, pair0, n_pairs):
        """packed stereo results of pairs [pair0, pair0 + n_pairs): (right_u [n][NF], depth [n][NF], n_matches [n])"""
        nf = max(self.n_features, 1)
        ru = np.zeros((n_pairs, nf), np.float64)
        dp = np.zeros((n_pairs, nf), np.float64)
        nm = np.zeros(n_pairs, np.int32)
        self._check(self.lib.orbfe_fetch_stereo_batch(self.h, pair0, n_pairs, ptr(ru), ptr(dp), ptr(nm)))
        return ru, dp, nm

    def pyramid(self, slot, level, blurred=False) -> np.ndarray:
        li = self.level_info(level)
        out = np.zeros((li.height, li.width), np.uint8)
        self._check(self.lib.orbfe_get_pyramid(self.h, slot, level, int(blurred), ptr(out)))
        return out

    def debug_candidates(self, slot, level) -> np.ndarray:
        n = C.c_int32(0)
        self._check(self.lib.orbfe_debug_candidates(self.h, slot, level, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.float32)
        self._check(self.lib.orbfe_debug_candidates(self.h, slot, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    # ---- stereo -------------------------------------------------------------------------------
    def stereo_match(self, slot_left, slot_right, fx, bf):
        nf = max(self.n_features, 1)
        ru = np.zeros(nf, np.float64)
        dp = np.zeros(nf, np.float64)
        br = np.zeros(nf, np.int32)
        bd = np.zeros(nf, np.int32)
        nm = C.c_int32(0)
        self._check(self.lib.orbfe_stereo_match(self.h, slot_left, slot_right, fx, bf, ptr(ru), ptr(dp), C.byref(nm), ptr(br), ptr(bd)))
        return nm.value, ru, dp, br, bd

    def stereo_batch_device(self, d_left_ptr, d_right_ptr, stride, image_pitch, n_pairs, fx, bf):
        self._check(self.lib.orbfe_stereo_batch_device(self.h, d_left_ptr, d_right_ptr, stride, image_pitch, n_pairs, fx, bf))

    def sync(self):
        self._check(self.lib.orbfe_sync(self.h))

    # ---- host-image stream -----------------------------------------------------------------------
    def alloc_batch_results(self, n_pairs, pinned=True):
        """result arrays of one batch for stream_submit: dict of numpy arrays (page-locked unless pinned=False)"""
        nf = max(self.n_features, 1)
        spec = dict(kps=((2 * n_pairs, nf), KP_DTYPE), desc=((2 * n_pairs, nf, 32), np.uint8), counts=((2 * n_pairs,), np.int32),
                    right_u=((n_pairs, nf), np.float64), depth=((n_pairs, nf), np.float64), n_matches=((n_pairs,), np.int32))
        out, keep = {}, []
        for k, (shape, dt) in spec.items():
            if pinned:
                pa = PinnedArray(shape, dt)
                keep.append(pa)
                out[k] = pa.array
            else:
                out[k] = np.zeros(shape, dt)
        out["_pinned"] = keep
        return out

    def stream_submit(self, left, right, n_pairs, fx, bf, out, stride=None, image_pitch=None):
        """left / right: uint8 arrays [n_pairs, H, W] (C-contiguous; page-locked for true overlap) -> ticket"""
        stride = stride or self.width
        image_pitch = image_pitch or stride * self.height
        out = out or {}
        res = BatchResults(*[out[k].ctypes.data if out.get(k) is not None else None
                             for k in ("kps", "desc", "counts", "right_u", "depth", "n_matches")])
        t = C.c_int64(-1)
        self._check(self.lib.orbfe_stream_submit(self.h, left.ctypes.data, right.ctypes.data, stride, image_pitch, n_pairs, fx, bf,
                                                 C.byref(res), C.byref(t)))
        return t.value

    def stream_wait(self, ticket):
        self._check(self.lib.orbfe_stream_wait(self.h, ticket))

    def record_bytes(self) -> int:
        return int(self.lib.orbfe_record_bytes(self.h))

    def stream_pack_records(self, ticket, n_pairs, d_records_ptr):
        """frame records of a live ticket -> device memory at d_records_ptr (n_pairs * record_bytes() bytes); returns when complete"""
        self._check(self.lib.orbfe_stream_pack_records(self.h, ticket, n_pairs, d_records_ptr))

    def stream_device_results(self, ticket, n_pairs):
        """device pointers (ints) of the packed results of a live ticket: dict kps, desc, counts, right_u, depth, n_match"""
        ps = [C.c_void_p(None) for _ in range(6)]
        self._check(self.lib.orbfe_stream_device_results(self.h, ticket, n_pairs, *[C.byref(p) for p in ps]))
        return dict(zip(["kps", "desc", "counts", "right_u", "depth", "n_match"], [p.value for p in ps]))

    def fetch_features(self, slot):
        nf = max(self.n_features, 1)
        kps = np.zeros(nf, KP_DTYPE)
        desc = np.zeros((nf, 32), np.uint8)
        n = C.c_int32(0)
        self._check(self.lib.orbfe_fetch_features(self.h, slot, ptr(kps), ptr(desc), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def fetch_stereo(self, pair):
        nf = max(self.n_features, 1)
        ru = np.zeros(nf, np.float64)
        dp = np.zeros(nf, np.float64)
        br = np.zeros(nf, np.int32)
        bd = np.zeros(nf, np.int32)
        nm = C.c_int32(0)
        self._check(self.lib.orbfe_fetch_stereo(self.h, pair, ptr(ru), ptr(dp), C.byref(nm), ptr(br), ptr(bd)))
        return nm.value, ru, dp, br, bd

    def device_results(self):
        ps = [C.c_void_p(None) for _ in range(6)]
        self._check(self.lib.orbfe_device_results(self.h, *[C.byref(p) for p in ps]))
        return dict(zip(["kps", "desc", "counts", "right_u", "depth", "n_match"], [p.value for p in ps]))

    # ---- matching -----------------------------------------------------------------------------
    def match_bruteforce(self, q, t, cand_offsets=None, cand_idx=None):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = q.shape[0], t.shape[0]
        if cand_offsets is not None:
            cand_offsets = np.ascontiguousarray(cand_offsets, np.uint32)
            cand_idx = np.ascontiguousarray(cand_idx if cand_idx is not None else np.zeros(0), np.uint32)
            assert cand_offsets.size == nq + 1
        bi = np.zeros(max(nq, 1), np.int32)
        bd = np.zeros(max(nq, 1), np.int32)
        sd = np.zeros(max(nq, 1), np.int32)
        self._check(self.lib.orbfe_match_bruteforce(self.h, ptr(q), nq, ptr(t), nt, ptr(cand_offsets), ptr(cand_idx), ptr(bi), ptr(bd), ptr(sd)))
        return bi[:nq], bd[:nq], sd[:nq]

    def search_in_area(self, slot, qxy, radius, min_level, max_level, q_desc, exclude=None):
        qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
        nq = qxy.shape[0]
        radius = np.ascontiguousarray(radius, np.float32)
        min_level = np.ascontiguousarray(min_level, np.int8)
        max_level = np.ascontiguousarray(max_level, np.int8)
        q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        ex = None
        if exclude is not None:
            ex = np.zeros(max(self.n_features, 1), np.uint8)
            ex[:len(exclude)] = np.asarray(exclude, np.uint8)
        out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
        self._check(self.lib.orbfe_search_in_area(self.h, slot, nq, ptr(qxy), ptr(radius), ptr(min_level), ptr(max_level), ptr(q_desc),
                                                  ptr(ex), *[ptr(o) for o in out]))
        return tuple(o[:nq] for o in out)

    def search_in_area_features(self, t_kps, t_desc, qxy, radius, min_level, max_level, q_desc, exclude=None, bounds=None, want_hits=False):
        """orbfe_search_in_area against a caller-supplied feature set (a KeyFrame's keypoints [KP_DTYPE] and descriptors).
        bounds = (min_u, max_u, min_v, max_v) of the target frame (None: the image); want_hits: also return, per excluded feature, how many
        queries had it in their window (orbfe_search_in_area_features_ex)."""
        t_kps = np.ascontiguousarray(t_kps, KP_DTYPE)
        t_desc = np.ascontiguousarray(t_desc, np.uint8).reshape(-1, 32)
        nt = t_kps.shape[0]
        qxy = np.ascontiguousarray(qxy, np.float32).reshape(-1, 2)
        nq = qxy.shape[0]
        radius = np.ascontiguousarray(radius, np.float32)
        min_level = np.ascontiguousarray(min_level, np.int8)
        max_level = np.ascontiguousarray(max_level, np.int8)
        q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        ex = None
        if exclude is not None:
            ex = np.zeros(max(nt, 1), np.uint8)
            ex[:len(exclude)] = np.asarray(exclude, np.uint8)
        out = [np.zeros(max(nq, 1), np.int32) for _ in range(4)]
        if bounds is None and not want_hits:
            self._check(self.lib.orbfe_search_in_area_features(self.h, nt, ptr(t_kps), ptr(t_desc), nq, ptr(qxy), ptr(radius), ptr(min_level),
                                                               ptr(max_level), ptr(q_desc), ptr(ex), *[ptr(o) for o in out]))
            return tuple(o[:nq] for o in out)
        bnd = None if bounds is None else np.ascontiguousarray(bounds, np.float32).reshape(4)
        hits = np.zeros(max(nt, 1), np.int32) if want_hits else None
        self._check(self.lib.orbfe_search_in_area_features_ex(self.h, nt, ptr(t_kps), ptr(t_desc), ptr(bnd), nq, ptr(qxy), ptr(radius),
                                                              ptr(min_level), ptr(max_level), ptr(q_desc), ptr(ex), *[ptr(o) for o in out], ptr(hits)))
        res = tuple(o[:nq] for o in out)
        return res + (hits[:nt],) if want_hits else res

    # ---- BA -------------------------------------------------------------------------------------
    def ba_eval_edges(self, poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf,
                      jacobians=True):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        is_stereo = np.ascontiguousarray(is_stereo, np.uint8)
        info = np.ascontiguousarray(info, np.float64)
        huber_delta = np.ascontiguousarray(huber_delta, np.float64)
        E = edge_pose.size
        prob = BaProblem(poses.shape[0], points.shape[0], E, ptr(poses).value, ptr(points).value, ptr(edge_pose).value,
                         ptr(edge_point).value, ptr(meas).value, ptr(is_stereo).value, ptr(info).value, ptr(huber_delta).value,
                         fx, fy, cx, cy, bf)
        out = dict(error=np.zeros((E, 3)), chi2=np.zeros(E), rho=np.zeros((E, 2)), depth_positive=np.zeros(E, np.uint8))
        if jacobians:
            out["j_point"] = np.zeros((E, 3, 3))
            out["j_pose"] = np.zeros((E, 3, 6))
        o = BaEdgeOut(ptr(out["error"]).value, ptr(out["chi2"]).value, ptr(out["rho"]).value,
                      ptr(out["j_point"]).value if jacobians else None, ptr(out["j_pose"]).value if jacobians else None,
                      ptr(out["depth_positive"]).value)
        self._check(self.lib.orbfe_ba_eval_edges(self.h, C.byref(prob), C.byref(o)))
        return out

    def ba_build_system(self, poses, points, edge_pose, edge_point, meas, is_stereo, info, huber_delta, fx, fy, cx, cy, bf,
                        pose_fixed=None, want_hpl=True):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        edge_pose = np.ascontiguousarray(edge_pose, np.int32)
        edge_point = np.ascontiguousarray(edge_point, np.int32)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        is_stereo = np.ascontiguousarray(is_stereo, np.uint8)
        info = np.ascontiguousarray(info, np.float64)
        huber_delta = np.ascontiguousarray(huber_delta, np.float64)
        fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
        nk, npt, E = poses.shape[0], points.shape[0], edge_pose.size
        prob = BaProblem(nk, npt, E, ptr(poses).value, ptr(points).value, ptr(edge_pose).value, ptr(edge_point).value,
                         ptr(meas).value, ptr(is_stereo).value, ptr(info).value, ptr(huber_delta).value, fx, fy, cx, cy, bf)
        out = dict(Hpp=np.zeros((nk, 6, 6)), bp=np.zeros((nk, 6)), Hll=np.zeros((npt, 3, 3)), bl=np.zeros((npt, 3)))
        if want_hpl:
            out["Hpl"] = np.zeros((E, 6, 3))
        o = BaSystemOut(ptr(out["Hpp"]).value, ptr(out["bp"]).value, ptr(out["Hll"]).value, ptr(out["bl"]).value,
                        ptr(out["Hpl"]).value if want_hpl else None)
        self._check(self.lib.orbfe_ba_build_system(self.h, C.byref(prob), ptr(fixed), C.byref(o)))
        return out

    def ba_local_optimize(self, prob, pose_fixed=None, iters_first=5, iters_second=10, stop=None):
        """prob: dict with the orbfe_ba_problem arrays (see ba_synth.make_problem) -> dict(poses, points, level, chi2, bad, iters).
        stop: a numpy uint8 array of one element another thread may set to 1 while the call runs (the reference's `bool& isStop`)"""
        f64 = lambda a, shape: np.ascontiguousarray(a, np.float64).reshape(shape)
        poses, points, meas = f64(prob["poses"], (-1, 7)), f64(prob["points"], (-1, 3)), f64(prob["meas"], (-1, 3))
        info, delta = f64(prob["info"], -1), f64(prob["huber_delta"], -1)
        ek = np.ascontiguousarray(prob["edge_pose"], np.int32)
        ep = np.ascontiguousarray(prob["edge_point"], np.int32)
        st = np.ascontiguousarray(prob["is_stereo"], np.uint8)
        fixed = None if pose_fixed is None else np.ascontiguousarray(pose_fixed, np.uint8)
        nk, npt, E = poses.shape[0], points.shape[0], ek.size
        bp = BaProblem(nk, npt, E, ptr(poses).value, ptr(points).value, ptr(ek).value, ptr(ep).value, ptr(meas).value, ptr(st).value,
                       ptr(info).value, ptr(delta).value, prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
        out = dict(poses=np.zeros((nk, 7)), points=np.zeros((npt, 3)), level=np.zeros(max(E, 1), np.uint8), chi2=np.zeros(max(E, 1)),
                   bad=np.zeros(max(E, 1), np.uint8), iters=np.zeros(2, np.int32))
        o = BaOptimizeOut(ptr(out["poses"]).value, ptr(out["points"]).value, ptr(out["level"]).value, ptr(out["chi2"]).value,
                          ptr(out["bad"]).value, ptr(out["iters"]).value)
        if stop is not None:
            assert isinstance(stop, np.ndarray) and stop.dtype == np.uint8 and stop.size >= 1
        self._check(self.lib.orbfe_ba_local_optimize(self.h, C.byref(bp), ptr(fixed), iters_first, iters_second, ptr(stop), C.byref(o)))
        for k in ("level", "chi2", "bad"):
            out[k] = out[k][:E]
        return out

    def project_map_points(self, pos, view_dir, max_dist, min_dist, Rcw, tcw, cam, bounds):
        """MapPoint::isInVision + predictLevel for n map points (MapPoint.cc:141-201); cam = (fx, fy, cx, cy), bounds = (minU, maxU,
        minV, maxV) -> dict(uv, distance, cos_theta, level, visible)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, view_dir, max_dist, min_dist = f32(pos).reshape(-1, 3), f32(view_dir).reshape(-1, 3), f32(max_dist), f32(min_dist)
        n = pos.shape[0]
        m = max(n, 1)
        out = dict(uv=np.zeros((m, 2), np.float32), distance=np.zeros(m, np.float32), cos_theta=np.zeros(m, np.float32),
                   level=np.zeros(m, np.int8), visible=np.zeros(m, np.uint8))
        fp = FramePose((C.c_float * 9)(*f32(Rcw).reshape(9)), (C.c_float * 3)(*f32(tcw).reshape(3)), *[float(np.float32(b)) for b in bounds])
        cm = Camera(*[float(np.float32(v)) for v in cam], 0, 0, 0, 0, 0, 0)
        self._check(self.lib.orbfe_project_map_points(self.h, n, ptr(pos), ptr(view_dir), ptr(max_dist), ptr(min_dist), C.byref(fp),
                                                      C.byref(cm), ptr(out["uv"]), ptr(out["distance"]), ptr(out["cos_theta"]),
                                                      ptr(out["level"]), ptr(out["visible"])))
        return {k: v[:n] for k, v in out.items()}

    def track_local_map(self, slot, pos, view_dir, max_dist, min_dist, desc, flags, Rcw, tcw, cam, bounds, pose_se3, level_sigma2,
                        level_inv_sigma2, held=None, right_u=None, th=3.0, ratio=0.8, min_threshold=50, min_matches=30):
        """Tracking::trackLocalMap's device work as one call (orbfe_track_local_map): searchByProjection(frame, map points, th) against
        the features of `slot`, then OptimizePoseOnly on what the frame holds.  cam = (fx, fy, cx, cy, bf); bounds = (minU, maxU, minV,
        maxV) -> dict(assigned, edge_of, inlier, n_matches, n_edges, n_good, pose)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, view_dir, max_dist, min_dist = f32(pos).reshape(-1, 3), f32(view_dir).reshape(-1, 3), f32(max_dist), f32(min_dist)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        flags = np.ascontiguousarray(flags, np.uint8)
        n, NF = pos.shape[0], self.n_features
        held = None if held is None else np.ascontiguousarray(held, np.int32)
        right_u = None if right_u is None else np.ascontiguousarray(right_u, np.float64)
        if (held is not None and held.size != NF) or (right_u is not None and right_u.size != NF):
            raise ValueError("held / right_u must have n_features entries")
        s2, is2, p0 = f32(level_sigma2), f32(level_inv_sigma2), np.ascontiguousarray(pose_se3, np.float64)
        fp = FramePose((C.c_float * 9)(*f32(Rcw).reshape(9)), (C.c_float * 3)(*f32(tcw).reshape(3)), *[float(np.float32(b)) for b in bounds])
        cm = Camera(*[float(np.float32(v)) for v in cam[:4]], 0, 0, 0, 0, 0, float(np.float32(cam[4])))
        out = dict(assigned=np.zeros(NF, np.int32), edge_of=np.zeros(NF, np.int32), inlier=np.zeros(NF, np.uint8), pose=np.zeros(7))
        nm, ne, ng = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        ti = TrackInput(n, *[ptr(a) for a in (pos, view_dir, max_dist, min_dist, desc, flags, held, right_u, s2, is2, p0)], th, ratio,
                        min_threshold, min_matches)
        to = TrackOutput(ptr(out["assigned"]), ptr(out["edge_of"]), ptr(out["inlier"]), C.cast(C.byref(nm), C.c_void_p),
                         C.cast(C.byref(ne), C.c_void_p), C.cast(C.byref(ng), C.c_void_p), ptr(out["pose"]))
        self._check(self.lib.orbfe_track_local_map(self.h, slot, C.byref(fp), C.byref(cm), C.byref(ti), C.byref(to)))
        out.update(n_matches=nm.value, n_edges=ne.value, n_good=ng.value)
        return out

    def track_motion_model(self, slot, qxy, q_octave, q_min_level, q_max_level, desc, pos, cam, bounds, pose_se3, level_sigma2, level_inv_sigma2,
                           held=None, right_u=None, th=15.0, th_second=30.0, ratio=0.9, min_threshold=50, min_matches=20):
        """The middle of Tracking::trackMotionModel as one call (orbfe_track_motion_model): searchByProjection(frame, lastFrame, th) around the
        last frame's feature positions (radius th * sigma2(octave)) against the features of `slot` (+ the th_second search when fewer than
        min_matches matched), then
        OptimizePoseOnly.  cam = (fx, fy, cx, cy, bf); bounds = (minU, maxU, minV, maxV)
        -> dict(assigned, edge_of, inlier, excluded_hits, query_matches, n_matches, n_edges, n_good, pose, passes)"""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        qxy, pos = f32(qxy).reshape(-1, 2), f32(pos).reshape(-1, 3)
        lo, hi = np.ascontiguousarray(q_min_level, np.int8), np.ascontiguousarray(q_max_level, np.int8)
        octv = np.ascontiguousarray(q_octave, np.int8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n, NF = qxy.shape[0], self.n_features
        if not (pos.shape[0] == n and lo.size == n and hi.size == n and octv.size == n and desc.shape[0] == n):
            raise ValueError("qxy / levels / desc / pos disagree in length")
        held = None if held is None else np.ascontiguousarray(held, np.int32)
        right_u = None if right_u is None else np.ascontiguousarray(right_u, np.float64)
        if (held is not None and held.size != NF) or (right_u is not None and right_u.size != NF):
            raise ValueError("held / right_u must have n_features entries")
        s2, is2, p0 = f32(level_sigma2), f32(level_inv_sigma2), np.ascontiguousarray(pose_se3, np.float64)
        b4 = f32(bounds)
        cm = Camera(*[float(np.float32(v)) for v in cam[:4]], 0, 0, 0, 0, 0, float(np.float32(cam[4])))
        out = dict(assigned=np.zeros(NF, np.int32), edge_of=np.zeros(NF, np.int32), inlier=np.zeros(NF, np.uint8), pose=np.zeros(7),
                   excluded_hits=np.zeros(NF, np.int32), query_matches=np.zeros(max(n, 1), np.int32))
        nm, ne, ng, np_ = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        mi = MotionInput(n, *[ptr(a) for a in (qxy, octv, lo, hi, desc, pos, held, right_u, s2, is2, p0)], th, th_second, ratio, min_threshold, min_matches)
        to = TrackOutput(ptr(out["assigned"]), ptr(out["edge_of"]), ptr(out["inlier"]), C.cast(C.byref(nm), C.c_void_p),
                         C.cast(C.byref(ne), C.c_void_p), C.cast(C.byref(ng), C.c_void_p), ptr(out["pose"]))
        self._check(self.lib.orbfe_track_motion_model(self.h, slot, ptr(b4), C.byref(cm), C.byref(mi), C.byref(to), ptr(out["excluded_hits"]),
                                                      ptr(out["query_matches"]), C.byref(np_)))
        out.update(n_matches=nm.value, n_edges=ne.value, n_good=ng.value, passes=np_.value, query_matches=out["query_matches"][:n])
        return out

    def map_local_ba(self, pb: bytes, kf_id: int, fx, fy, cx, cy, bf):
        """Optimizer::OptimizeLocalMap around keyframe kf_id of a map.pb -> (updated map.pb bytes, report dict)"""
        cam = Camera(fx, fy, cx, cy, 0, 0, 0, 0, 0, bf)
        n, rep = C.c_size_t(0), MapBaReport()
        out = np.zeros(len(pb) + 4096, np.uint8)  # the result never grows: observations are only erased (-1 takes more bytes than an id, hence the slack)
        st = self.lib.orbfe_map_local_ba(self.h, pb, len(pb), kf_id, C.byref(cam), None, ptr(out), out.size, C.byref(n), C.byref(rep))
        if st == 4:  # ORBFE_ECAPACITY
            out = np.zeros(n.value, np.uint8)
            st = self.lib.orbfe_map_local_ba(self.h, pb, len(pb), kf_id, C.byref(cam), None, ptr(out), out.size, C.byref(n), C.byref(rep))
        self._check(st)
        r = {k: getattr(rep, k) for k, _ in MapBaReport._fields_ if k != "iterations"}
        r["iterations"] = list(rep.iterations)
        return out[:n.value].tobytes(), r

    def pose_only_optimize(self, Xw, meas, info, sigma2, pose, fx, fy, cx, cy, bf):
        Xw = np.ascontiguousarray(Xw, np.float64).reshape(-1, 3)
        meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 3)
        info = np.ascontiguousarray(info, np.float64)
        sigma2 = np.ascontiguousarray(sigma2, np.float32)
        pose = np.ascontiguousarray(pose, np.float64)
        n = Xw.shape[0]
        out = np.zeros(7)
        inl = np.zeros(max(n, 1), np.uint8)
        ng = C.c_int32(0)
        self._check(self.lib.orbfe_pose_only_optimize(self.h, n, ptr(Xw), ptr(meas), ptr(info), ptr(sigma2), ptr(pose), fx, fy, cx, cy,
                                                      bf, ptr(out), ptr(inl), C.byref(ng)))
        return ng.value, out, inl[:n].astype(bool)

    # ---- frame glue ------------------------------------------------------------------------------
    def extract_color(self, img: np.ndarray, order: int):
        """img: (h, w, 3) uint8, order 1 = RGB / 2 = BGR -> (keypoints, descriptors) of slot 0"""
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.height, self.width, 3)
        kps = np.zeros(max(self.n_features, 1), KP_DTYPE)
        desc = np.zeros((max(self.n_features, 1), 32), np.uint8)
        n = C.c_int32(0)
        self._check(self.lib.orbfe_extract_color(self.h, ptr(img), img.strides[0], order, ptr(kps), ptr(desc), C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def frame_rgbd(self, slot, cam: dict, depth=None, depth_scale=1.0):
        """cam: dict(fx fy cx cy k1 k2 p1 p2 k3 bf) -> (undistorted keypoints, depth, right_u); depth None: undistortion only"""
        cm = Camera(*[float(cam[k]) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")])
        nf = max(self.n_features, 1)
        kps = np.zeros(nf, KP_DTYPE)
        d, ru = np.zeros(nf), np.zeros(nf)
        dtype, stride = 0, 0
        if depth is not None:
            depth = np.ascontiguousarray(depth)
            assert depth.dtype in (np.uint16, np.float32) and depth.shape == (self.height, self.width)
            dtype, stride = (0 if depth.dtype == np.uint16 else 1), depth.strides[0]
        self._check(self.lib.orbfe_frame_rgbd(self.h, slot, C.byref(cm), ptr(depth), dtype, stride, depth_scale, ptr(kps), ptr(d), ptr(ru)))
        return kps, d, ru   # [n_features] each; entries past the slot's keypoint count are zero (kps) / -1

    def frame_rgbd_image(self, img, cam: dict, depth=None, depth_scale=1.0, color_order=0, slot=0):
        """orbfe_frame_rgbd_image: (colour -> gray,) extraction, undistortion and the depth / rightU lookup of one RGB-D frame as one call
        (Frame::createRGBD's device work).  img: (h, w) uint8, or (h, w, 3) with color_order 1 (RGB) / 2 (BGR).
        -> (undistorted keypoints [n], descriptors [n], depth [n_features], right_u [n_features])"""
        img = np.asarray(img)
        want = (self.height, self.width, 3) if color_order else (self.height, self.width)
        if img.shape != want:
            raise ValueError(f"image shape {img.shape} != {want}")
        if not (img.dtype == np.uint8 and img.flags.c_contiguous):
            img = np.ascontiguousarray(img, np.uint8)
        cm = Camera(*[float(cam[k]) for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3", "bf")])
        nf = max(self.n_features, 1)
        kps, desc = np.zeros(nf, KP_DTYPE), np.zeros((nf, 32), np.uint8)
        d, ru = np.zeros(nf), np.zeros(nf)
        n = C.c_int32(0)
        dtype, stride = 0, 0
        if depth is not None:
            depth = np.ascontiguousarray(depth)
            assert depth.dtype in (np.uint16, np.float32) and depth.shape == (self.height, self.width)
            dtype, stride = (0 if depth.dtype == np.uint16 else 1), depth.strides[0]
        self._check(self.lib.orbfe_frame_rgbd_image(self.h, slot, img.ctypes.data, img.strides[0], color_order, C.byref(cm), ptr(depth), dtype,
                                                    stride, depth_scale, ptr(kps), ptr(desc), C.byref(n), ptr(d), ptr(ru)))
        return kps[:n.value].copy(), desc[:n.value].copy(), d, ru

    # ---- instrumentation ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """False / 0: off, True / 1: every stage timed alone, 2 + stage id: that stage only, in the production schedule"""
        self._check(self.lib.orbfe_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        ms = np.zeros(STAGE_COUNT, np.float64)
        n = np.zeros(STAGE_COUNT, np.int64)
        self._check(self.lib.orbfe_profile_read(self.h, ptr(ms), ptr(n), int(reset)))
        names = [self.lib.orbfe_stage_name(i).decode() for i in range(STAGE_COUNT)]
        return {names[i]: (float(ms[i]), int(n[i])) for i in range(STAGE_COUNT)}
