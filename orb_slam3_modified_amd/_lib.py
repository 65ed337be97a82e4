"""ctypes loader for liborbx.so (the C-ABI drop-in boundary, include/orbx.h).

The shared library is built in-tree by `python -m orb_slam3_modified_amd.build` (hipcc, gfx950).
There is no Python/CPU fallback: if the library is missing the import of any compute class raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB", os.path.join(_HERE, "liborbx.so"))   # ORBX_LIB: experiment builds (tools/)
DEBUG_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "liborbx_debug.so")   # the diagnostic ABI (include/orbx_debug.h): tests and tools only

ORBX_OK, ORBX_E_INVALID, ORBX_E_EMPTY, ORBX_E_DEVICE, ORBX_E_CAPACITY, ORBX_E_FORMAT = 0, -1, -2, -3, -4, -5
NUM_KERNELS = 6

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbxError(RuntimeError):
    def __init__(self, code: int, msg: str = ""):
        super().__init__(f"orbx error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built — run `python -m orb_slam3_modified_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, sz, f32 = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    ip = C.POINTER(C.c_int)
    sig = {
        "orbx_create": (i32, [C.POINTER(vp), i32, f32, i32, i32, i32, i32]),
        "orbx_destroy": (None, [vp]),
        "orbx_last_error": (C.c_char_p, [vp]),
        "orbx_keypoint_capacity": (i32, [vp]),
        "orbx_levels": (i32, [vp]),
        "orbx_scale_tables": (i32, [vp, vp, vp, vp, vp, vp]),
        "orbx_extract": (i32, [vp, vp, i32, i32, sz, i32, i32, vp, vp, ip, ip]),
        "orbx_extract_color": (i32, [vp, vp, i32, i32, C.c_size_t, i32, i32, i32, i32, vp, vp, ip, ip]),
        "orbx_extract_resized": (i32, [vp, vp, i32, i32, sz, i32, i32, i32, i32, vp, vp, ip, ip]),
        "orbx_extract_batch_device": (i32, [vp, vp, i32, i32, i32, sz, sz, i32, i32, vp, vp, vp, vp]),
        "orbx_extract_batch": (i32, [vp, vp, i32, i32, i32, sz, sz, i32, i32, vp, vp, vp]),
        "orbx_pyramid_level": (i32, [vp, i32, i32, vp, sz, ip, ip]),
        "orbx_set_host_pyramid": (i32, [vp, i32]),
        "orbx_host_pyramid_level": (i32, [vp, i32, C.POINTER(vp), C.POINTER(sz), ip, ip]),
        "orbx_profile_enable": (i32, [vp, i32]),
        "orbx_profile_read": (i32, [vp, vp, vp]),
        "orbx_kernel_name": (C.c_char_p, [i32]),
        "orbx_hamming": (i32, [vp, vp]),
        "orbx_nn_csr": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
        "orbx_knn2_allpairs": (i32, [vp, vp, i32, vp, i32, vp, vp]),
        "orbx_nn_csr_device": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
        "orbx_knn2_allpairs_device": (i32, [vp, vp, i32, vp, i32, vp, vp, vp]),
        "orbx_features_in_area": (i32, [vp, vp, i32, f32, f32, f32, f32, vp, vp, vp, vp, vp, i32, vp, vp, i32]),
        "orbx_search_for_initialization": (i32, [vp, vp, vp, i32, vp, vp, i32, f32, f32, f32, f32, vp, i32, f32, i32, vp, ip]),
        "orbx_window_search": (i32, [vp, vp, vp, i32, f32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
        "orbx_window_search_grid": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
        "orbx_window_nearest": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "orbx_target_create": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, C.POINTER(C.c_void_p)]),
        "orbx_target_assign": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32]),
        "orbx_target_destroy": (None, [vp]),
        "orbx_target_size": (i32, [vp]),
        "orbx_target_search": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
        "orbx_target_nearest": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "orbx_undistort_keypoints": (i32, [vp, vp, i32, f32, f32, f32, f32, vp, i32, vp]),
        "orbx_undistort_keypoints_device": (i32, [vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, i32, vp, vp]),
        "orbx_search_by_projection": (i32, [vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32,
                                            vp, ip]),
        "orbx_search_by_projection_last": (i32, [vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                                 f32, i32, i32, vp, ip]),
        "orbx_search_by_bow": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, f32, i32, vp, ip]),
        "orbx_stereo_matches": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, ip]),
        "orbx_voc_load_text": (i32, [vp, C.c_char_p, C.POINTER(vp)]),
        "orbx_voc_create": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp)]),
        "orbx_set_option": (i32, [vp, C.c_char_p, i32]),
        "orbx_get_option": (i32, [vp, C.c_char_p]),
        "orbx_reserve": (i32, [vp, i32, i32, i32]),
        "orbx_voc_save_text": (i32, [vp, C.c_char_p]),
        "orbx_voc_save_binary": (i32, [vp, C.c_char_p]),
        "orbx_voc_load_binary": (i32, [vp, C.c_char_p, C.POINTER(vp)]),
        "orbx_voc_destroy": (None, [vp]),
        "orbx_voc_info": (i32, [vp, ip, ip, ip, ip]),
        "orbx_bow_transform": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "orbx_bow_transform_published": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "orbx_nn_groups": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_int)]),
        "orbx_bow_transform_device": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
        "orbx_bow_finalize": (i32, [vp, vp, vp, i32, vp, vp, ip]),
        "orbx_bow_score_l1": (C.c_double, [vp, vp, i32, vp, vp, i32]),
        "orbx_bow_score_l1_batch": (i32, [vp, vp, vp, i32, vp, vp, vp, i32, vp]),
        "orbx_kfdb_create": (i32, [vp, C.POINTER(vp)]),
        "orbx_kfdb_destroy": (None, [vp]),
        "orbx_kfdb_add": (i32, [vp, C.c_int64, vp, vp, i32]),
        "orbx_kfdb_erase": (i32, [vp, C.c_int64]),
        "orbx_kfdb_clear": (i32, [vp]),
        "orbx_kfdb_size": (i32, [vp]),
        "orbx_kfdb_query": (i32, [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, i32, ip, ip, ip]),
        "orbx_target_search_view": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]),
        "orbx_target_search_view_begin": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32]),
        "orbx_target_search_view_end": (i32, [vp, i32, vp, vp]),
        "orbx_target_search_view_cancel": (i32, [vp, i32]),
        "orbx_last_graph_device_us": (C.c_double, [vp]),
        "orbx_last_window_device_us": (C.c_double, [vp]),
        "orbx_publish_descriptors": (i32, [vp, vp, i32]),
        "orbx_kfdb_sharing": (i32, [vp, vp, i32, vp, vp, i32, ip]),
        "orbx_kfdb_score": (i32, [vp, vp, vp, i32, vp, i32, vp]),
        "orbx_cpu_profile_name": (C.c_char_p, [i32]),
        "orbx_cpu_profile_description": (C.c_char_p, [C.c_char_p]),
        "orbx_cpu_profile_values": (i32, [C.c_char_p, i32, vp]),
        "orbx_set_cpu_profile": (i32, [vp, C.c_char_p, i32]),
        "orbx_get_cpu_profile": (i32, [vp, vp, sz, vp]),
        "orbx_replay_unique_id": (i32, [vp]),
        "orbx_replay_rccl_info": (C.c_char_p, []),
        "orbx_replay_create": (i32, [C.POINTER(vp), vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
        "orbx_replay_prepare": (i32, [C.POINTER(vp), vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
        "orbx_replay_connect": (i32, [vp, vp]),
        "orbx_replay_wait_gathered": (i32, [vp, i32, vp]),
        "orbx_replay_release_gathered": (i32, [vp, i32, vp]),
        "orbx_replay_wait_gathered_host": (i32, [vp, i32, i32]),
        "orbx_replay_failed": (i32, [vp]),
        "orbx_replay_abort": (i32, [vp]),
        "orbx_replay_destroy": (None, [vp]),
        "orbx_replay_last_error": (C.c_char_p, [vp]),
        "orbx_replay_transport": (C.c_char_p, [vp]),
        "orbx_replay_layout": (i32, [vp, ip, ip, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), ip]),
        "orbx_replay_lane_range": (i32, [vp, i32, ip, ip]),
        "orbx_replay_step": (i32, [vp, vp, sz, sz, i32, i32]),
        "orbx_replay_drain": (i32, [vp]),
        "orbx_replay_set_gather": (i32, [vp, i32]),
        "orbx_replay_block": (i32, [vp, i32, C.POINTER(vp)]),
        "orbx_replay_gathered": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "orbx_replay_read": (i32, [vp, i32, i32, vp, sz, sz]),
        "orbx_replay_write_block": (i32, [vp, i32, vp, sz, sz]),
        "orbx_replay_gather_ms": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    L._orbx_symbols = tuple(sig)
    # the diagnostic ABI lives in a library of its own (the product has no debug entry point): its functions are bound onto the same object so
    # that the mirrors' debug helpers read `L.orbx_debug_*`; without the library they raise when called
    dsig = {
        "orbx_debug_blur_level": (i32, [vp, i32, i32, vp, sz]),
        "orbx_debug_level_points": (i32, [vp, i32, i32, i32, vp, i32]),
        "orbx_debug_trig": (i32, [vp, vp, vp, i32, i32, vp, vp, vp]),
        "orbx_debug_trig_hash": (i32, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
        "orbx_debug_atan_hash": (i32, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
        "orbx_debug_brief_hash": (i32, [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
        "orbx_debug_calib_copy": (i32, [vp, vp, vp, sz, i32, vp]),
        "orbx_debug_gnu_sort": (i32, [vp, vp, i32, i32]),
    }
    L._orbx_debug_symbols = tuple(dsig)
    D = C.CDLL(DEBUG_LIB_PATH) if os.path.exists(DEBUG_LIB_PATH) else None
    for name, (res, args) in dsig.items():
        if D is not None:
            fn = getattr(D, name)
            fn.restype = res
            fn.argtypes = args
        else:
            def fn(*_a, _n=name):
                raise ImportError(f"{_n}: {DEBUG_LIB_PATH} not built (the diagnostic ABI is not part of liborbx.so)")
        setattr(L, name, fn)
    L.HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)   # orbx_host_exchange_fn
    _lib = L
    return L


class OrbxGrid(C.Structure):
    """orbx_grid (include/orbx.h): a Frame's / KeyFrame's feature grid as the caller holds it."""
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float),
                ("cell_start", C.c_void_p), ("cell_idx", C.c_void_p)]


def ptr(a) -> C.c_void_p:
    """void* of a numpy array, an int address, or None."""
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


def check(rc: int, ctx=None) -> int:
    if rc < 0:
        msg = ""
        if ctx:
            m = lib().orbx_last_error(ctx)
            msg = m.decode() if m else ""
        raise OrbxError(rc, msg)
    return rc
