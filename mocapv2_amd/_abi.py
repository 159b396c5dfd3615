"""ctypes binding of libmocap_hip.so (include/mocap_hip.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmocap_hip.so")
ABI_VERSION = 5
COMM_ID_BYTES = 128  # MOCAP_COMM_ID_BYTES


class MocapError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmocap_hip error {code}: {msg}")
        self.code = code


class BlobParams(C.Structure):
    _fields_ = [("ksize", C.c_int32), ("median", C.c_int32), ("thresh", C.c_double), ("min_area", C.c_double),
                ("min_circ", C.c_double)]


class UndistortInfo(C.Structure):
    _fields_ = [("identity", C.c_int32), ("compact_table", C.c_int32), ("early_out_provable", C.c_int32),
                ("max_source_weight", C.c_int32), ("sparse_path", C.c_int32)]


class Contour(C.Structure):
    _fields_ = [("key", C.c_int32), ("is_hole", C.c_int32), ("sx", C.c_int32), ("sy", C.c_int32), ("npts", C.c_int32),
                ("steps", C.c_int32), ("a00", C.c_int64), ("a10", C.c_int64), ("a01", C.c_int64), ("area", C.c_double),
                ("perimeter", C.c_double), ("kept", C.c_int32), ("cx", C.c_int32), ("cy", C.c_int32), ("link", C.c_int32),
                ("parent", C.c_int32), ("order", C.c_int32)]


_vp, _i, _d, _sz, _l = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_long
_dp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_int)

# name -> argtypes; every symbol include/mocap_hip.h declares (restype int unless noted)
SIGNATURES = {
    "mocap_abi_version": [],
    "mocap_last_error": [],
    "mocap_ctx_create": [_i, _i, _i, _i, C.POINTER(_vp)],
    "mocap_ctx_destroy": [_vp],
    "mocap_sync": [_vp, _vp],
    "mocap_set_blob_params": [_vp, C.POINTER(BlobParams)],
    "mocap_set_tuning": [_vp, C.c_char_p, _i],
    "mocap_set_undistort": [_vp, _i, _dp, _dp, _ip],
    "mocap_undistort_info": [_vp, _i, C.POINTER(UndistortInfo)],
    "mocap_set_cameras": [_vp, _i, _dp, _dp, _dp, _dp],
    "mocap_set_fundamentals": [_vp, _i, _dp],
    "mocap_blob_centroids": [_vp, _vp, _i, _i, _i, _sz, _i, _vp, _l, _vp, _l, _i, _vp],
    "mocap_blob_centroids_bayer": [_vp, _vp, _vp, _i, _i, _i, _sz, _i, _i, _i, _vp, _l, _vp, _l, _i, _vp],
    "mocap_filter_mask": [_vp, _vp, _i, _i, _i, _sz, _i, _vp, _vp],
    "mocap_contours_from_mask": [_vp, _vp, _i, _vp, _l, _vp, _l, _i, _vp, _vp, _i, _vp],
    "mocap_image_filter_u8": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mocap_undistort_u8": [_vp, _i, _vp, _vp, _i, _i, _vp],
    "mocap_box_blur_u8": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mocap_demosaic_u8": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mocap_bayer_gray_u8": [_vp, _vp, _vp, _i, _i, _i, _l, _l, _sz, _sz, _i, _i, _vp],
    "mocap_correspond": [_vp, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mocap_epipolar_scores": [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp],
    "mocap_ba_residuals": [_vp, _dp, _i, _vp, _vp, _i, _i, C.POINTER(C.c_float), _ip, _vp],
    "mocap_triangulate_batch": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "mocap_reproject_batch": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "mocap_comm_unique_id": [_vp],
    "mocap_comm_available": [],
    "mocap_comm_init": [_vp, _vp, _i, _i],
    "mocap_comm_share": [_vp, _vp],
    "mocap_comm_destroy": [_vp],
    "mocap_allgather_centroids": [_vp, _vp, _vp, _l, _vp],
    "mocap_tile_stats": [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "mocap_profile_enable": [_vp, _i],
    "mocap_profile_read": [_vp, _dp, _ip],
}

_lib = None


def load():
    """Load the HIP library; raises if it was not built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C mocapv2_amd/csrc` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # PyTorch (the device-buffer provider of the host side) ships its own libamdhip64; a process must initialise the GPU through
    # ONE copy of the runtime.  Loading this library first would bind /opt/rocm's copy and leave it without a device once torch
    # has initialised its own ("no ROCm-capable device is detected"): torch goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = library does not match the header
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "mocap_last_error" else C.c_int
    if lib.mocap_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libmocap_hip ABI {lib.mocap_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().mocap_last_error()
        raise MocapError(rc, msg.decode() if msg else "")
