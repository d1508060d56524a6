"""ctypes binding of include/dysf_hip.h.  There is no CPU fallback: if libdysf_hip.so is missing or
fails to load, every entry point raises (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYSF_LIB") or os.path.join(_HERE, "libdysf_hip.so")  # DYSF_LIB: diagnostic builds

DYSF_OK, DYSF_TOO_SHORT, DYSF_NONFINITE, DYSF_EMPTY = 0, 1, 2, 3
STATUS_NAMES = {0: "OK", 1: "TOO_SHORT", 2: "NONFINITE", 3: "EMPTY", 4: "TOO_LONG"}
PCM_I16, PCM_F32 = 0, 1
CNN_F16, CNN_F32 = 0, 1
ABI_VERSION = 2   # include/dysf_hip.h DYSF_ABI_VERSION
PRESET_REFERENCE_149, PRESET_BENCH_512 = 0, 1

#: every symbol include/dysf_hip.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "dysf_abi_version", "dysf_last_error", "dysf_params_preset", "dysf_create", "dysf_destroy",
    "dysf_pooled_dim", "dysf_frame_rows", "dysf_num_frames", "dysf_debug_table",
    "dysf_extract_batch_dev", "dysf_extract_batch", "dysf_debug_stft_power_dev", "dysf_debug_logmel_dev",
    "dysf_debug_timing", "dysf_debug_peaks", "dysf_metrics_batch_dev",
    "dysf_forest_last_error", "dysf_forest_create", "dysf_forest_destroy", "dysf_forest_predict_dev",
    "dysf_svc_last_error", "dysf_svc_create", "dysf_svc_destroy", "dysf_svc_predict_dev",
    "dysf_mlp_last_error", "dysf_mlp_create", "dysf_mlp_destroy", "dysf_mlp_forward_dev",
    "dysf_resampler_last_error", "dysf_resampler_create", "dysf_resampler_destroy", "dysf_resampler_out_len",
    "dysf_resample_batch_dev",
    "dysf_cnn_last_error", "dysf_cnn_create", "dysf_cnn_destroy", "dysf_cnn_forward_dev",
    "dysf_clean_last_error", "dysf_clean_create", "dysf_clean_destroy", "dysf_clean_batch_dev",
)


class DysfParams(ctypes.Structure):
    _fields_ = [
        ("sr", ctypes.c_int32), ("n_fft", ctypes.c_int32), ("hop", ctypes.c_int32),
        ("n_mels", ctypes.c_int32), ("n_mfcc", ctypes.c_int32),
        ("fmin", ctypes.c_float), ("fmax", ctypes.c_float), ("top_db", ctypes.c_float), ("amin", ctypes.c_float),
        ("delta_width", ctypes.c_int32), ("chroma", ctypes.c_int32), ("rich", ctypes.c_int32),
        ("parity", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3),
    ]


class DysfError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libdysf_hip.so once; raise loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DysfError(f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback. "
                        "Run `make -C recognizing-speech-dysfluencies-in-stuttering_amd/csrc` or __graft_entry__.build().")
    try:
        # One HIP runtime per process: torch wheels bundle their own libamdhip64, and device pointers / streams
        # are only shareable when this library binds to the same instance.  Loading torch first makes the
        # dynamic loader resolve our libamdhip64.so.7 dependency to the copy torch has already mapped.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.dysf_abi_version.restype = i32
    L.dysf_last_error.restype = ctypes.c_char_p
    L.dysf_params_preset.argtypes = [i32, ctypes.POINTER(DysfParams)]
    L.dysf_create.argtypes = [ctypes.POINTER(DysfParams), i32, ctypes.POINTER(vp)]
    L.dysf_destroy.argtypes = [vp]
    L.dysf_destroy.restype = None
    L.dysf_pooled_dim.argtypes = [vp]
    L.dysf_frame_rows.argtypes = [vp]
    L.dysf_num_frames.argtypes = [vp, i64]
    L.dysf_num_frames.restype = i64
    L.dysf_debug_table.argtypes = [vp, i32, i32, vp, i64]
    L.dysf_debug_table.restype = i64
    L.dysf_extract_batch_dev.argtypes = [vp, vp, i32, vp, i32, i64, vp, vp, i64, vp, vp, vp]
    L.dysf_extract_batch.argtypes = [vp, vp, i32, vp, i32, vp, vp, i64, vp, vp]
    L.dysf_metrics_batch_dev.argtypes = [vp, vp, i32, vp, i32, i64, vp, vp]
    L.dysf_debug_stft_power_dev.argtypes = [vp, vp, i32, i64, vp, vp]
    L.dysf_debug_logmel_dev.argtypes = [vp, vp, i32, i64, vp, vp]
    L.dysf_debug_timing.argtypes = [vp, i32, vp]
    L.dysf_debug_peaks.argtypes = [vp, vp, vp, i64]
    L.dysf_debug_peaks.restype = i64
    L.dysf_forest_last_error.restype = ctypes.c_char_p
    L.dysf_forest_create.argtypes = [i32, i32, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.dysf_forest_destroy.argtypes = [vp]
    L.dysf_forest_destroy.restype = None
    L.dysf_forest_predict_dev.argtypes = [vp, vp, i32, vp, vp, vp]
    L.dysf_svc_last_error.restype = ctypes.c_char_p
    L.dysf_svc_create.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, ctypes.c_double, vp, vp, ctypes.POINTER(vp)]
    L.dysf_svc_destroy.argtypes = [vp]
    L.dysf_svc_destroy.restype = None
    L.dysf_svc_predict_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.dysf_mlp_last_error.restype = ctypes.c_char_p
    L.dysf_mlp_create.argtypes = [i32, i32, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.dysf_mlp_destroy.argtypes = [vp]
    L.dysf_mlp_destroy.restype = None
    L.dysf_mlp_forward_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.dysf_resampler_last_error.restype = ctypes.c_char_p
    L.dysf_resampler_create.argtypes = [i32, i32, i32, vp, i32, i32, ctypes.POINTER(vp)]
    L.dysf_resampler_destroy.argtypes = [vp]
    L.dysf_resampler_destroy.restype = None
    L.dysf_resampler_out_len.argtypes = [vp, i64]
    L.dysf_resampler_out_len.restype = i64
    L.dysf_resample_batch_dev.argtypes = [vp, vp, i32, vp, i32, vp, vp, i64, vp]
    L.dysf_cnn_last_error.restype = ctypes.c_char_p
    L.dysf_cnn_create.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(vp)]
    L.dysf_cnn_destroy.argtypes = [vp]
    L.dysf_cnn_destroy.restype = None
    L.dysf_cnn_forward_dev.argtypes = [vp, vp, i32, i32, i32, vp, i64, vp, vp]
    f64 = ctypes.c_double
    L.dysf_clean_last_error.restype = ctypes.c_char_p
    L.dysf_clean_create.argtypes = [i32, i32, f64, f64, f64, f64, f64, f64, i32, ctypes.POINTER(vp)]
    L.dysf_clean_destroy.argtypes = [vp]
    L.dysf_clean_destroy.restype = None
    L.dysf_clean_batch_dev.argtypes = [vp, vp, i32, vp, i32, i64, i64, vp, vp, vp, vp, vp]
    for name in SYMBOLS:
        getattr(L, name)
    if L.dysf_abi_version() != ABI_VERSION:
        raise DysfError("libdysf_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int):
    if rc < 0:
        raise DysfError(f"dysf error {rc}: {lib().dysf_last_error().decode(errors='replace')}")
    return rc
