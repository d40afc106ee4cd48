"""ctypes binding of libspectro.so -- the C ABI in include/spectro.h, nothing else.

The product path has no CPU fallback: if the library is missing it is built
with hipcc, and if that fails (or there is no HIP device at call time) the
error propagates.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libspectro.so")

# enum mirrors of include/spectro.h
WINDOW_HANN, WINDOW_BLACKMAN, WINDOW_RECT = 0, 1, 2
SCALE_NONE, SCALE_INV_N = 0, 1
OUT_POWER, OUT_MAG, OUT_DB, OUT_DB_U8 = 0, 1, 2, 3
PCM_F32, PCM_S16, PCM_S24 = 0, 1, 2
LOGFREQ_PIXEL_CENTRE, LOGFREQ_VERTEX = 0, 1
PTR_HOST, IN_DEVICE, OUT_DEVICE, PTR_DEVICE, IN_PINNED, OUT_PINNED = 0, 1, 2, 3, 4, 8
SG_OK, ERR_INVALID, ERR_NOMEM, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4, -5


class SgConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("fft_size", C.c_int32), ("hop", C.c_int32),
                ("window", C.c_int32), ("scale", C.c_int32), ("out_kind", C.c_int32),
                ("nyquist", C.c_int32), ("min_db", C.c_float), ("max_db", C.c_float),
                ("floor_db", C.c_float), ("smoothing", C.c_float), ("device_id", C.c_int32),
                ("pcm_format", C.c_int32), ("downmix", C.c_int32), ("logfreq_cols", C.c_int32),
                ("logfreq_align", C.c_int32), ("history_samples", C.c_int32)]


class SgWavInfo(C.Structure):
    _fields_ = [("pcm_format", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_int32),
                ("bits_per_sample", C.c_int32), ("block_align", C.c_int32), ("data_offset", C.c_int64),
                ("data_bytes", C.c_int64), ("n_frames", C.c_int64)]


class SgLaunchInfo(C.Structure):
    _fields_ = [("load_variant", C.c_int32), ("run_len", C.c_int32), ("single_round", C.c_int32),
                ("reserved", C.c_int32), ("total_runs", C.c_int64)]


class SgLogScale(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double)]


class SpectroError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libspectro error %d: %s" % (code, msg))
        self.code = code


# every symbol include/spectro.h declares: name -> (restype, argtypes)
_vp, _i64, _i32, _u32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_float
SYMBOLS = {
    "sg_config_default": (C.c_int, [C.POINTER(SgConfig)]),
    "sg_create": (C.c_int, [C.POINTER(SgConfig), C.POINTER(_vp)]),
    "sg_destroy": (C.c_int, [_vp]),
    "sg_last_error": (C.c_char_p, []),
    "sg_version": (C.c_int, []),
    "sg_num_frames": (_i64, [_vp, _i64]),
    "sg_num_bins": (_i32, [_vp]),
    "sg_out_elem_size": (_i32, [_vp]),
    "sg_get_config": (C.c_int, [_vp, C.POINTER(SgConfig)]),
    "sg_stft": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _u32, _vp]),
    "sg_stream_begin": (C.c_int, [_vp, _i32]),
    "sg_push": (C.c_int, [_vp, _vp, _i64, _i64, _u32, _vp]),
    "sg_synchronize": (C.c_int, [_vp, _vp]),
    "sg_poll_u8": (C.c_int, [_vp, _vp, _i64, _i64, _u32, _vp]),
    "sg_poll_f32": (C.c_int, [_vp, _vp, _i64, _i64, _u32, _vp]),
    "sg_push_poll_u8": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _u32, _vp]),
    "sg_push_poll_f32": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _u32, _vp]),
    "sg_poll_time_u8": (C.c_int, [_vp, _vp, _i64, _i64, _u32, _vp]),
    "sg_poll_time_f32": (C.c_int, [_vp, _vp, _i64, _i64, _u32, _vp]),
    "sg_ring_begin": (C.c_int, [_vp, _i32]),
    "sg_ring_set_pinned": (C.c_int, [_vp, _i32]),
    "sg_ring_bytes": (_i64, [_vp]),
    "sg_ring_offset": (_i32, [_vp]),
    "sg_alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(_vp)]),
    "sg_free_pinned": (C.c_int, [_vp]),
    "sg_wav_parse": (C.c_int, [_vp, C.c_size_t, C.POINTER(SgWavInfo)]),
    "sg_axis_log_scale": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(SgLogScale)]),
    "sg_axis_lin2log": (C.c_double, [C.POINTER(SgLogScale), C.c_double]),
    "sg_axis_log2lin": (C.c_double, [C.POINTER(SgLogScale), C.c_double]),
    "sg_bin_hz": (C.c_double, [_i32, C.c_double, C.c_double]),
    "sg_logfreq_col_bin": (C.c_double, [_i32, _i32, C.c_double, _i32]),
    "sg_logfreq_col_hz": (C.c_double, [_i32, _i32, C.c_double, C.c_double, _i32]),
    "sg_ring_read": (C.c_int, [_vp, _vp, _i64, _u32, _vp]),
    "sg_last_launch": (C.c_int, [_vp, C.POINTER(SgLaunchInfo)]),
    "sg_set_fft_size": (C.c_int, [_vp, _i32]),
    "sg_debug_set": (C.c_int, [_vp, C.c_char_p, _i64]),
    "sg_debug_twiddle_triplets": (_i64, [C.c_int32, _vp, _i64]),
    "sg_set_smoothing": (C.c_int, [_vp, _f]),
    "sg_set_db_range": (C.c_int, [_vp, _f, _f]),
}

_lib = None


def _prefer_torchs_hip_runtime():
    """A PyTorch-ROCm wheel ships its own libamdhip64; libspectro.so names /opt/rocm's by SONAME.  Whichever is loaded FIRST serves
    the whole process, and torch on top of the system runtime reports no device (found in round 5: a test that created an
    analyser before anything had imported torch left `torch.cuda.is_available()` False for the rest of the process).  So when
    torch is installed but not imported yet, its runtime is loaded first -- by path, without importing torch -- and both
    libspectro.so and a later `import torch` bind to it; with torch imported already, or not installed, nothing is done."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # a broken torch install must not keep the engine from loading against the system runtime


def load():
    """dlopen libspectro.so (building it first if absent) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SPECTRO_LIB", LIB_PATH)  # override: diagnostic builds only
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    _prefer_torchs_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != SG_OK:
        raise SpectroError(rc, load().sg_last_error().decode("utf-8", "replace"))
    return rc
