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
DOWNMIX_MEAN, DOWNMIX_SPEAKERS = 0, 1
PTR_HOST, IN_DEVICE, OUT_DEVICE, PTR_DEVICE, IN_PINNED, OUT_PINNED = 0, 1, 2, 3, 4, 8
SG_OK, ERR_INVALID, ERR_NOMEM, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4, -5


class SgConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("fft_size", C.c_int32), ("hop", C.c_int32),
                ("window", C.c_int32), ("scale", C.c_int32), ("out_kind", C.c_int32),
                ("nyquist", C.c_int32), ("min_db", C.c_float), ("max_db", C.c_float),
                ("floor_db", C.c_float), ("smoothing", C.c_float), ("device_id", C.c_int32),
                ("pcm_format", C.c_int32), ("downmix", C.c_int32), ("logfreq_cols", C.c_int32),
                ("logfreq_align", C.c_int32), ("history_samples", C.c_int32), ("downmix_mode", C.c_int32)]


class SgWavInfo(C.Structure):
    _fields_ = [("pcm_format", C.c_int32), ("channels", C.c_int32), ("sample_rate", C.c_int32),
                ("bits_per_sample", C.c_int32), ("block_align", C.c_int32), ("data_offset", C.c_int64),
                ("data_bytes", C.c_int64), ("n_frames", C.c_int64)]


class SgLaunchInfo(C.Structure):
    _fields_ = [("load_variant", C.c_int32), ("run_len", C.c_int32), ("single_round", C.c_int32),
                ("reserved", C.c_int32), ("total_runs", C.c_int64)]


class SgRuntime(C.Structure):
    _fields_ = [("runtime_version", C.c_int32), ("driver_version", C.c_int32), ("built_hip_version", C.c_int32),
                ("reserved", C.c_int32), ("library_path", C.c_char * 512)]


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
    "sg_runtime_info": (C.c_int, [C.POINTER(SgRuntime)]),
}

_lib = None


def elf_dynamic(path):
    """(SONAME or None, [NEEDED...]) of a 64-bit little-endian ELF shared object, read from its .dynamic section -- enough to
    tell whether one libamdhip64 can stand in for another without asking an external tool."""
    import struct
    with open(path, "rb") as f:
        d = f.read()
    if d[:6] != b"\x7fELF\x02\x01":
        raise ValueError("%s: not a 64-bit little-endian ELF file" % path)
    shoff, = struct.unpack_from("<Q", d, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", d, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", d, shoff + i * shentsize) for i in range(shnum)]
    soname, needed = None, []
    for sec in secs:
        if sec[1] != 6:                       # SHT_DYNAMIC
            continue
        stroff = secs[sec[6]][4]              # sh_link: its string table
        for o in range(sec[4], sec[4] + sec[5], 16):
            tag, val = struct.unpack_from("<qQ", d, o)
            if tag == 0:
                break
            if tag in (1, 14):                # DT_NEEDED, DT_SONAME
                name = d[stroff + val:d.index(b"\0", stroff + val)].decode()
                if tag == 1:
                    needed.append(name)
                else:
                    soname = name
    return soname, needed


# Which HIP runtime serves the process (a stated policy instead of silent sniffing).
#   libspectro.so names /opt/rocm's libamdhip64 by SONAME (DT_NEEDED libamdhip64.so.7, RUNPATH /opt/rocm/lib); a PyTorch-ROCm
#   wheel ships its own copy under torch/lib.  Whichever is loaded FIRST serves the whole process, and torch on top of the
#   system's runtime reports no device (round 5: an analyser created before `import torch` left torch.cuda.is_available()
#   False for the rest of the process).  Policy, chosen at the FIRST load() of a process (argument, else the
#   SPECTRO_HIP_RUNTIME environment variable -- read here in the binding, never by the library -- else "auto"):
#     "auto"    torch imported already: nothing to do, the process is on torch's runtime.  torch installed but not imported:
#               its libamdhip64 is loaded first (by path, RTLD_GLOBAL, without importing torch) PROVIDED its SONAME is the
#               one libspectro.so needs -- otherwise it is left alone with a warning (a runtime of another major version
#               cannot stand in).  No torch: the system's runtime.
#     "torch"   as auto, but a missing / unsuitable torch runtime is an error.
#     "system"  never preload: libspectro.so binds /opt/rocm's runtime (a later `import torch` in the same process then
#               finds no device -- the caller's choice).
#   What was done is kept in `runtime_policy()`, and `runtime_info()` asks the library itself which libamdhip64 it is bound
#   to (sg_runtime_info: dladdr on the runtime's own entry point) -- bench.py prints both under `host`.
_runtime = {"policy": None, "preloaded": None, "note": None}


def _needed_hip_soname(lib_path):
    for n in elf_dynamic(lib_path)[1]:
        if n.startswith("libamdhip64"):
            return n
    return None


def _apply_runtime_policy(policy, lib_path):
    import sys
    import warnings
    if policy not in ("auto", "torch", "system"):
        raise ValueError("hip_runtime must be 'auto', 'torch' or 'system', got %r" % (policy,))
    _runtime["policy"] = policy
    if policy == "system":
        _runtime["note"] = "system runtime by request"
        return
    if "torch" in sys.modules:
        _runtime["note"] = "torch was imported first: the process is on its runtime already"
        return
    cand = None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.origin:
            c = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            cand = c if os.path.exists(c) else None
    except Exception as e:       # a broken torch install must not keep the engine from loading against the system runtime
        _runtime["note"] = "torch could not be located (%s)" % type(e).__name__
    if cand is None:
        if policy == "torch":
            raise RuntimeError("hip_runtime='torch': no torch/lib/libamdhip64.so in this environment")
        _runtime["note"] = _runtime["note"] or "no torch in this environment: system runtime"
        return
    try:
        want, have = _needed_hip_soname(lib_path), elf_dynamic(cand)[0]
    except Exception as e:
        want, have = None, "unreadable (%s)" % type(e).__name__
    if want is None or have != want:
        msg = "torch's %s has SONAME %s but %s needs %s: not preloaded, libspectro binds the system runtime" % (cand, have, os.path.basename(lib_path), want)
        if policy == "torch":
            raise RuntimeError("hip_runtime='torch': " + msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        _runtime["note"] = msg
        return
    C.CDLL(cand, mode=C.RTLD_GLOBAL)
    _runtime["preloaded"] = cand
    _runtime["note"] = "torch installed but not imported: its runtime loaded first (SONAME %s matches)" % have


def runtime_policy():
    """{"policy", "preloaded", "note"}: what load() decided about the HIP runtime in this process."""
    return dict(_runtime)


def load(hip_runtime=None):
    """dlopen libspectro.so (building it first if absent) and type every entry point.  hip_runtime: "auto" | "torch" |
    "system" (see above); only the first load of a process decides."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SPECTRO_LIB", LIB_PATH)  # override: diagnostic builds only
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    _apply_runtime_policy(hip_runtime or os.environ.get("SPECTRO_HIP_RUNTIME") or "auto", path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def runtime_info():
    """Which HIP runtime libspectro.so is bound to, from the library itself (sg_runtime_info), plus this binding's policy."""
    info = SgRuntime()
    check(load().sg_runtime_info(C.byref(info)))
    return {"library_path": info.library_path.decode("utf-8", "replace"), "runtime_version": info.runtime_version,
            "driver_version": info.driver_version, "built_with_hip_version": info.built_hip_version,
            "policy": _runtime["policy"], "preloaded": _runtime["preloaded"], "note": _runtime["note"]}


def check(rc):
    if rc != SG_OK:
        raise SpectroError(rc, load().sg_last_error().decode("utf-8", "replace"))
    return rc
