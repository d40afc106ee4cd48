"""ctypes binding of oracle/liboracle.so (float64 CPU oracle; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # ORACLE_LIB: sanitizer build
_lib = None

WINDOW_HANN, WINDOW_BLACKMAN, WINDOW_RECT = 0, 1, 2
SCALE_NONE, SCALE_INV_N = 0, 1
OUT_POWER, OUT_MAG, OUT_DB, OUT_DB_U8 = 0, 1, 2, 3


class OracleConfig(C.Structure):
    _fields_ = [("fft_size", C.c_int), ("hop", C.c_int), ("window", C.c_int),
                ("scale", C.c_int), ("out_kind", C.c_int), ("nyquist", C.c_int),
                ("min_db", C.c_double), ("max_db", C.c_double), ("floor_db", C.c_double)]

    @classmethod
    def make(cls, fft_size, hop, window=WINDOW_HANN, scale=SCALE_NONE, out_kind=OUT_POWER,
             nyquist=0, min_db=-100.0, max_db=-30.0, floor_db=-200.0):
        return cls(fft_size, hop, window, scale, out_kind, nyquist, min_db, max_db, floor_db)


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    if os.environ.get("ORACLE_LIB"):
        return _SO
    src = os.path.join(_HERE, "stft_oracle.c")
    hdr = os.path.join(_HERE, "stft_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        f32p, f64p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        L.so_window.argtypes = [C.c_int, C.c_int, f64p]
        L.so_fft_radix2.argtypes = [f64p, f64p, C.c_int]
        L.so_num_frames.argtypes = [C.c_int64, C.c_int, C.c_int]
        L.so_num_frames.restype = C.c_int64
        L.so_num_bins.argtypes = [C.POINTER(OracleConfig)]
        L.so_stft.argtypes = [C.POINTER(OracleConfig), f32p, C.c_int64, C.c_int64, f64p, u8p]
        L.so_stft.restype = C.c_int64
        L.so_stft_f32.argtypes = [C.POINTER(OracleConfig), f32p, C.c_int64, C.c_int64, f32p]
        L.so_stft_f32.restype = C.c_int64
        L.so_stft_f32js.argtypes = [C.POINTER(OracleConfig), f32p, C.c_int64, C.c_int64, f32p]
        L.so_stft_f32js.restype = C.c_int64
        L.so_analyser_poll.argtypes = [C.c_int, f32p, C.c_double, C.c_double, C.c_double,
                                       f64p, f64p, u8p]
        L.so_time_domain_bytes.argtypes = [f32p, C.c_int, u8p]
        L.so_logfreq_resample.argtypes = [f64p, C.c_int, C.c_int, f64p]
        L.so_logfreq_resample_at.argtypes = [f64p, C.c_int, C.c_int, C.c_int, f64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def window(kind, n):
    w = np.empty(n, np.float64)
    if lib().so_window(kind, n, _p(w, C.c_double)):
        raise ValueError("bad window")
    return w


def fft_radix2(z):
    z = np.asarray(z, np.complex128)
    re, im = np.ascontiguousarray(z.real), np.ascontiguousarray(z.imag)
    if lib().so_fft_radix2(_p(re, C.c_double), _p(im, C.c_double), len(z)):
        raise ValueError("length must be a power of two")
    return re + 1j * im


def num_frames(n_samples, fft_size, hop):
    return int(lib().so_num_frames(n_samples, fft_size, hop))


def stft(cfg, pcm, sample_stride=1):
    """One channel. Returns (frames x bins float64, frames x bins uint8 or None)."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    n_samples = (len(pcm) + sample_stride - 1) // sample_stride
    frames = num_frames(n_samples, cfg.fft_size, cfg.hop)
    bins = lib().so_num_bins(C.byref(cfg))
    out = np.empty((frames, bins), np.float64)
    u8 = np.empty((frames, bins), np.uint8) if cfg.out_kind == OUT_DB_U8 else None
    r = lib().so_stft(C.byref(cfg), _p(pcm, C.c_float), n_samples, sample_stride,
                      _p(out, C.c_double), _p(u8, C.c_uint8) if u8 is not None else None)
    if r < 0:
        raise ValueError("so_stft rejected the configuration")
    return out, u8


def stft_f32(cfg, pcm, sample_stride=1, out=None):
    pcm = np.ascontiguousarray(pcm, np.float32)
    n_samples = (len(pcm) + sample_stride - 1) // sample_stride
    frames = num_frames(n_samples, cfg.fft_size, cfg.hop)
    bins = lib().so_num_bins(C.byref(cfg))
    if out is None:
        out = np.empty((frames, bins), np.float32)
    r = lib().so_stft_f32(C.byref(cfg), _p(pcm, C.c_float), n_samples, sample_stride,
                          _p(out, C.c_float))
    if r < 0:
        raise ValueError("so_stft_f32 rejected the configuration")
    return out


def stft_f32js(cfg, pcm, out=None):
    """oracle/stft_oracle.js on Float32Array, restated in C (float32 storage, double expressions rounded at each
    store): north_star's "float32 radix-2 JS FFT", the yardstick of the stated tolerance."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    frames = num_frames(len(pcm), cfg.fft_size, cfg.hop)
    if out is None:
        out = np.empty((frames, cfg.fft_size // 2), np.float32)
    r = lib().so_stft_f32js(C.byref(cfg), _p(pcm, C.c_float), len(pcm), 1, _p(out, C.c_float))
    if r < 0:
        raise ValueError("so_stft_f32js rejected the configuration")
    return out


def analyser_poll(fft_size, last_n, tau, min_db, max_db, state):
    """Returns (dB float64[bins], bytes uint8[bins]); updates `state` in place."""
    last_n = np.ascontiguousarray(last_n, np.float32)
    assert len(last_n) == fft_size and state.dtype == np.float64 and len(state) == fft_size // 2
    db = np.empty(fft_size // 2, np.float64)
    by = np.empty(fft_size // 2, np.uint8)
    if lib().so_analyser_poll(fft_size, _p(last_n, C.c_float), tau, min_db, max_db,
                              _p(state, C.c_double), _p(db, C.c_double), _p(by, C.c_uint8)):
        raise ValueError("so_analyser_poll rejected the arguments")
    return db, by


def time_domain_bytes(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(len(x), np.uint8)
    lib().so_time_domain_bytes(_p(x, C.c_float), len(x), _p(out, C.c_uint8))
    return out


def logfreq_resample(row, n_cols, align="centre"):
    """align: "centre" (pixel centres, the fragment shader) or "vertex" (the 3-D sonogram's mesh, u = c / (cols - 1))."""
    row = np.ascontiguousarray(row, np.float64)
    out = np.empty(n_cols, np.float64)
    if lib().so_logfreq_resample_at(_p(row, C.c_double), len(row), n_cols, {"centre": 0, "vertex": 1}[align], _p(out, C.c_double)):
        raise ValueError("so_logfreq_resample_at rejected the arguments")
    return out
