"""Batch STFT engine: the host-side face of libspectro's sg_stft.

Replaces, for a recorded signal, the loop the reference runs at display rate:
spec3D.draw_() -> AnalyserView.doFrequencyAnalysis() -> analyser.getByteFrequencyData()
(UI/spectrogram.js:153-161, 3D/visualizer.js:346-368).  PyTorch appears here
only as the owner of device memory and streams; every number is produced by the
HIP kernels behind the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def make_config(fft_size=1024, hop=256, window="hann", scale="none", out="power", nyquist=False,
                min_db=-100.0, max_db=-30.0, floor_db=-200.0, smoothing=0.0, device_id=-1, logfreq_cols=0, pcm="f32", downmix=0,
                logfreq_align="centre", history_samples=0, downmix_mode="mean"):
    cfg = L.SgConfig()
    L.check(L.load().sg_config_default(C.byref(cfg)))
    cfg.fft_size, cfg.hop = int(fft_size), int(hop)
    cfg.window = {"hann": L.WINDOW_HANN, "blackman": L.WINDOW_BLACKMAN, "rect": L.WINDOW_RECT}[window]
    cfg.scale = {"none": L.SCALE_NONE, "inv_n": L.SCALE_INV_N}[scale]
    cfg.out_kind = {"power": L.OUT_POWER, "mag": L.OUT_MAG, "db": L.OUT_DB, "db_u8": L.OUT_DB_U8}[out]
    cfg.nyquist = int(bool(nyquist))
    cfg.min_db, cfg.max_db, cfg.floor_db = float(min_db), float(max_db), float(floor_db)
    cfg.smoothing = float(smoothing)
    cfg.device_id = int(device_id)
    cfg.logfreq_cols = int(logfreq_cols)
    cfg.downmix = int(downmix)
    cfg.pcm_format = {"f32": L.PCM_F32, "s16": L.PCM_S16, "s24": L.PCM_S24}[pcm]
    cfg.logfreq_align = {"centre": L.LOGFREQ_PIXEL_CENTRE, "vertex": L.LOGFREQ_VERTEX}[logfreq_align]
    cfg.history_samples = int(history_samples)
    cfg.downmix_mode = {"mean": L.DOWNMIX_MEAN, "speakers": L.DOWNMIX_SPEAKERS}[downmix_mode]
    return cfg


# tuning / test switches of a handle (sg_debug_set).  libspectro never reads the environment; for the measurement scripts
# under tools/ this binding translates the switches' old environment names when an engine is made (tests pass `debug=`).
DEBUG_KEYS = ("run_len", "no_lring", "lring_all", "no_single_round", "no_fused_smoothing", "no_pinned_direct", "extra_lds",
              "pinned_direct_generic", "no_lffast", "no_gen_unit")


def debug_from_env():
    import os
    out = {}
    for k in DEBUG_KEYS:
        v = os.environ.get("SG_" + k.upper())
        if v is not None:
            out[k] = int(v) if v.lstrip("-").isdigit() else 1
    return out


def apply_debug(lib, h, debug):
    for k, v in (debug or {}).items():
        if v is not None:
            L.check(lib.sg_debug_set(h, k.encode(), int(v)))


class Engine:
    """One sg_handle.  Not thread-safe (one per stream/thread, like the C ABI)."""

    def __init__(self, debug=None, hip_runtime=None, **kw):
        """debug: {key: value} for sg_debug_set (run geometry / load variant switches; output bits never change).
        hip_runtime: "auto" | "torch" | "system" -- which libamdhip64 the process binds when this is the first thing in it
        that loads libspectro.so (spectrogram_amd/_lib.py states the policy; later objects cannot change it)."""
        self._lib = L.load(hip_runtime)
        self.cfg = make_config(**kw)
        h = C.c_void_p()
        L.check(self._lib.sg_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        apply_debug(self._lib, h, debug_from_env() if debug is None else debug)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.sg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- shape helpers
    @property
    def fft_size(self):
        return self.cfg.fft_size

    def _live(self):
        if not getattr(self, "_h", None):
            raise RuntimeError("Engine is closed")
        return self._h

    @property
    def bins(self):
        return int(self._lib.sg_num_bins(self._live()))

    def num_frames(self, n_samples):
        return int(self._lib.sg_num_frames(self._live(), int(n_samples)))

    def last_launch(self):
        """(load_variant, run_len, single_round, total_runs) of the last transform (sg_last_launch)."""
        info = L.SgLaunchInfo()
        L.check(self._lib.sg_last_launch(self._live(), C.byref(info)))
        return info.load_variant, info.run_len, info.single_round, info.total_runs

    @property
    def out_dtype(self):
        return np.uint8 if self.cfg.out_kind == L.OUT_DB_U8 else np.float32

    # -- batch, host memory (numpy in / numpy out)
    def stft_host(self, pcm, sample_stride=1):
        """pcm: float32 / int16 [channels, samples*stride] (or 1-D), or uint8 [channels, 3*samples*stride] for
        pcm="s24" (packed little-endian 24-bit).  Returns [channels, frames, bins]."""
        fmt = self.cfg.pcm_format
        pcm = np.ascontiguousarray(pcm, {L.PCM_S16: np.int16, L.PCM_S24: np.uint8}.get(fmt, np.float32))
        squeeze = pcm.ndim == 1
        if squeeze:
            pcm = pcm[None, :]
        n_ch, width = pcm.shape
        if fmt == L.PCM_S24:       # packed 24-bit: rows of bytes, three per sample (strides count samples)
            if width % 3:
                raise ValueError("packed int24 rows must hold a whole number of 3-byte samples")
            width //= 3
        k = max(1, self.cfg.downmix)
        n_samples = ((width + sample_stride - 1) // sample_stride) // k   # analysed (down-mixed) samples
        frames = self.num_frames(n_samples)
        out = np.empty((n_ch, frames, self.bins), self.out_dtype)
        L.check(self._lib.sg_stft(self._live(), pcm.ctypes.data, n_ch, n_samples, width, sample_stride,
                                  out.ctypes.data, frames * self.bins, L.PTR_HOST, None))
        return out[0] if squeeze else out

    # -- batch, device memory (raw pointers; torch tensors welcome)
    def stft_device_ptr(self, d_pcm, n_channels, n_samples, channel_stride, sample_stride, d_out,
                        out_channel_stride=0, stream=0):
        """n_samples counts ANALYSED samples: with cfg.downmix = K > 1 the kernel reads
        n_samples * K * sample_stride input elements per channel (include/spectro.h, sg_stft)."""
        L.check(self._lib.sg_stft(self._live(), d_pcm, n_channels, n_samples, channel_stride, sample_stride,
                                  d_out, out_channel_stride, L.PTR_DEVICE, stream or None))

    def stft(self, pcm, out=None, sample_stride=1):
        """torch CUDA tensor [channels, samples*stride] float32 -> [channels, frames, bins] on the
        same device, enqueued on torch's current stream."""
        import torch
        fmt = self.cfg.pcm_format
        want = {L.PCM_S16: torch.int16, L.PCM_S24: torch.uint8}.get(fmt, torch.float32)
        if not (pcm.is_cuda and pcm.dtype == want and pcm.dim() in (1, 2)):
            raise TypeError("pcm must be a CUDA tensor of %s with 1 or 2 dimensions (packed int24: uint8 rows of 3-byte samples)" % want)
        x = pcm if pcm.dim() == 2 else pcm[None, :]
        assert x.stride(1) == 1, "samples must be contiguous within a channel"
        n_ch, width = x.shape
        cs = x.stride(0)
        if fmt == L.PCM_S24:   # rows of bytes, three per sample; strides count samples
            if width % 3 or (n_ch > 1 and cs % 3):
                raise ValueError("packed int24 rows (and the stride between them) must hold whole 3-byte samples")
            width //= 3
            cs //= 3
        k = max(1, self.cfg.downmix)
        n_samples = ((width + sample_stride - 1) // sample_stride) // k   # analysed (down-mixed) samples
        frames = self.num_frames(n_samples)
        dt = torch.uint8 if self.out_dtype == np.uint8 else torch.float32
        if out is None:
            out = torch.empty((n_ch, frames, self.bins), dtype=dt, device=pcm.device)
        assert out.is_contiguous() and out.dtype == dt and out.numel() >= n_ch * frames * self.bins
        stream = torch.cuda.current_stream(pcm.device).cuda_stream
        self.stft_device_ptr(x.data_ptr(), n_ch, n_samples, cs, sample_stride, out.data_ptr(),
                             frames * self.bins, stream)
        return out if pcm.dim() == 2 else out[0]
