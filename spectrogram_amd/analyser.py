"""AnalyserNode-shaped object backed by libspectro's poll mode.

Mirrors the members the reference touches on `context.createAnalyser()`:
  fftSize (UI/player.js:10), smoothingTimeConstant (UI/player.js:11,
  3D/visualizer.js:351,357,362), frequencyBinCount (3D/visualizer.js:299,301),
  getByteFrequencyData(Uint8Array) (:352,358), getByteTimeDomainData(Uint8Array)
  (:363), connect() (UI/player.js:25-26), minDecibels / maxDecibels (defaults).
Same names, same argument meaning, same error behaviour (IndexSizeError-like
ValueError for a bad fftSize; short or long caller arrays are filled up to
min(len, bins) and otherwise left untouched).  A native object cannot sit in a
Web Audio graph, so audio is fed with push(samples).  The JS twin of this class
lives in spectrogram_amd/js/analyser.js on top of the N-API shim.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .engine import make_config


class _SampleRows:
    """A [channels, bytes] array whose row length counts SAMPLES for the C ABI (packed int24: three bytes each)."""

    def __init__(self, array, samples_per_row):
        self.ctypes = array.ctypes
        self.shape = (array.shape[0], samples_per_row)
        self._keep = array


class Analyser:
    def __init__(self, fftSize=2048, n_channels=1, window="blackman", device_id=-1, logfreq_cols=0, downmix=0, pcm="f32",
                 logfreq_align="vertex", hip_runtime=None, downmix_mode="mean"):
        """Beyond the platform object: `logfreq_cols` > 0 makes a frequency poll hand out (and the history keep) the
        log-frequency columns the reference's shader samples out of each row (sonogram-vertex.shader:51-52) instead of
        the fftSize/2 bins; `downmix` = K makes push() take K interleaved channels per analyser and average them, as
        the `mix` GainNode does with a stereo source (UI/player.js:25,110-115); `pcm` = "s16" makes push() take
        int16 samples (a 16-bit capture device), converted on the way into the device-resident history;
        `logfreq_align`: "vertex" (default) samples the columns where the mesh of the reference's default display, the
        3-D sonogram, does, u = x / (sonogram3DWidth - 1) (3D/visualizer.js:52,54,213-214); "centre" at pixel centres,
        like its 2-D sonogram's fragment shader (sonogram-fragment.shader:16);
        `hip_runtime`: "auto" | "torch" | "system", as for Engine (only the first load of a process decides);
        `downmix_mode`: "mean" (what the reference's `mix` node does with its mono / stereo sources) or "speakers", the W3C
        mono down-mix an AnalyserNode applies to a multi-channel input (5.1: sqrt(1/2) (L + R) + C + 0.5 (SL + SR))."""
        self._lib = L.load(hip_runtime)
        self._pcm = pcm
        self._logfreq_cols, self._downmix = int(logfreq_cols), int(downmix)
        self._logfreq_align = logfreq_align
        self._downmix_mode = downmix_mode
        self._n_channels = int(n_channels)
        self._window = window
        self._device_id = device_id
        self._tau = 0.8          # W3C default smoothingTimeConstant
        self._min_db, self._max_db = -100.0, -30.0
        self._h = None
        self._plan(fftSize)

    def _plan(self, fft_size):
        fft_size = int(fft_size)
        if fft_size < 32 or fft_size > 32768 or fft_size & (fft_size - 1):
            raise ValueError("IndexSizeError: fftSize must be a power of two in [32, 32768]")
        if self._h:
            # a live analyser keeps its samples (the platform holds the latest 32768 whatever fftSize is); the smoothing
            # state, whose length is fftSize / 2, starts again
            L.check(self._lib.sg_set_fft_size(self._h, fft_size))
            self._fft = fft_size
            return
        cfg = make_config(fft_size=fft_size, hop=fft_size, window=self._window, scale="inv_n",
                          out="mag", min_db=self._min_db, max_db=self._max_db,
                          smoothing=self._tau, device_id=self._device_id, logfreq_cols=self._logfreq_cols,
                          downmix=self._downmix, pcm=self._pcm, logfreq_align=self._logfreq_align, history_samples=32768,
                          downmix_mode=self._downmix_mode)
        h = C.c_void_p()
        L.check(self._lib.sg_create(C.byref(cfg), C.byref(h)))
        L.check(self._lib.sg_stream_begin(h, self._n_channels))
        self._h, self._fft = h, fft_size

    def close(self):
        if self._h:
            self._lib.sg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- attributes of AnalyserNode
    @property
    def fftSize(self):
        return self._fft

    @fftSize.setter
    def fftSize(self, v):  # UI/player.js:10 -- re-plans: smoothing restarts, the sample history stays
        self._plan(v)

    @property
    def frequencyBinCount(self):  # read-only, 3D/visualizer.js:299
        return self._fft // 2

    @property
    def rowLength(self):
        """Values per channel a frequency poll writes: frequencyBinCount, or logfreq_cols when set."""
        return self._logfreq_cols if self._logfreq_cols > 0 else self._fft // 2

    @property
    def smoothingTimeConstant(self):
        return self._tau

    @smoothingTimeConstant.setter
    def smoothingTimeConstant(self, v):  # 3D/visualizer.js:351,357,362
        v = float(v)
        if not 0.0 <= v <= 1.0:
            raise ValueError("IndexSizeError: smoothingTimeConstant must be in [0, 1]")
        L.check(self._lib.sg_set_smoothing(self._h, v))
        self._tau = v

    @property
    def minDecibels(self):
        return self._min_db

    @minDecibels.setter
    def minDecibels(self, v):
        if not float(v) < self._max_db:
            raise ValueError("IndexSizeError: minDecibels must be < maxDecibels")
        L.check(self._lib.sg_set_db_range(self._h, float(v), self._max_db))
        self._min_db = float(v)

    @property
    def maxDecibels(self):
        return self._max_db

    @maxDecibels.setter
    def maxDecibels(self, v):
        if not float(v) > self._min_db:
            raise ValueError("IndexSizeError: maxDecibels must be > minDecibels")
        L.check(self._lib.sg_set_db_range(self._h, self._min_db, float(v)))
        self._max_db = float(v)

    # ---- graph wiring is a no-op for a native object (UI/player.js:25-26,158-167)
    def connect(self, *_a):
        return _a[0] if _a else None

    def disconnect(self, *_a):
        return None

    # ---- feed
    def _rows(self, samples):
        samples = np.asarray(samples)
        if self._pcm == "s16" and samples.dtype != np.int16:
            # a silent cast would turn float samples in [-1, 1) into zeros
            raise TypeError("an analyser made with pcm='s16' takes int16 samples, got %s" % samples.dtype)
        if self._pcm == "s24":
            # packed little-endian 24-bit samples: rows of bytes, three per sample (strides count samples)
            if samples.dtype != np.uint8:
                raise TypeError("an analyser made with pcm='s24' takes uint8 rows of packed 3-byte samples, got %s" % samples.dtype)
            x = np.ascontiguousarray(samples)
            if x.ndim == 1:
                x = x[None, :]
            k = max(1, self._downmix)
            if x.shape[0] != self._n_channels or x.shape[1] % (3 * k):
                raise ValueError("expected %d rows of whole %d-channel groups of 3-byte samples" % (self._n_channels, k))
            return _SampleRows(x, x.shape[1] // 3), x.shape[1] // (3 * k)
        x = np.ascontiguousarray(samples, np.int16 if self._pcm == "s16" else np.float32)
        if x.ndim == 1:
            x = x[None, :]
        if x.shape[0] != self._n_channels:
            raise ValueError("expected %d channels" % self._n_channels)
        k = max(1, self._downmix)
        if x.shape[1] % k:
            raise ValueError("rows must hold a whole number of %d-channel sample groups" % k)
        return x, x.shape[1] // k

    def push(self, samples):
        """float32 [n_new] (mono) or [n_channels, n_new]; with downmix = K the rows hold n_new * K interleaved samples."""
        x, n_new = self._rows(samples)
        L.check(self._lib.sg_push(self._h, x.ctypes.data, n_new, x.shape[1], L.PTR_HOST, None))

    def _push_poll(self, fn, samples, array, dtype):
        x, n_new = self._rows(samples)
        if not isinstance(array, np.ndarray) or array.dtype != dtype:
            raise TypeError("expected a numpy array of dtype %s" % np.dtype(dtype).name)
        a = array if array.ndim == 2 else array[None, :]
        if a.shape[0] != self._n_channels or not a.flags.c_contiguous:
            raise ValueError("expected a C-contiguous [%d, len] array" % self._n_channels)
        L.check(fn(self._h, x.ctypes.data, n_new, x.shape[1], a.ctypes.data, a.shape[1], a.shape[1], L.PTR_HOST, None))

    def pushAndGetByteFrequencyData(self, samples, array):
        """One animation frame of the reference as one call (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358): the new
        audio reaches the analyser, then getByteFrequencyData(array) -- ONE kernel launch and one synchronise when the block is
        an even number of samples up to one window (sg_push_poll_u8); any other block: push, then poll, same results."""
        self._push_poll(self._lib.sg_push_poll_u8, samples, array, np.uint8)

    def pushAndGetFloatFrequencyData(self, samples, array):
        self._push_poll(self._lib.sg_push_poll_f32, samples, array, np.float32)

    # ---- polls: write in place into the caller's array, return None (like the platform)
    def _poll(self, fn, array, dtype):
        if not isinstance(array, np.ndarray) or array.dtype != dtype:
            raise TypeError("expected a numpy array of dtype %s" % np.dtype(dtype).name)
        a = array if array.ndim == 2 else array[None, :]
        if a.shape[0] != self._n_channels or not a.flags.c_contiguous:
            raise ValueError("expected a C-contiguous [%d, len] array" % self._n_channels)
        L.check(fn(self._h, a.ctypes.data, a.shape[1], a.shape[1], L.PTR_HOST, None))

    def getByteFrequencyData(self, array):  # 3D/visualizer.js:352,358
        self._poll(self._lib.sg_poll_u8, array, np.uint8)

    def getFloatFrequencyData(self, array):
        self._poll(self._lib.sg_poll_f32, array, np.float32)

    def getByteTimeDomainData(self, array):  # 3D/visualizer.js:363
        self._poll(self._lib.sg_poll_time_u8, array, np.uint8)

    def getFloatTimeDomainData(self, array):  # the AnalyserNode's fourth getter; the reference reads only the byte form
        self._poll(self._lib.sg_poll_time_f32, array, np.float32)

    # ---- the history texture AnalyserView keeps (3D/visualizer.js:60,293-331,390-417), on device
    def startHistory(self, height=256):
        """Allocate the [channels][height][frequencyBinCount] byte ring; every later
        getByteFrequencyData() also writes its row at `historyOffset` and advances it."""
        L.check(self._lib.sg_ring_begin(self._h, int(height)))
        self._ring_h = int(height)

    def pinHistory(self, pinned=True):
        """pinned: every poll writes row 0 and `historyOffset` stays 0 -- the reference outside its sonogram
        modes (3D/visualizer.js:385-390 sets yoffset = 0 for the 2-D frequency / waveform displays)."""
        L.check(self._lib.sg_ring_set_pinned(self._h, 1 if pinned else 0))

    @property
    def historyOffset(self):  # the reference's `yoffset`
        return int(self._lib.sg_ring_offset(self._h))

    def readHistory(self):
        """The ring unrolled, oldest row first: uint8 [channels, height, frequencyBinCount]."""
        out = np.empty((self._n_channels, self._ring_h, self.rowLength), np.uint8)
        L.check(self._lib.sg_ring_read(self._h, out.ctypes.data, 0, L.PTR_HOST, None))
        return out
