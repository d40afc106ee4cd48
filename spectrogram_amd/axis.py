"""Frequency axes beside the log-frequency epilogue: the reference's legend scale and its shaders' column map.

`LogScale` mirrors `Util.setLogScale / lin2log / log2lin` (util/util.js:32-43; the page configures
`setLogScale(20, 20, 20000, 20000)`, UI/spectrogram.js:77) -- same names, same argument meaning; the numbers come
from libspectro's sg_axis_* (plain C) and are tested against the reference's own code run under node.
"""
import ctypes as C

from . import _lib as L


class LogScale:
    def __init__(self, x1=20.0, y1=20.0, x2=20000.0, y2=20000.0):
        self.setLogScale(x1, y1, x2, y2)

    def setLogScale(self, x1, y1, x2, y2):
        self._s = L.SgLogScale()
        L.check(L.load().sg_axis_log_scale(float(x1), float(y1), float(x2), float(y2), C.byref(self._s)))
        self.a, self.b = self._s.a, self._s.b

    def lin2log(self, x):
        return L.load().sg_axis_lin2log(C.byref(self._s), float(x))

    def log2lin(self, y):
        return L.load().sg_axis_log2lin(C.byref(self._s), float(y))


def bin_hz(fft_size, k, sample_rate):
    return L.load().sg_bin_hz(int(fft_size), float(k), float(sample_rate))


ALIGN = {"centre": L.LOGFREQ_PIXEL_CENTRE, "vertex": L.LOGFREQ_VERTEX}


def logfreq_col_bin(fft_size, cols, col, align="centre"):
    """Where column `col` samples the N/2-bin row (fractional bin, before CLAMP_TO_EDGE): at pixel centres
    u = (col + 0.5) / cols, or -- "vertex" -- at the reference's 3-D sonogram mesh u = col / (cols - 1)
    (3D/visualizer.js:213-214)."""
    return L.load().sg_logfreq_col_bin(int(fft_size), int(cols), float(col), ALIGN[align])


def logfreq_col_hz(fft_size, cols, col, sample_rate, align="centre"):
    return L.load().sg_logfreq_col_hz(int(fft_size), int(cols), float(col), float(sample_rate), ALIGN[align])
