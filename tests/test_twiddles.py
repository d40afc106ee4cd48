"""CPU: the twiddle triplets the host chooses for the sizes whose kernels derive four of a radix-8 pass's seven twiddles
from three (N >= 4096; spectro_api.hip `derived_triplets`).  Pure host arithmetic behind
sg_debug_twiddle_triplets -- no device.  The kernel's product sequence (pk_math.hpp pk_cmul: d = (fl(a.x w.x), fl(a.y w.x));
d = (fma(-a.y, w.y, d.x), fma(a.x, w.y, d.y))) is restated here in numpy; a float32 fused multiply-add is evaluated in
float64, where the product of two float32 values is exact and the sum is rounded once more only when it needs more than
53 bits -- the comparison below is between error levels a factor of two apart, not between last bits."""
import ctypes as C

import numpy as np
import pytest

from spectrogram_amd import _lib as L


def triplets(n):
    lib = L.load()
    count = lib.sg_debug_twiddle_triplets(n, None, 0)
    out = np.empty(max(count, 1), np.float32)
    assert lib.sg_debug_twiddle_triplets(n, out.ctypes.data, count) == count
    return out[:count]


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)


def cmul(a, w):
    ax, ay = a
    wx, wy = w
    return f32(-ay * wy + f32(ax * wx)), f32(ax * wy + f32(ay * wx))


def seven(b0, b1, b2):
    w3 = cmul(b0, b1)
    return [b0, b1, w3, b2, cmul(b0, b2), cmul(b1, b2), cmul(w3, b2)]


def sq_err(ws, exact):
    return sum((w[0] - e.real) ** 2 + (w[1] - e.imag) ** 2 for w, e in zip(ws, exact))


def test_small_sizes_have_no_table_and_bad_sizes_are_refused():
    lib = L.load()
    for n in (32, 1024, 2048):
        assert lib.sg_debug_twiddle_triplets(n, None, 0) == 0
    assert lib.sg_debug_twiddle_triplets(3000, None, 0) == L.ERR_INVALID
    assert lib.sg_debug_twiddle_triplets(4096, None, 8) == L.ERR_INVALID


@pytest.mark.parametrize("n", [4096, 8192, 16384, 32768])
def test_chosen_triplets_beat_the_nearest_floats(n):
    log2m = n.bit_length() - 2                      # M = N / 2 complex points
    p8 = log2m // 3                                 # radix-8 passes (stft_kernel.hpp, Geo::P8)
    t = triplets(n).astype(np.float64).reshape(-1, 6)
    assert len(t) == sum(8 ** ps for ps in range(1, p8))
    row = 0
    u2 = (2.0 ** -24) ** 2
    for ps in range(1, p8):
        ns = 8 ** ps
        k = np.arange(ns)
        exact = [np.exp(-2j * np.pi * k * r / (8 * ns)) for r in range(1, 8)]
        got = t[row:row + ns]
        row += ns
        # every stored component is the nearest float of the true value or one of its two neighbours
        for c, e in enumerate((exact[0].real, exact[0].imag, exact[1].real, exact[1].imag, exact[3].real, exact[3].imag)):
            near = e.astype(np.float32)
            lo, hi = np.nextafter(near, np.float32(-2)), np.nextafter(near, np.float32(2))
            assert np.all((got[:, c] == near) | (got[:, c] == lo) | (got[:, c] == hi)), (n, ps, c)
        chosen = seven((got[:, 0], got[:, 1]), (got[:, 2], got[:, 3]), (got[:, 4], got[:, 5]))
        nearest = seven(*[(f32(exact[i].real), f32(exact[i].imag)) for i in (0, 1, 3)])
        table = [(f32(e.real), f32(e.imag)) for e in exact]   # what holding all seven table entries would give
        e_chosen, e_near, e_table = (sq_err(w, exact) for w in (chosen, nearest, table))
        # never worse than the nearest floats for any k (they are candidate 0 of the search) ...
        assert np.all(e_chosen <= e_near * (1 + 1e-9) + 1e-30), (n, ps)
        # ... and on average well under their error (measured 0.20-0.21 against 0.35-0.46 per twiddle in units of
        # (2^-24)^2; table entries 0.12-0.15), within 1.8 x of holding all seven table entries
        m_chosen, m_near, m_table = (float(np.mean(e)) / 7 / u2 for e in (e_chosen, e_near, e_table))
        print("N=%d pass %d (%d k): per twiddle, units of (2^-24)^2: table %.2f  nearest-derived %.2f  chosen %.2f"
              % (n, ps, ns, m_table, m_near, m_chosen))
        if ns >= 64:
            assert m_chosen <= 0.62 * m_near and m_chosen <= 1.8 * m_table, (n, ps, m_chosen, m_near, m_table)
    assert row == len(t)
