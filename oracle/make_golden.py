#!/usr/bin/env python3
"""Generate tests/golden/*.npz -- known-answer vectors for the STFT hot path.

The reference has no golden vectors (it has no tests and no FFT code; SURVEY.md
section 8c), so these are produced HERE, in the build container, by an
implementation that shares nothing with either the C oracle or the HIP kernels:
numpy.fft.rfft over scipy.signal.get_window frames, float64.  They pin the
oracle (tests/test_oracle.py) and the GPU path (tests/test_parity_gpu.py).
The GPU box only reads the .npz files; it never runs this script.

    python oracle/make_golden.py        # rewrites tests/golden/
"""
import os

import numpy as np
import scipy.signal as ss

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def frames_of(x, n, hop):
    v = np.lib.stride_tricks.sliding_window_view(np.asarray(x, np.float64), n)
    return v[::hop]


def power(x, n, hop, window="hann", inv_n=False, nyquist=False):
    w = ss.get_window(window, n, fftbins=True) if window != "rect" else np.ones(n)
    X = np.fft.rfft(frames_of(x, n, hop) * w, axis=1)
    if inv_n:
        X = X / n
    P = X.real ** 2 + X.imag ** 2
    return P if nyquist else P[:, : n // 2]


def sweep(sr=48000, dur=1.0, f0=20.0, f1=20000.0, amp=0.5):
    t = np.arange(int(sr * dur)) / sr
    ph = 2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * dur))
    return (amp * np.sin(ph)).astype(np.float32)


def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print(name, {k: (v.shape, str(v.dtype)) for k, v in kw.items() if hasattr(v, "shape")})


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0x5EED)

    # BASELINE.json configs[0]: 1 s mono 48 kHz sine sweep, N=256, hop=128 -> 374 frames
    x = sweep()
    save("sweep_n256_h128", pcm=x, fft_size=256, hop=128, power=power(x, 256, 128))

    # seeded noise, 8 frames at hop N/4, every size the BASELINE configs use (+ edge sizes)
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        hop = n // 4
        nf = 8 if n <= 4096 else 2
        x = rng.uniform(-1, 1, n + (nf - 1) * hop).astype(np.float32)
        # nyquist = the extra bin N/2 that the N/2+1 output mode appends
        save(f"noise_n{n}", pcm=x, fft_size=n, hop=hop, power=power(x, n, hop),
             nyquist=power(x, n, hop, nyquist=True)[:, -1])

    # odd hops / hop > N / hop == N (generic load path)
    x = rng.uniform(-1, 1, 6000).astype(np.float32)
    for n, hop in ((1024, 100), (1024, 1024), (256, 333), (2048, 1), (1024, 512), (1024, 128)):
        xx = x[: n + 9 * hop] if hop > 1 else x[: n + 40]
        save(f"hop_n{n}_h{hop}", pcm=xx, fft_size=n, hop=hop, power=power(xx, n, hop))

    # dynamic range: full-scale tone + tone at -80 dB, bin-centred and off-centre
    n = 1024
    t = np.arange(n + 7 * 256)
    x = (np.sin(2 * np.pi * 100.0 * t / n) + 1e-4 * np.sin(2 * np.pi * 300.37 * t / n)).astype(np.float32)
    save("two_tone_n1024", pcm=x, fft_size=n, hop=256, power=power(x, n, 256))

    # degenerate inputs: zeros (log floor), +-1 square (clamp), denormal-level
    z = np.zeros(n + 3 * 256, np.float32)
    sq = np.where((np.arange(n + 3 * 256) // 7) % 2 == 0, 1.0, -1.0).astype(np.float32)
    dn = (rng.uniform(-1, 1, n + 3 * 256) * 1e-39).astype(np.float32)
    save("degenerate_n1024", zeros=z, square=sq, denorm=dn, fft_size=n, hop=256,
         power_zeros=power(z, n, 256), power_square=power(sq, n, 256), power_denorm=power(dn, n, 256))

    # Blackman + 1/N (AnalyserNode numerics) and the byte mapping, computed here in float64
    n = 2048  # UI/player.js:10
    x = (0.8 * rng.uniform(-1, 1, n + 5 * 735)).astype(np.float32)  # 735 ~ 44.1 kHz / 60 fps
    x[: n // 2] *= 1e-3
    P = power(x, n, 735, window="blackman", inv_n=True)
    with np.errstate(divide="ignore"):
        Y = 10 * np.log10(P)
    b = np.clip(np.floor(255.0 / 70.0 * (Y + 100.0)), 0, 255).astype(np.uint8)
    # tau = 0.75 smoothing over the same 6 polls (3D/visualizer.js:351)
    mag = np.sqrt(P)
    sm = np.zeros(n // 2)
    bs = []
    for f in range(mag.shape[0]):
        sm = 0.75 * sm + 0.25 * mag[f]
        with np.errstate(divide="ignore"):
            ys = 20 * np.log10(sm)
        bs.append(np.clip(np.floor(255.0 / 70.0 * (ys + 100.0)), 0, 255).astype(np.uint8))
    save("analyser_n2048", pcm=x, fft_size=n, hop=735, power=P, db=Y, bytes_tau0=b,
         bytes_tau075=np.stack(bs))

    # stereo interleaved (sample_stride = 2): L and R are different signals
    n = 2048
    L = rng.uniform(-1, 1, n + 3 * 512).astype(np.float32)
    R = sweep(dur=(n + 3 * 512) / 48000.0 + 1e-9)[: n + 3 * 512]
    inter = np.empty(2 * len(L), np.float32)
    inter[0::2], inter[1::2] = L, R
    save("stereo_n2048", pcm_interleaved=inter, fft_size=n, hop=512,
         power_l=power(L, n, 512), power_r=power(R, n, 512))

    # log-frequency column lookup (sonogram-vertex.shader:51-52 with LINEAR/CLAMP sampling)
    row = rng.uniform(0, 255, 1024)
    cols = 256
    u = (np.arange(cols) + 0.5) / cols
    p = 256.0 ** (u - 1.0) * 1024 - 0.5
    i0 = np.floor(p).astype(int)
    fr = p - i0
    a = row[np.clip(i0, 0, 1023)]
    bb = row[np.clip(i0 + 1, 0, 1023)]
    save("logfreq_1024_to_256", row=row, cols=cols, out=a * (1 - fr) + bb * fr)


if __name__ == "__main__":
    main()
