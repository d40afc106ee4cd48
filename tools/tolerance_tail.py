#!/usr/bin/env python3
"""How far does float32 FFT noise reach into the stated tolerance -- and how does the HIP kernel compare with the float32
radix-2 FFT north_star names as the reference?  ~1e8 bins per FFT size; for the HIP output and for the two float32
radix-2 twins (oracle/yardstick.py: "twin_js" = oracle/stft_oracle.js on Float32Array, bit for bit; "twin_f32" = the same
algorithm with every operation in float32), ON THE SAME FRAMES, against the float64 oracle:
    F=1e-5 / F=2e-5   max |dP| / max(P, F max_k P)      (the tolerance statement's left side at two floors)
    peak max / rms    max and rms of |dP| / max_k P     (float32 FFT noise scales with the frame's peak)
Usage: tolerance_tail.py [bins_per_case] [fft sizes ...]  ->  profiles/r04_tolerance_tail.txt"""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import spectrogram_amd as S
from oracle import yardstick

bins_per_case = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
workers = min(16, len(os.sched_getaffinity(0)))
print("# tools/tolerance_tail.py: HIP kernel and the float32 radix-2 twins against the float64 oracle, same frames, %.0e bins per case" % bins_per_case)
for n in ([int(a) for a in sys.argv[2:]] or (256, 512, 1024, 2048, 4096, 8192)):
    hop = n // 2 + 5
    frames = int(bins_per_case) // (n // 2) + 1
    ns = n + (frames - 1) * hop
    for window, kind in (("hann", "uniform"), ("blackman", "int16")):
        rng = np.random.default_rng(n * 7 + len(window) + 1000)
        if kind == "uniform":
            x = rng.uniform(-1, 1, ns).astype(np.float32)
            dev, pcm = torch.from_numpy(x).cuda(), "f32"
        else:
            q = rng.integers(-32768, 32768, ns, dtype=np.int64).astype(np.int16)
            x = q.astype(np.float32) / np.float32(32768.0)
            dev, pcm = torch.from_numpy(q).cuda(), "s16"
        with S.Engine(fft_size=n, hop=hop, window=window, pcm=pcm) as e:
            got = e.stft(dev)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
        del dev
        r = yardstick.error_stats(x, n, hop, 1 if window == "blackman" else 0, {"hip": got}, workers=workers)
        for name in ("hip", "twin_f32", "twin_js"):
            v = r[name]
            print("N=%5d %-8s %-7s %9d bins  %-8s F=1e-5 %.2e  F=2e-5 %.2e  peak max %.2e  peak rms %.2e"
                  % (n, window, kind, r["bins"], name, v["F=1e-05"], v["F=2e-05"], v["peak_max"], v["peak_rms"]), flush=True)
        del got, x
