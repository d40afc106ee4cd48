#!/usr/bin/env python3
"""How far does float32 FFT noise reach into the stated tolerance?  max over ~1e8 bins per size of
|P_gpu - P_ref| / max(P_ref, F max_k P_ref) for floors F = 1e-5, 2e-5, 3e-5 against a float64 FFT of the same float32
samples (torch.fft on the device: an independent implementation, used here only because the oracle is too slow for 1e8
bins); uniform and int16-like inputs, Hann and Blackman."""
import os, sys
sys.path.insert(0, os.getcwd())
import math, torch
import spectrogram_amd as S

torch.manual_seed(5)
for n in (256, 512, 1024, 2048, 4096, 8192):
    hop = n // 2 + 5
    frames = max(2000, 50_000_000 // n)
    for window, kind in (("hann", "uniform"), ("blackman", "int16")):
        ns = n + (frames - 1) * hop
        if kind == "uniform":
            x = torch.rand(4, ns, device="cuda") * 2 - 1
        else:
            x = torch.randint(-32768, 32768, (4, ns), device="cuda").float() / 32768.0
        with S.Engine(fft_size=n, hop=hop, window=window) as e:
            got = e.stft(x).double()
        k = torch.arange(n, device="cuda", dtype=torch.float64)
        w = 0.5 - 0.5 * torch.cos(2 * math.pi * k / n) if window == "hann" else \
            0.42 - 0.5 * torch.cos(2 * math.pi * k / n) + 0.08 * torch.cos(4 * math.pi * k / n)
        worst = {1e-5: 0.0, 2e-5: 0.0, 3e-5: 0.0}
        peak_noise = 0.0
        for c in range(4):
            fr = x[c].double().unfold(0, n, hop) * w
            ref = torch.fft.rfft(fr, dim=1)[:, : n // 2].abs() ** 2
            mx = ref.max(dim=1, keepdim=True).values
            d = (got[c] - ref).abs()
            peak_noise = max(peak_noise, float((d / mx).max()))
            for f in worst:
                worst[f] = max(worst[f], float((d / torch.maximum(ref, f * mx)).max()))
            del fr, ref, d
        print("N=%5d %-8s %-7s %9d bins: max err / (floor-normalised P): F=1e-5 %.2e  F=2e-5 %.2e  F=3e-5 %.2e ; max |dP| / frame peak %.2e"
              % (n, window, kind, 4 * frames * (n // 2), worst[1e-5], worst[2e-5], worst[3e-5], peak_noise), flush=True)
