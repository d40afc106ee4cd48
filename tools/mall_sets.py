#!/usr/bin/env python3
"""Round 6: how many rotating buffer sets does the config-2 launch need before nothing of its 115 MB input survives in the
256 MiB Infinity Cache from its previous use?  K input / output pairs, launched round robin (as bench.py does with K = 3) and
in a shuffled order; per-set and overall launch times from a timed region of back-to-back launches (HIP events around the
whole region, so no per-launch marker gaps) and from per-launch events.    usage (GPU box): tools/mall_sets.py"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
eng = S.Engine(fft_size=1024, hop=256)
KMAX = 16
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(KMAX)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(KMAX)]
for i in range(800):                                   # settle the clock
    eng.stft(ins[i % 3], out=outs[i % 3])
torch.cuda.synchronize()
random.seed(6)
print("K sets: region = us per launch over 240 back-to-back launches (one event pair); per set = median of per-launch events")
for k in (1, 2, 3, 4, 5, 6, 8, 12, 16):
    for order_name in ("round robin", "shuffled"):
        n = 240
        seq = [i % k for i in range(n)]
        if order_name == "shuffled":
            if k < 3:
                continue
            seq = []
            while len(seq) < n:
                p = list(range(k)); random.shuffle(p); seq += p
            seq = seq[:n]
        for i in seq[:60]:
            eng.stft(ins[i], out=outs[i])
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in seq:
            eng.stft(ins[i], out=outs[i])
        b.record(); torch.cuda.synchronize()
        region = a.elapsed_time(b) * 1e3 / n
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for j, i in enumerate(seq):
            eng.stft(ins[i], out=outs[i])
            ev[j + 1].record()
        torch.cuda.synchronize()
        per = [[] for _ in range(k)]
        for j, i in enumerate(seq):
            per[i].append(ev[j].elapsed_time(ev[j + 1]) * 1e3)
        med = [sorted(p)[len(p) // 2] for p in per]
        print("K = %2d %-11s region %.2f us   per set: %s" % (k, order_name, region, " ".join("%.1f" % m for m in med)))
eng.close()
