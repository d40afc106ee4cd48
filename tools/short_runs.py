#!/usr/bin/env python3
"""Frames per run for launches too small to fill the chip (SG_RUN_LEN sweep on 7k ... 56k-frame streams, rotating buffers)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
import bench
FFT, HOP = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 256)
def make(rl):
    if rl: os.environ["SG_RUN_LEN"] = str(rl)
    try: return S.Engine(fft_size=FFT, hop=HOP)
    finally: os.environ.pop("SG_RUN_LEN", None)
print("N=%d hop=%d" % (FFT, HOP), flush=True)
for ns in [FFT + HOP * (f - 1) for f in (1800, 3500, 7028, 14059, 28122, 56247)] if len(sys.argv) > 2 else \
        (bench.N_SAMPLES // 16, bench.N_SAMPLES // 8, bench.N_SAMPLES // 4, bench.N_SAMPLES // 2):
    x = [torch.rand(1, ns, device="cuda") * 2 - 1 for _ in range(3)]
    engs = [(rl, make(rl)) for rl in (0, 4, 6, 7, 8, 10, 12, 14)]
    o = [torch.empty(1, engs[0][1].num_frames(ns), FFT // 2, device="cuda") for _ in range(3)]
    frames = engs[0][1].num_frames(ns)
    res = {rl: [] for rl, _ in engs}
    for rnd in range(3):
        for rl, e in engs:
            i = 0
            for _ in range(600):
                e.stft(x[i % 3], out=o[i % 3]); i += 1
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(400):
                e.stft(x[i % 3], out=o[i % 3]); i += 1
            b.record(); torch.cuda.synchronize()
            res[rl].append(a.elapsed_time(b))
    print(frames, "frames:", "  ".join("rl %s: %.1f us (%d runs)" % (rl or "auto", sorted(res[rl])[1], e.last_launch()[3]) for rl, e in engs), flush=True)
    for _, e in engs: e.close()
