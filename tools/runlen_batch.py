#!/usr/bin/env python3
"""Round 6: frames per run on LARGE batches (the launcher caps runs at 64 frames; tools/residency.py showed a run of the
N = 2048 kernel spending ~15 % of its life in front of its first frame).  Same process, interleaved over two rounds:
    tools/runlen_batch.py            config 3 at full size, a 256-channel slice of it, config 5's chunk, a 4096-utterance batch"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

SHAPES = [(2048, 512, 2048, 1_440_000, 6), (2048, 512, 256, 1_440_000, 20), (1024, 256, 7813, 160_000, 8), (1024, 256, 4096, 160_000, 12),
          (4096, 1024, 512, 1_024_000, 12)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(a)] for a in sys.argv[1:]]
for n, hop, ch, ns, iters in SHAPES:
    x = torch.rand(ch, ns, device="cuda") * 2 - 1
    frames = (ns - n) // hop + 1
    o = torch.empty(ch, frames, n // 2, device="cuda")
    lens = [0, 32, 64, 96, 128, 192, 256, 512, 1024, frames]
    lens = [r for r in lens if r <= frames]
    res = {r: [] for r in lens}
    engines = {r: S.Engine(fft_size=n, hop=hop, debug=({"run_len": r} if r else {})) for r in lens}
    for r in lens:                                   # warm every engine, settle the clock
        for _ in range(3):
            engines[r].stft(x, out=o)
    torch.cuda.synchronize()
    info = {}
    for rnd in range(3):
        for r in lens:
            e = engines[r]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e.stft(x, out=o)
            a.record()
            for _ in range(iters):
                e.stft(x, out=o)
            b.record(); torch.cuda.synchronize()
            res[r].append(a.elapsed_time(b) / iters)
            info[r] = e.last_launch()
    for r in lens:
        ms = sorted(res[r])[1]
        print(json.dumps({"fft": n, "hop": hop, "channels": ch, "frames_per_channel": frames, "run_len_asked": r, "run_len": info[r][1],
                          "total_runs": info[r][3], "ms": [round(v, 4) for v in res[r]], "median_ms": round(ms, 4),
                          "frames_per_s": round(ch * frames / ms * 1e3), "alg_GBps": round(ch * frames * (hop * 4 + n * 2) / ms / 1e6, 1)}), flush=True)
    for e in engines.values():
        e.close()
    del x, o
    torch.cuda.empty_cache()
