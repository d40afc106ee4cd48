#!/usr/bin/env python3
"""Throughput against launch size with bench.py's buffer rotation (three input / output sets, nothing reused out of the
Infinity Cache): the bench stream (10 min, 112,497 frames) as 1, 2, 4, 8, 16 channels per launch, and shorter streams.
Which part of the gap between the bench launch and a large batch is the size of the launch?"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
import bench

with S.Engine(fft_size=1024, hop=256) as e:
    for ch, ns in ((1, bench.N_SAMPLES // 4), (1, bench.N_SAMPLES // 2), (1, bench.N_SAMPLES), (2, bench.N_SAMPLES), (4, bench.N_SAMPLES),
                   (8, bench.N_SAMPLES), (16, bench.N_SAMPLES), (4096, 160_000)):
        sets = 3 if ch * ns * 12 * 3 < 40e9 else 2
        x = [torch.rand(ch, ns, device="cuda") * 2 - 1 for _ in range(sets)]
        o = [torch.empty(ch, e.num_frames(ns), 512, device="cuda") for _ in range(sets)]
        frames = ch * e.num_frames(ns)
        reps = max(20, int(60e-3 / (frames / 1.5e9)))          # ~60 ms settle, ~60 ms timed
        i = 0
        res = []
        for rnd in range(3):
            for _ in range(reps):
                e.stft(x[i % sets], out=o[i % sets]); i += 1
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                e.stft(x[i % sets], out=o[i % sets]); i += 1
            b.record(); torch.cuda.synchronize()
            res.append(a.elapsed_time(b) / reps)
        ms = sorted(res)[1]
        info = e.last_launch_info() if hasattr(e, "last_launch_info") else None
        v, rl, single, runs = e.last_launch()
        print("%5d ch x %9d samples: %9d frames/launch  %8.1f us/launch  %.3fe9 frames/s  %.0f GB/s alg   (run_len %d, %d runs, single_round %d, %d sets)"
              % (ch, ns, frames, ms * 1e3, frames / ms / 1e6, frames * 3072 / ms / 1e6, rl, runs, single, sets), flush=True)
        del x, o
