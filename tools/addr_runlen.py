#!/usr/bin/env python3
"""Round 6: the config-2 launch is 5-10 % slower into some output allocations than into others (tools/addr_grid.py).  Is it the
stride between the concurrently written regions (run length x 2 KiB rows) meeting that allocation's physical placement?
Frames per run 12 ... 44 on the fastest and on the slowest of eight output allocations, inputs cycling through eight.
usage (GPU box): tools/addr_runlen.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(K)]


def timed(eng, targets, reps=14):
    """median us per launch into each target output; inputs cycle through all eight"""
    seq = [(i % K, targets[i % len(targets)]) for i in range(len(targets) * reps)]
    for a, o in seq[:2 * len(targets)]:
        eng.stft(ins[a], out=o)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
    ev[0].record()
    for j, (a, o) in enumerate(seq):
        eng.stft(ins[a], out=o)
        ev[j + 1].record()
    torch.cuda.synchronize()
    acc = [[] for _ in targets]
    for j in range(len(seq)):
        acc[j % len(targets)].append(ev[j].elapsed_time(ev[j + 1]) * 1e3)
    return [sorted(a)[len(a) // 2] for a in acc]


with S.Engine(fft_size=1024, hop=256) as e0:
    for i in range(800):
        e0.stft(ins[i % K], out=outs[i % K])
    torch.cuda.synchronize()
    base = timed(e0, outs)
    print("default geometry, eight outputs: " + " ".join("%.1f" % v for v in base) + "   launch: %s" % (e0.last_launch(),))
fast, slow = outs[base.index(min(base))], outs[base.index(max(base))]
mid = outs[sorted(range(K), key=lambda i: base[i])[K // 2]]
print("frames per run -> us per launch into the fastest / a middle / the slowest output (runs, stride between runs in KiB)")
for rl in (0, 12, 14, 16, 18, 20, 22, 24, 25, 26, 27, 28, 29, 30, 31, 32, 34, 36, 40, 44):
    with S.Engine(fft_size=1024, hop=256, debug=({"run_len": rl} if rl else {})) as e:
        t = timed(e, [fast, mid, slow])
        info = e.last_launch()
    print("  %2d: %.1f / %.1f / %.1f   (%d runs, %d KiB)" % (rl, t[0], t[1], t[2], info[3], info[1] * 2))
