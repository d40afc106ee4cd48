#!/usr/bin/env python3
"""Round 6, third look at allocation-dependent launch times: per output allocation the STFT launch (three rotating inputs), a
plain fill (write stream) and a plain sum (read stream) of the same bytes; then, if the box shows both classes, the halves of
a slow and of a fast buffer, and the same buffers after a free / re-allocate.   usage (GPU box): tools/addr_ab3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
eng = S.Engine(fft_size=1024, hop=256)
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(3)]


def per_buffer(fn, bufs, reps):
    n = len(bufs) * reps
    for i in range(3 * len(bufs)):
        fn(i, bufs[i % len(bufs)])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(i, bufs[i % len(bufs)])
        ev[i + 1].record()
    torch.cuda.synchronize()
    acc = [[] for _ in bufs]
    for i in range(n):
        acc[i % len(bufs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    return [sorted(a)[len(a) // 2] for a in acc]


def settle():
    o = torch.empty(1, FR, BINS, device=dev)
    for i in range(600):
        eng.stft(ins[i % 3], out=o)
    torch.cuda.synchronize()


def survey(tag, outs):
    settle()
    t_stft = per_buffer(lambda i, o: eng.stft(ins[i % 3], out=o), outs, 15)
    t_fill = per_buffer(lambda i, o: o.fill_(1.0), outs, 15)
    acc = torch.zeros((), device=dev)
    t_sum = per_buffer(lambda i, o: acc.add_(o.sum()), outs, 15)
    print(tag)
    for o, a, b, c in zip(outs, t_stft, t_fill, t_sum):
        print("   %#x  stft %.2f us   fill %.2f us (%.0f GB/s)   sum %.2f us" % (o.data_ptr(), a, b, o.numel() * 4 / b / 1e3, c))
    return t_stft


outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(10)]
t = survey("ten output allocations (median of 15 launches each, interleaved)", outs)
lo, hi = min(t), max(t)
print("spread: %.2f ... %.2f us" % (lo, hi))
if hi - lo > 4.0:
    fast, slow = outs[t.index(lo)], outs[t.index(hi)]
    # halves: a half-length stream (56,248 frames) into the first / second half of the fast and of the slow buffer
    half = FR // 2
    x_half = [x[:, :1024 + (half - 1) * 256].contiguous() for x in ins]
    views = [fast[:, :half], fast[:, half:2 * half], slow[:, :half], slow[:, half:2 * half]]
    th = per_buffer(lambda i, o: eng.stft(x_half[i % 3], out=o), views, 20)
    print("halves (half-length launches): fast buffer %.2f / %.2f us, slow buffer %.2f / %.2f us" % tuple(th))
    # free the slow buffers and allocate again
    keep = [o for o, v in zip(outs, t) if v < lo + 3.0]
    n_new = len(outs) - len(keep)
    del outs, slow, fast, views
    torch.cuda.empty_cache()
    fresh = [torch.empty(1, FR, BINS, device=dev) for _ in range(n_new)]
    survey("after freeing the %d slow ones and allocating %d again (the kept fast ones first)" % (n_new, n_new), keep + fresh)
# inputs: the same question for the input allocation, against one fixed output
out0 = torch.empty(1, FR, BINS, device=dev)
xs = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(8)]
ti = per_buffer(lambda i, x: eng.stft(x, out=out0), xs, 15)
print("eight input allocations against one output: " + " ".join("%.2f" % v for v in ti))
eng.close()
