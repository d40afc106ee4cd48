#!/usr/bin/env python3
"""Run-to-run determinism soak: every configuration's output must be bit-identical launch after launch (a race between
waves that exchange through LDS would show up as a difference).  usage: tools/soak.py [launches per configuration]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cases = [(1024, 256, 512, 400_000, "power"), (1024, 256, 512, 400_000, "db_u8"), (1024, 512, 256, 400_000, "power"),
         (2048, 512, 256, 720_000, "power"), (2048, 512, 256, 720_000, "db_u8"), (2048, 735, 64, 720_000, "db_u8"),
         (4096, 1024, 256, 512_000, "power"), (8192, 2048, 64, 1_024_000, "power"), (512, 128, 64, 1_200_000, "power"),
         (256, 128, 64, 1_200_000, "db"), (1024, 300, 64, 800_000, "power"), (16384, 4096, 16, 1_000_000, "power"),
         (32768, 8192, 8, 1_000_000, "power"),
         # round 3: the LDS sample ring in both layouts, odd and even hops, K = 2 / K = 4 staging, multi-wave sizes
         (2048, 735, 64, 720_000, "power"), (2048, 736, 64, 720_000, "db_u8"), (2048, 1501, 64, 720_000, "power"),
         (1024, 301, 64, 800_000, "db_u8"), (1024, 777, 64, 800_000, "power"), (4096, 1471, 64, 800_000, "power"),
         (8192, 2941, 32, 1_600_000, "power"),
         # round 4: the log-frequency epilogue's consecutive-column path (256 columns at N = 1024 / 2048), both alignments
         (1024, 256, 256, 400_000, "db_u8", dict(logfreq_cols=256, logfreq_align="vertex")),
         (1024, 256, 256, 400_000, "db", dict(logfreq_cols=256, logfreq_align="centre")),
         (2048, 512, 128, 720_000, "db_u8", dict(logfreq_cols=256, logfreq_align="vertex")),
         (2048, 512, 128, 720_000, "power", dict(logfreq_cols=256, logfreq_align="centre")),
         (4096, 1024, 64, 720_000, "db_u8", dict(logfreq_cols=256, logfreq_align="vertex")),
         # round 6: batch smoothing with log-frequency columns (fused: a run is a whole channel; two-pass: scratch + scan),
         # columns at N = 32768, the W3C speakers down-mix in the framing load
         (1024, 256, 2100, 20_000, "db_u8", dict(smoothing=0.75, logfreq_cols=256, logfreq_align="vertex", window="blackman", scale="inv_n")),
         (2048, 512, 1100, 40_000, "mag", dict(smoothing=0.6, logfreq_cols=256, window="blackman", scale="inv_n")),
         (2048, 512, 3, 720_000, "power", dict(smoothing=0.75, logfreq_cols=256, window="blackman", scale="inv_n")),
         (32768, 8192, 8, 1_000_000, "db_u8", dict(logfreq_cols=256)),
         (1024, 256, 64, 600_000, "power", dict(downmix=6, downmix_mode="speakers"))]
bad = 0
for n, hop, ch, ns, out, *extra in cases:
    g = torch.Generator(device="cuda").manual_seed(n + hop)
    x = torch.rand(ch, ns, device="cuda", generator=g) * 2 - 1
    kw = extra[0] if extra else (dict(logfreq_cols=256) if (n, hop) == (2048, 735) else {})
    with S.Engine(fft_size=n, hop=hop, out=out, **kw) as e:
        ref = e.stft(x).clone()
        o = torch.empty_like(ref)
        t0 = time.time()
        diff = 0
        for i in range(reps):
            e.stft(x, out=o)
            if i % 8 == 7 or i == reps - 1:          # compare every 8th launch (the compare itself is a big kernel)
                if not torch.equal(o, ref):
                    diff += 1
        torch.cuda.synchronize()
    print("N=%-5d hop=%-4d ch=%-3d %-6s %s: %d launches, %d differing  (%.1f s)" % (n, hop, ch, out, "logfreq" if kw else "", reps, diff, time.time() - t0), flush=True)
    bad += diff

# round 4, poll mode: two handles fed the same blocks -- one with sg_push + sg_poll_u8, one with the fused sg_push_poll_u8
# (one launch that reads the samples in place and files them in the history itself) -- must hand out the same bytes hop
# after hop, ragged block sizes included; and a third handle replaying the fused sequence must repeat it bit for bit.
import ctypes as C
import numpy as np
from spectrogram_amd import _lib as L
from spectrogram_amd.engine import make_config

lib = L.load()


def handle(n, ch, cols, pcm):
    cfg = make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag", smoothing=0.5, logfreq_cols=cols,
                      pcm=pcm, logfreq_align="vertex", history_samples=32768)
    h = C.c_void_p()
    L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
    L.check(lib.sg_stream_begin(h, ch))
    L.check(lib.sg_ring_begin(h, 64))
    return h


for n, ch, cols, pcm in [(2048, 64, 0, "f32"), (2048, 64, 256, "s16"), (4096, 512, 0, "f32"), (4096, 128, 256, "f32"),
                         (1024, 32, 256, "f32"), (8192, 16, 0, "s16"), (512, 8, 0, "f32")]:
    rng = np.random.default_rng(n + ch)
    hops = max(40, reps)
    sizes = [int(rng.integers(1, n // 2 + 1)) * 2 for _ in range(hops)]          # even, 2 ... N
    row = cols if cols else n // 2
    hs = [handle(n, ch, cols, pcm) for _ in range(3)]
    blocks = []
    for b in sizes:
        v = rng.uniform(-1, 1, (ch, b)).astype(np.float32)
        t = torch.from_numpy((v * 32767).astype(np.int16) if pcm == "s16" else v).pin_memory()
        blocks.append(t)
    outs = [torch.empty(ch, row, dtype=torch.uint8).pin_memory() for _ in range(3)]
    diff = 0
    t0 = time.time()
    for t, b in zip(blocks, sizes):
        L.check(lib.sg_push(hs[0], t.data_ptr(), b, b, L.IN_PINNED, None))
        L.check(lib.sg_poll_u8(hs[0], outs[0].data_ptr(), row, row, L.OUT_PINNED, None))
        for k in (1, 2):
            L.check(lib.sg_push_poll_u8(hs[k], t.data_ptr(), b, b, outs[k].data_ptr(), row, row, L.IN_PINNED | L.OUT_PINNED, None))
        for h in hs:
            L.check(lib.sg_synchronize(h, None))
        if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])):
            diff += 1
    rings = []
    for h in hs:
        r = np.empty((ch, 64, row), np.uint8)
        L.check(lib.sg_ring_read(h, r.ctypes.data, 0, L.PTR_HOST, None))
        rings.append(r)
        lib.sg_destroy(h)
    diff += int(not (np.array_equal(rings[0], rings[1]) and np.array_equal(rings[1], rings[2])))
    print("poll N=%-5d ch=%-3d cols=%-3d %s: %d hops (two calls | fused | fused again), %d differing  (%.1f s)"
          % (n, ch, cols, pcm, hops, diff, time.time() - t0), flush=True)
    bad += diff
print("soak: %s" % ("FAILED" if bad else "all bit-identical"))
sys.exit(1 if bad else 0)
