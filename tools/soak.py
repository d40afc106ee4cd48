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
         (8192, 2941, 32, 1_600_000, "power")]
bad = 0
for n, hop, ch, ns, out in cases:
    g = torch.Generator(device="cuda").manual_seed(n + hop)
    x = torch.rand(ch, ns, device="cuda", generator=g) * 2 - 1
    kw = dict(logfreq_cols=256) if (n, hop) == (2048, 735) else {}
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
print("soak: %s" % ("FAILED" if bad else "all bit-identical"))
sys.exit(1 if bad else 0)
