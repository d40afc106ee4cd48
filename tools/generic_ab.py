#!/usr/bin/env python3
"""Round 5: the generic loads of plain rows (unit stride, float32, no down-mix) as an instantiation of their own (load variant 5:
100-112 VGPRs, four waves per SIMD) against the general generic kernel (variant 0: 148-154 VGPRs, three; N = 16384 spilled 140 B per
lane), same process, interleaved.  The general kernel is reached here through an input whose base is misaligned... no: through
sg_debug_set "no_gen_unit"."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

cases = [(512, 100, 256, 400_000, "power"), (256, 100, 256, 400_000, "power"), (1024, 1024, 1024, 320_000, "power"), (1024, 1024, 1024, 320_000, "db_u8"),
         (2048, 2048, 512, 640_000, "power"), (512, 200, 256, 400_000, "db_u8"), (16384, 5000, 64, 2_000_000, "power"), (32768, 10000, 32, 4_000_000, "power"),
         (1024, 300, 64, 1_600_000, "power")]
for n, hop, ch, ns, out in cases:
    x = torch.rand(ch, ns, device="cuda") * 2 - 1
    dbg = {"no_lring": 1} if (n, hop) == (1024, 300) else {}
    eng = {"unit": S.Engine(fft_size=n, hop=hop, out=out, debug=dict(dbg)), "general": S.Engine(fft_size=n, hop=hop, out=out, debug=dict(dbg, no_gen_unit=1))}
    outs = {k: e.stft(x) for k, e in eng.items()}
    torch.cuda.synchronize()
    variants = {k: e.last_launch()[0] for k, e in eng.items()}
    same = bool(torch.equal(outs["unit"], outs["general"]))
    res = {k: [] for k in eng}
    for rnd in range(3):
        for k, e in eng.items():
            for _ in range(3):
                e.stft(x, out=outs[k])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                e.stft(x, out=outs[k])
            b.record(); torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 20)
    fr = ch * eng["unit"].num_frames(ns)
    med = {k: sorted(v)[1] for k, v in res.items()}
    print(json.dumps({"fft": n, "hop": hop, "out": out, "variants": variants, "bit_equal": same, "frames_per_s": {k: round(fr / m * 1e3) for k, m in med.items()},
                      "gain_pct": round((med["general"] / med["unit"] - 1) * 100, 2)}), flush=True)
    for e in eng.values():
        e.close()
