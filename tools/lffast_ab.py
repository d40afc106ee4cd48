#!/usr/bin/env python3
"""Round 5: the 256-column log-frequency rows of N = 1024 / 2048 through their own instantiation (OUT_LFFAST: the consecutive-column
path alone) against the general log-frequency kernel (sg_debug_set "no_lffast"), same process, interleaved; outputs compared bit for bit."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

cases = [(1024, 256, 4096, 160_000, dict(out="db_u8", window="blackman", scale="inv_n")),
         (1024, 256, 4096, 160_000, dict(out="power", window="blackman", scale="inv_n")),
         (2048, 512, 1024, 320_000, dict(out="db_u8", window="blackman", scale="inv_n")),
         (2048, 735, 64, 44100 * 60, dict(out="db_u8", window="blackman", scale="inv_n")),
         (2048, 735, 64, 44100 * 60, dict(out="power", window="blackman", scale="inv_n", logfreq_align="vertex"))]
for n, hop, ch, ns, kw in cases:
    x = torch.rand(ch, ns, device="cuda") * 2 - 1
    eng = {"lffast": S.Engine(fft_size=n, hop=hop, logfreq_cols=256, debug={}, **kw),
           "general": S.Engine(fft_size=n, hop=hop, logfreq_cols=256, debug={"no_lffast": 1}, **kw)}
    outs = {k: e.stft(x) for k, e in eng.items()}
    torch.cuda.synchronize()
    same = bool(torch.equal(outs["lffast"], outs["general"]))
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
    fr = ch * eng["lffast"].num_frames(ns)
    med = {k: sorted(v)[1] for k, v in res.items()}
    print(json.dumps({"fft": n, "hop": hop, "out": kw["out"], "bit_equal": same, "frames_per_s": {k: round(fr / m * 1e3) for k, m in med.items()},
                      "gain_pct": round((med["general"] / med["lffast"] - 1) * 100, 2)}), flush=True)
    for e in eng.values():
        e.close()
