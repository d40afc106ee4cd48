import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
import spectrogram_amd as S
n, hop, ch, ns, iters = 2048, 512, 2048, 1_440_000, 5
x = torch.rand(ch, ns, device="cuda") * 2 - 1
frames = (ns - n) // hop + 1
o = torch.empty(ch, frames, n // 2, device="cuda")
lens = [0, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 144, 160, 176, 192, 208, 224, 240, 256, 288, 320, 352, 384]
engines = {r: S.Engine(fft_size=n, hop=hop, debug=({"run_len": r} if r else {})) for r in lens}
for r in lens:
    for _ in range(2): engines[r].stft(x, out=o)
torch.cuda.synchronize()
res = {r: [] for r in lens}
for rnd in range(3):
    for r in (lens if rnd % 2 == 0 else lens[::-1]):
        e = engines[r]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e.stft(x, out=o)
        a.record()
        for _ in range(iters): e.stft(x, out=o)
        b.record(); torch.cuda.synchronize()
        res[r].append(round(a.elapsed_time(b) / iters, 4))
for r in lens:
    rpc = (frames + (r or 64) - 1) // (r or 64)
    print(json.dumps({"run_len": r, "runs_per_channel": rpc, "last_run": frames - (rpc - 1) * (r or 64), "ms": res[r], "median": sorted(res[r])[1]}))
