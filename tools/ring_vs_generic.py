import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
def run(n, hop, ch, ns, out="power"):
    x = torch.rand(ch, ns, device="cuda") * 2 - 1
    res = {}
    for tag, env in (("ring", None), ("generic", "1")):
        if env: os.environ["SG_NO_LRING"] = env
        else: os.environ.pop("SG_NO_LRING", None)
        with S.Engine(fft_size=n, hop=hop, out=out) as e:
            o = e.stft(x); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10): e.stft(x, out=o)
                b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) / 10)
            fr = ch * e.num_frames(ns)
            res[tag] = (round(fr / sorted(ts)[2] * 1e3 / 1e6, 1), e.last_launch()[:2])
    os.environ.pop("SG_NO_LRING", None)
    print(n, hop, out, res, flush=True)
for n, hop in ((1024, 100), (1024, 300), (1024, 700), (1024, 1000), (2048, 200), (2048, 735), (2048, 1500), (4096, 1000), (4096, 1470), (4096, 3000), (8192, 2000), (8192, 2940), (8192, 6000)):
    run(n, hop, 64, 64 * 44100)
