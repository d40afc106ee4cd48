#!/usr/bin/env python3
"""Same-box, interleaved A/B of the bench launch's run geometry: the default (one exact round of 28-frame runs, waves
ranked by remaining work), the same 28-frame runs without the ranking (SG_RUN_LEN=28), 16-frame runs
(SG_NO_SINGLE_ROUND=1) and a few other run lengths.  Per round and engine: `settle` launches untimed, then 200 timed by
HIP events; `--long` keeps every engine under load for 1.5 s first (the sustained state)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
import bench


def make(**env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return S.Engine(fft_size=1024, hop=256)
    finally:
        for k in env: del os.environ[k]


x = [torch.rand(1, bench.N_SAMPLES, device="cuda") * 2 - 1 for _ in range(3)]
o = [torch.empty(1, bench.FRAMES, 512, device="cuda") for _ in range(3)]
engines = [("default (28, ranked)", make()), ("SG_RUN_LEN=28", make(SG_RUN_LEN=28)), ("SG_NO_SINGLE_ROUND (16)", make(SG_NO_SINGLE_ROUND=1)),
           ("SG_RUN_LEN=24", make(SG_RUN_LEN=24)), ("SG_RUN_LEN=30", make(SG_RUN_LEN=30)), ("SG_RUN_LEN=32", make(SG_RUN_LEN=32))]
long_s = 1.5 if "--long" in sys.argv else 0.0
res = {n: [] for n, _ in engines}
for rnd in range(0 if "--buffers" in sys.argv else 5):
    for name, e in engines:
        t0 = time.perf_counter(); i = 0
        while True:
            for _ in range(100):
                e.stft(x[i % 3], out=o[i % 3]); i += 1
            torch.cuda.synchronize()
            if i >= 300 and time.perf_counter() - t0 >= long_s: break
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(100):
            e.stft(x[i % 3], out=o[i % 3]); i += 1
        a.record()
        for _ in range(200):
            e.stft(x[i % 3], out=o[i % 3]); i += 1
        b.record(); torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / 200 * 1e3)
if "--buffers" in sys.argv:
    # one input / output pair against three rotating pairs, same engine, interleaved: does the PCM of a single pair stay in
    # the 256 MiB Infinity Cache from launch to launch?
    e = engines[0][1]
    for rnd in range(4):
        for sets in (1, 3, 2):
            i = 0
            for _ in range(400):
                e.stft(x[i % sets], out=o[i % sets]); i += 1
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200):
                e.stft(x[i % sets], out=o[i % sets]); i += 1
            b.record(); torch.cuda.synchronize()
            print("  %d buffer pair(s): %.1f us/launch" % (sets, a.elapsed_time(b) / 200 * 1e3), flush=True)
    sys.exit(0)
for name, _ in engines:
    v = res[name]
    print("%-26s %s  median %.1f us/launch (run_len %d)" % (name, " ".join("%.1f" % t for t in v), sorted(v)[len(v) // 2], dict(engines)[name].last_launch()[1]), flush=True)
