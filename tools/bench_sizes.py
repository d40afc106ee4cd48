#!/usr/bin/env python3
"""Per-size throughput of the batch path (not the headline bench): frames/s and algorithmic GB/s for
the FFT sizes / hops BASELINE.json's configs use, inputs resident in HBM."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

def run(n, hop, channels, samples, out="power", iters=30):
    x = torch.rand(channels, samples, device="cuda") * 2 - 1
    with S.Engine(fft_size=n, hop=hop, out=out) as e:
        frames = e.num_frames(samples)
        o = torch.empty(channels, frames, e.bins, device="cuda", dtype=torch.uint8 if out == "db_u8" else torch.float32)
        for _ in range(3):
            e.stft(x, out=o)
        torch.cuda.synchronize()
        # median of 5 back-to-back batches: the first tens of ms of a run sit in the clock transient (docs/EXPERIMENTS.md 5.1)
        batches = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                e.stft(x, out=o)
            b.record()
            torch.cuda.synchronize()
            batches.append(a.elapsed_time(b) / iters)
        ms = sorted(batches)[2]
    tot = frames * channels
    bpf = hop * 4 + (n // 2) * (1 if out == "db_u8" else 4)
    print(json.dumps({"fft": n, "hop": hop, "channels": channels, "frames": tot, "out": out, "ms": round(ms, 4),
                      "frames_per_s": round(tot / ms * 1e3), "alg_GBps": round(tot * bpf / ms / 1e6, 1)}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":      # one large N=1024 batch (config-5 shape)
        run(1024, 256, 4096, 160_000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "k10":      # the N = 1024 kernel: config 2, a big batch, bytes
        run(1024, 256, 1, 28_800_000, iters=100)
        run(1024, 256, 4096, 160_000)
        run(1024, 256, 4096, 160_000, out="db_u8")
        run(1024, 512, 2048, 160_000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c2x":      # config 2's shape at 1x, 2x, 4x, 8x the length (launch-size effect)
        for k in (1, 2, 4, 8):
            run(1024, 256, 1, 28_800_000 * k, iters=100 // k + 10)
        for k in (4, 16):
            run(1024, 256, k, 28_800_000 // 4, iters=40)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ref":      # the reference's own operating point, replayed as a batch:
        # fftSize 2048 (UI/player.js:10), one poll per animation frame = 735 samples at 44.1 kHz / 60 fps, Blackman, bytes
        x = torch.rand(64, 44100 * 60, device="cuda") * 2 - 1
        for kw in (dict(out="db_u8"), dict(out="db_u8", logfreq_cols=256), dict(out="power")):
            with S.Engine(fft_size=2048, hop=735, window="blackman", scale="inv_n", **kw) as e:
                o = e.stft(x); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20): e.stft(x, out=o)
                b.record(); torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 20
                fr = 64 * e.num_frames(x.shape[1])
            print(json.dumps({"fft": 2048, "hop": 735, "case": "reference cadence " + "/".join("%s=%s" % kv for kv in kw.items()),
                              "frames": fr, "ms": round(ms, 4), "frames_per_s": round(fr / ms * 1e3)}))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lf":
        x = torch.rand(4096, 160_000, device="cuda") * 2 - 1
        for kw in (dict(out="db_u8", logfreq_cols=256), dict(out="power", logfreq_cols=256)):
            with S.Engine(fft_size=1024, hop=256, window="blackman", scale="inv_n", **kw) as e:
                o = e.stft(x); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20): e.stft(x, out=o)
                b.record(); torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 20
            print(json.dumps({"fft": 1024, "case": "logfreq256 " + kw["out"], "ms": round(ms, 4), "frames_per_s": round(4096 * 622 / ms * 1e3)}))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r03gen":   # round 3: arbitrary-hop loads against their sliding-window neighbours
        run(1024, 300, 64, 1_600_000)
        run(1024, 256, 64, 1_600_000)
        for hop in (735, 512):
            x = torch.rand(64, 44100 * 60, device="cuda") * 2 - 1
            with S.Engine(fft_size=2048, hop=hop, window="blackman", scale="inv_n", out="db_u8") as e:
                o = e.stft(x); torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20): e.stft(x, out=o)
                b.record(); torch.cuda.synchronize()
                ms = a.elapsed_time(b) / 20
                fr = 64 * e.num_frames(x.shape[1])
            print(json.dumps({"fft": 2048, "hop": hop, "out": "db_u8", "window": "blackman", "frames": fr, "ms": round(ms, 4),
                              "frames_per_s": round(fr / ms * 1e3), "alg_GBps": round(fr * (hop * 4 + 1024) / ms / 1e6, 1)}), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "long":     # dispatches of >= 10 ms (GRBM_GUI_ACTIVE / 8 / wall reads the clock there)
        run(1024, 256, 32768, 160_000, iters=6)
        run(1024, 256, 32768, 160_000, out="db_u8", iters=6)
        run(2048, 512, 8192, 320_000, iters=6)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ingest":   # PCM ingest formats: algorithmic bytes = hop * sample bytes + out
        for n, hop, ch, ns in ((1024, 256, 1024, 160_000), (1024, 300, 64, 1_600_000), (2048, 735, 64, 44100 * 60)):
            for pcm, ssz in (("f32", 4), ("s16", 2), ("s24", 3)):
                if pcm == "f32":
                    x = torch.rand(ch, ns, device="cuda") * 2 - 1
                elif pcm == "s16":
                    x = torch.randint(-32768, 32767, (ch, ns), device="cuda", dtype=torch.int16)
                else:
                    x = torch.randint(0, 255, (ch, 3 * ns), device="cuda", dtype=torch.uint8)
                with S.Engine(fft_size=n, hop=hop, pcm=pcm) as e:
                    o = e.stft(x); torch.cuda.synchronize()
                    ts = []
                    for _ in range(5):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record()
                        for _ in range(20): e.stft(x, out=o)
                        b.record(); torch.cuda.synchronize()
                        ts.append(a.elapsed_time(b) / 20)
                    ms = sorted(ts)[2]
                    fr = ch * e.num_frames(ns)
                    var = e.last_launch()[0]
                print(json.dumps({"fft": n, "hop": hop, "pcm": pcm, "load_variant": var, "frames": fr, "ms": round(ms, 4),
                                  "frames_per_s": round(fr / ms * 1e3), "alg_GBps": round(fr * (hop * ssz + n * 2) / ms / 1e6, 1)}), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "smooth":   # batch smoothing, config-3- and config-5-shaped slices: tau = 0 against 0.75
        for n, hop, ch, ns, out in ((2048, 512, 2048, 160_000, "db_u8"), (2048, 512, 2048, 160_000, "power"), (1024, 256, 7813, 160_000, "db_u8")):
            x = torch.rand(ch, ns, device="cuda") * 2 - 1
            for tau in (0.0, 0.75):
                with S.Engine(fft_size=n, hop=hop, window="blackman", scale="inv_n", out=out, smoothing=tau) as e:
                    o = e.stft(x); torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(10): e.stft(x, out=o)
                    b.record(); torch.cuda.synchronize()
                    ms = a.elapsed_time(b) / 10
                    fr = ch * e.num_frames(ns)
                    info = e.last_launch()
                print(json.dumps({"fft": n, "hop": hop, "channels": ch, "out": out, "tau": tau, "run_len": info[1], "frames": fr, "ms": round(ms, 4),
                                  "frames_per_s": round(fr / ms * 1e3)}), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "generic":
        run(1024, 300, 64, 1_600_000)
        run(1024, 301, 64, 1_600_000)
        run(2048, 736, 64, 1_600_000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "u8":
        run(1024, 256, 4096, 160_000, out="db_u8")
        run(2048, 512, 256, 1_440_000, out="db_u8")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r04":      # round 4's marks: N = 2048 / 4096 / 8192
        run(2048, 512, 1024, 320_000)
        run(2048, 512, 1024, 320_000, out="db_u8")
        run(4096, 1024, 512, 640_000)
        run(4096, 1024, 512, 640_000, out="db_u8")
        run(4096, 1111, 512, 640_000)                  # LDS ring
        run(8192, 2048, 256, 1_280_000)
        run(8192, 4096, 256, 1_280_000)
        run(8192, 2940, 256, 1_280_000)                # LDS ring
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5":       # config 5's chunk (7,813 utterances: 5 GB in, 10 GB out) against smaller batches
        for ch in (1024, 4096, 7813, 7813):
            run(1024, 256, ch, 160_000, iters=10)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n2048":
        run(2048, 512, 256, 1_440_000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "multi":
        run(2048, 512, 256, 1_440_000)
        run(4096, 1024, 512, 1_024_000)
        run(8192, 2048, 64, 4_096_000)
        run(2048, 1024, 256, 1_440_000)
        sys.exit(0)
    run(1024, 256, 1, 28_800_000)              # config 2
    run(1024, 256, 64, 1_600_000)              # bigger batch of the same shape
    run(1024, 256, 4096, 160_000)              # config 5 slice: 4096 utterances x 10 s @ 16 kHz
    run(2048, 512, 256, 1_440_000)             # config 3 slice: 256 channels x 30 s
    run(4096, 1024, 512, 1_024_000)            # config 4 shape as a batch
    run(256, 128, 64, 4_800_000)               # config 1 shape, batched
    run(512, 128, 64, 2_400_000)
    run(8192, 2048, 64, 4_096_000)
    run(1024, 512, 64, 1_600_000)              # hop N/2
    run(1024, 300, 64, 1_600_000)              # generic path
    run(1024, 256, 64, 1_600_000, out="db")
    run(1024, 256, 64, 1_600_000, out="db_u8")
    run(2048, 512, 256, 1_440_000, out="db_u8")
