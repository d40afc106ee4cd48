#!/usr/bin/env python3
"""BASELINE.json configs[3]: streaming 512-channel 48 kHz array, N=4096, hop=1024 -- per-hop latency.
Every hop pushes 512 x 1024 new float32 samples (2.1 MB, host memory) and polls one frame per
channel (bytes or dB floats back to host memory).  Budget: 1024 / 48 kHz = 21.33 ms per hop."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spectrogram_amd as S

def main(hops=1000, n_ch=512, n=4096, hop=1024):
    rng = np.random.default_rng(0x5EED0004)
    a = S.Analyser(fftSize=n, n_channels=n_ch, window="hann")
    a.smoothingTimeConstant = 0
    blocks = [rng.uniform(-1, 1, (n_ch, hop)).astype(np.float32) for _ in range(8)]
    res = {}
    for kind, buf, poll in (("u8", np.zeros((n_ch, n // 2), np.uint8), a.getByteFrequencyData),
                            ("f32", np.zeros((n_ch, n // 2), np.float32), a.getFloatFrequencyData)):
        for i in range(20):
            a.push(blocks[i % 8]); poll(buf)
        t_push, t_poll = [], []
        for i in range(hops):
            t0 = time.perf_counter(); a.push(blocks[i % 8]); t1 = time.perf_counter(); poll(buf); t2 = time.perf_counter()
            t_push.append(t1 - t0); t_poll.append(t2 - t1)
        tot = np.array(t_push) + np.array(t_poll)
        q = lambda v, p: float(np.percentile(v, p) * 1e3)
        res[kind] = {"hop_ms_p50": q(tot, 50), "hop_ms_p99": q(tot, 99), "hop_ms_max": float(tot.max() * 1e3),
                     "push_ms_p50": q(t_push, 50), "poll_ms_p50": q(t_poll, 50)}
    print(json.dumps({"config": "512 ch x N=4096 hop=1024 streaming", "budget_ms": 1024 / 48000 * 1e3, "hops": hops, **res}))

if __name__ == "__main__":
    main()
