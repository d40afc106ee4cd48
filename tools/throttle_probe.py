#!/usr/bin/env python3
"""Where config 4's 12-73 ms single-hop host stalls came from: the harness, not the library.

torch sizes its intra-op OpenMP pool by the HOST's cores; this container may only use cgroup cpu.max of them.  After a CPU
tensor op (the streaming leg builds its page-locked inputs with torch.rand) the pool's threads spin; with more spinners
than quota cores the cgroup's CPU budget for the 100 ms period is gone within milliseconds and the kernel parks every
thread of the process until the next period.  The hop that happens to be waiting in stream.synchronize() then shows
tens of milliseconds on the host clock with the device idle.

Each scenario runs in a fresh process (the pool is sized once): the per-hop loop of bench.py's streaming leg (512
channels, N = 4096, hop 1024, 256 log-frequency byte columns, page-locked buffers in place), 6,000 hops, right after four
CPU tensor ops.
    default      torch's own thread count, loop starts right behind the CPU ops
    limited      torch.set_num_threads(cgroup quota) first -- what bench.py does now
    settled      default thread count, 0.5 s pause between the CPU ops and the loop
Prints max / p99.9 host wall, the slowest hop's stream.synchronize() share, and the cgroup's throttle counters."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_stat():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except Exception:
        return (0, 0)


def scenario(kind):
    import ctypes as C
    import numpy as np
    import torch
    import bench
    from spectrogram_amd import _lib as L
    import spectrogram_amd as S
    default_threads = torch.get_num_threads()
    if kind == "limited":
        torch.set_num_threads(bench.host_cores())
    lib = L.load()
    n, hop, n_ch, cols, hops, warm = 4096, 1024, 512, 256, 6000, 20   # 6,000 hops ~ 0.45 s: several 100 ms quota periods
    cfg = S.make_config(fft_size=n, hop=n, window="blackman", scale="inv_n", out="mag", logfreq_cols=cols)
    h = C.c_void_p()
    L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
    L.check(lib.sg_stream_begin(h, n_ch))
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)
    pin_out = torch.empty(n_ch, cols, dtype=torch.uint8).pin_memory()
    warm_in = torch.zeros(n_ch, hop).pin_memory()
    for _ in range(warm):   # first-use costs (code-object load of the push / poll kernels: ~6 ms) stay out of the picture
        L.check(lib.sg_push(h, warm_in.data_ptr(), hop, hop, L.IN_PINNED, sp))
        L.check(lib.sg_poll_u8(h, pin_out.data_ptr(), cols, cols, L.OUT_PINNED, sp))
        stream.synchronize()
    time.sleep(0.5)         # whatever the set-up above left spinning has gone quiet
    torch.cuda.synchronize()
    t0s = cpu_stat()
    # the CPU tensor ops of bench.py's streaming leg, right in front of the loop: they wake torch's whole OpenMP pool
    pin_in = [(torch.rand(n_ch, hop) * 2 - 1).pin_memory() for _ in range(4)]
    big = torch.rand(4096, 4096) @ torch.rand(4096, 64)
    if kind == "settled":
        time.sleep(0.5)
    wall, sync = np.empty(hops), np.empty(hops)
    for i in range(hops):
        t0 = time.perf_counter()
        L.check(lib.sg_push(h, pin_in[i & 3].data_ptr(), hop, hop, L.IN_PINNED, sp))
        L.check(lib.sg_poll_u8(h, pin_out.data_ptr(), cols, cols, L.OUT_PINNED, sp))
        t1 = time.perf_counter()
        stream.synchronize()
        t2 = time.perf_counter()
        wall[i], sync[i] = (t2 - t0) * 1e3, (t2 - t1) * 1e3
    t1s = cpu_stat()
    lib.sg_destroy(h)
    k = int(wall.argmax())
    print(json.dumps({"scenario": kind, "torch_threads": torch.get_num_threads(), "torch_threads_default": default_threads,
                      "cpu_quota_cores": bench.host_cores(), "os_cpu_count": os.cpu_count(),
                      "host_wall_ms": {"p50": float(np.percentile(wall, 50)), "p99_9": float(np.percentile(wall, 99.9)), "max": float(wall.max())},
                      "slowest_hop": {"hop": k, "host_wall_ms": float(wall[k]), "in_stream_synchronize_ms": float(sync[k])},
                      "cgroup_throttled_periods": t1s[0] - t0s[0], "cgroup_throttled_ms": (t1s[1] - t0s[1]) / 1e3,
                      "keep": float(big.sum())}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
    else:
        try:
            print("# cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
        except Exception:
            pass
        for rep in range(3):
            for kind in ("default", "limited", "settled"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), kind], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "%s: failed" % kind, flush=True)
