#!/usr/bin/env python3
"""Round 6: frames per run (= the stride between the concurrently written regions) against how the output is backed:
hipMalloc of exactly the output's size, a physically contiguous allocation (hipExtMallocWithFlags, hipDeviceMallocContiguous),
the start of a 1 GiB hipMalloc.   usage (GPU box): tools/addr_runlen2.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
OUT_B = FR * BINS * 4
dev = torch.device("cuda", 0)
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]


def raw(flag, size):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(size)) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(size), C.c_uint(flag))
    assert rc == 0, rc
    return p.value


targets = [("hipMalloc exact", raw(None, OUT_B)), ("hipMalloc exact", raw(None, OUT_B)), ("contiguous", raw(0x4, OUT_B)),
           ("contiguous", raw(0x4, OUT_B)), ("hipMalloc 1 GiB", raw(None, 1 << 30)), ("hipMalloc 1 GiB", raw(None, 1 << 30))]
stream = torch.cuda.current_stream(dev).cuda_stream
print("frames per run: us per launch into " + " | ".join(t[0] for t in targets) + "   (runs)")
for rl in (0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 36, 40, 48, 56, 64):
    with S.Engine(fft_size=1024, hop=256, debug=({"run_len": rl} if rl else {})) as e:
        def launch(i):
            e.stft_device_ptr(ins[i % K].data_ptr(), 1, NS, NS, 1, targets[i % len(targets)][1], FR * BINS, stream)
        for i in range(300 if rl == 0 else 36):
            launch(i)
        n = len(targets) * 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            launch(i)
            ev[i + 1].record()
        torch.cuda.synchronize()
        acc = [[] for _ in targets]
        for i in range(n):
            acc[i % len(targets)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
        info = e.last_launch()
    print("  %2d: " % rl + " | ".join("%6.1f" % sorted(a)[len(a) // 2] for a in acc) + "   (%d)" % info[3])
