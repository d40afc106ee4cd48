#!/usr/bin/env python3
"""Round 6: the config-2 launch's duration follows its OUTPUT allocation (tools/addr_grid.py).  Does the way the output is
allocated decide it?  Outputs from hipMalloc, hipExtMallocWithFlags(Contiguous / Uncached / Finegrained), torch, and hipMalloc
of larger blocks (1 GiB, output at its start), all in one process; eight inputs cycle.   usage (GPU box): tools/addr_alloc.py"""
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
eng = S.Engine(fft_size=1024, hop=256)


def raw(kind, size):
    p = C.c_void_p()
    if kind == "hipMalloc":
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(size))
    else:
        flag = {"contiguous": 0x4, "uncached": 0x3, "finegrained": 0x1}[kind]
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(size), C.c_uint(flag))
    return (p.value if rc == 0 else None), rc


bufs = []   # (label, device pointer)
for kind, size, n in (("hipMalloc", OUT_B, 6), ("contiguous", OUT_B, 4), ("uncached", OUT_B, 2), ("finegrained", OUT_B, 2),
                      ("hipMalloc", 1 << 30, 3), ("contiguous", 1 << 30, 2)):
    for i in range(n):
        p, rc = raw(kind, size)
        if p is None:
            print("%s of %d bytes: error %d" % (kind, size, rc))
            continue
        bufs.append(("%s %s" % (kind, "1 GiB" if size == 1 << 30 else "exact"), p))
tt = [torch.empty(1, FR, BINS, device=dev) for _ in range(4)]
bufs += [("torch", t.data_ptr()) for t in tt]
stream = torch.cuda.current_stream(dev).cuda_stream


def launch(i, ptr):
    x = ins[i % K]
    eng.stft_device_ptr(x.data_ptr(), 1, NS, NS, 1, ptr, FR * BINS, stream)


for i in range(900):
    launch(i, bufs[i % len(bufs)][1])
torch.cuda.synchronize()
REPS = 14
n = len(bufs) * REPS
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    launch(i, bufs[i % len(bufs)][1])
    ev[i + 1].record()
torch.cuda.synchronize()
acc = [[] for _ in bufs]
for i in range(n):
    acc[i % len(bufs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
for (label, p), a in zip(bufs, acc):
    a.sort()
    print("%-22s %#x   median %.1f us  (min %.1f)" % (label, p, a[len(a) // 2], a[0]))
eng.close()
