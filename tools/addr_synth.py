#!/usr/bin/env python3
"""Round 6: do synthetic stream kernels see the fast / slow OUTPUT allocations the STFT launch sees (tools/addr_grid.py)?  Ten
output allocations; per allocation the STFT launch and three synthetic kernels of the same geometry (4,018 waves, 28 rows of
2 KiB each): stores only; stores + the 1 KiB of loads per row; the same with a frame's worth of arithmetic in between.
usage (GPU box): tools/addr_synth.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libstreams.so",
                       os.path.join(ROOT, "tools", "streams_lib.hip")])
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
sb = C.CDLL("/tmp/libstreams.so")
K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(10)]
eng = S.Engine(fft_size=1024, hop=256)
stream = torch.cuda.current_stream(dev).cuda_stream


def per_buffer(fn, reps=12):
    n = len(outs) * reps
    for i in range(2 * len(outs)):
        fn(i, outs[i % len(outs)])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(i, outs[i % len(outs)])
        ev[i + 1].record()
    torch.cuda.synchronize()
    acc = [[] for _ in outs]
    for i in range(n):
        acc[i % len(outs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    return [sorted(a)[len(a) // 2] for a in acc]


def synth(mode, n_streams, rows, work):
    def fn(i, o):
        rc = sb.sb_launch(mode, C.c_void_p(ins[i % K].data_ptr()), C.c_void_p(o.data_ptr()), n_streams, rows, work, C.c_void_p(stream))
        assert rc == 0, rc
    return fn


for i in range(800):
    eng.stft(ins[i % K], out=outs[i % len(outs)])
torch.cuda.synchronize()
rows = [("STFT launch (4,018 runs of 28 frames)", per_buffer(lambda i, o: eng.stft(ins[i % K], out=o)))]
for label, mode, n, r, w in (("stores only, 4017 x 28 rows", 0, 4017, 28, 0), ("loads + stores, 4017 x 28", 1, 4017, 28, 0),
                             ("loads + 60 FMA chains + stores", 2, 4017, 28, 60), ("loads + 200 FMA chains + stores", 2, 4017, 28, 200),
                             ("loads + stores, 8034 x 14", 1, 8034, 14, 0), ("loads + stores, 2008 x 56", 1, 2008, 56, 0)):
    rows.append((label, per_buffer(synth(mode, n, r, w))))
rows.append(("STFT launch again", per_buffer(lambda i, o: eng.stft(ins[i % K], out=o))))
print("median us per launch into each of ten output allocations")
for label, t in rows:
    print("%-38s %s   spread %.1f %%" % (label, " ".join("%6.1f" % v for v in t), 100 * (max(t) - min(t)) / min(t)))
eng.close()
