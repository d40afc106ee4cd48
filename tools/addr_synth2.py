#!/usr/bin/env python3
"""Round 6: which store shapes of the 4,017-stream write skeleton does the output allocation's placement slow down?  (The STFT
launch and the stores-only skeleton agree on which allocations are fast: profiles/r06_addr_synth_split_box.txt.)
usage (GPU box): tools/addr_synth2.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libstreams2.so",
                       os.path.join(ROOT, "tools", "streams_lib2.hip")])
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
sb = C.CDLL("/tmp/libstreams2.so")
K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(12)]
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


def shape(k, n_streams, rows):
    def fn(i, o):
        rc = sb.sb2_launch(k, C.c_void_p(o.data_ptr()), n_streams, rows, C.c_void_p(stream))
        assert rc == 0, rc
    return fn


for i in range(800):
    eng.stft(ins[i % K], out=outs[i % len(outs)])
torch.cuda.synchronize()
rows = [("STFT launch", per_buffer(lambda i, o: eng.stft(ins[i % K], out=o)))]
for label, k in (("4 B / lane, non-temporal (the STFT's)", 0), ("4 B / lane, plain stores", 1), ("8 B / lane, non-temporal", 2),
                 ("16 B / lane, non-temporal", 3), ("16 B / lane, plain stores", 4), ("4 B / lane, nt, two rows per visit", 5),
                 ("16 B / lane, nt, two rows per visit", 6)):
    rows.append((label + ", 4017 x 28", per_buffer(shape(k, 4017, 28))))
rows.append(("16 B / lane, nt, 8034 x 14", per_buffer(shape(3, 8034, 14))))
rows.append(("torch fill_", per_buffer(lambda i, o: o.fill_(1.0))))
print("median us per launch into each of twelve output allocations")
for label, t in rows:
    print("%-46s %s   spread %.1f %%" % (label, " ".join("%5.1f" % v for v in t), 100 * (max(t) - min(t)) / min(t)))
eng.close()
