#!/usr/bin/env python3
"""Round 6, second look (tools/addr_ab.py: the launch's duration follows its OUTPUT allocation, 73 against 81 us, and not the
offset inside an allocation): many output allocations of several kinds, each timed against the same three rotating inputs.
usage (GPU box): tools/addr_ab2.py [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
OUT_B = FR * BINS * 4
dev = torch.device("cuda", 0)
eng = S.Engine(fft_size=1024, hop=256)
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(3)]
MiB = 1 << 20


def time_outputs(outs, reps=REPS):
    """every output against the three inputs in rotation; consecutive launches never share an output (Infinity Cache)"""
    n = len(outs) * reps
    for i in range(300):
        eng.stft(ins[i % 3], out=outs[i % len(outs)])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        eng.stft(ins[i % 3], out=outs[i % len(outs)])
        ev[i + 1].record()
    torch.cuda.synchronize()
    acc = [[] for _ in outs]
    for i in range(n):
        acc[i % len(outs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    return [sum(a) / len(a) for a in acc]


def show(tag, outs):
    t = time_outputs(outs)
    print("%s" % tag)
    for o, v in zip(outs, t):
        a = o.data_ptr()
        print("   %#x  (mod 1 GiB %4d MiB, mod 2 MiB %7d)  %.2f us" % (a, (a % (1 << 30)) // MiB, a % (2 * MiB), v))
    print("   fast (< 77 us): %d of %d; mean %.2f us" % (sum(v < 77 for v in t), len(t), sum(t) / len(t)))
    return t


print("inputs at %s" % [hex(t.data_ptr()) for t in ins])
# 1: twelve separate allocations from torch's allocator (one hipMalloc each at this size)
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(12)]
show("1: twelve torch allocations of %d bytes" % OUT_B, outs)
show("1b: the same twelve buffers, timed again", outs)
del outs
torch.cuda.empty_cache()
# 2: allocations padded to 256 MiB
raw = [torch.empty(256 * MiB, dtype=torch.uint8, device=dev) for _ in range(8)]
outs = [r[:OUT_B].view(torch.float32).view(1, FR, BINS) for r in raw]
show("2: eight allocations of exactly 256 MiB", outs)
del outs, raw
torch.cuda.empty_cache()
# 3: one 2 GiB allocation cut into eight 256 MiB slots; one 1 GiB into four
for gib in (2, 1):
    raw = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    outs = [raw[k * 256 * MiB:k * 256 * MiB + OUT_B].view(torch.float32).view(1, FR, BINS) for k in range(gib * 4)]
    show("3: one %d GiB allocation, 256 MiB slots" % gib, outs)
    del outs, raw
    torch.cuda.empty_cache()
# 4: small allocations: is it the size?  the output as FOUR separate quarter buffers cannot be one tensor; instead time outputs
#    whose allocation is just big enough, allocated AFTER 20 GB of other memory has been taken and released
junk = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(20)]
del junk
torch.cuda.empty_cache()
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(8)]
show("4: eight torch allocations after 20 GiB was taken and released", outs)
eng.close()
