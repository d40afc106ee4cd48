#!/usr/bin/env python3
"""Round 6: does the config-2 launch's duration depend on WHERE its buffers lie?  The kernel trace of the round's profile
(profiles/r06_launch_trace_3sets.txt) shows the three rotating buffer sets of bench.py at 74.5 / 78.2 / 81.4 us per launch,
set by set, on one box.  Same process, per-launch HIP events:
  A  three inputs x four outputs from torch's allocator, all twelve pairings interleaved: does the time follow the
     input buffer, the output buffer, or the pair?
  B  the same buffers carved out of one arena at chosen offsets: output base moved by 0 / 4 KiB / 64 KiB / 1 MiB / ...
     against inputs at fixed 256 MiB-aligned places.
usage (GPU box): tools/addr_ab.py [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
eng = S.Engine(fft_size=1024, hop=256)


def settle(x, o, n=700):
    for i in range(n):
        eng.stft(x[i % len(x)], out=o[i % len(o)])
    torch.cuda.synchronize()


def measure(pairs, reps):
    """pairs: list of (x, o); interleaved round robin; returns mean / min us per pair (per-launch events)"""
    acc = [[] for _ in pairs]
    n = len(pairs) * reps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        x, o = pairs[i % len(pairs)]
        eng.stft(x, out=o)
        ev[i + 1].record()
    torch.cuda.synchronize()
    for i in range(n):
        acc[i % len(pairs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    return [(sum(a) / len(a), min(a)) for a in acc]


def view_f32(arena, off_bytes, shape):
    n = 1
    for s in shape:
        n *= s
    return arena[off_bytes:off_bytes + 4 * n].view(torch.float32).view(*shape)


# ---- A: allocator buffers, all pairings
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(3)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(4)]
settle(ins, outs)
pairs = [(ins[i], outs[j]) for j in range(4) for i in range(3)]
order = list(range(12))
res = measure(pairs, REPS)
print("A: torch allocator; input addresses %s, output addresses %s" % ([hex(t.data_ptr()) for t in ins], [hex(t.data_ptr()) for t in outs]))
for j in range(4):
    print("   out %d: " % j + "  ".join("in %d %.2f (min %.2f)" % (i, *res[j * 3 + i]) for i in range(3)))
by_in = [sum(res[j * 3 + i][0] for j in range(4)) / 4 for i in range(3)]
by_out = [sum(res[j * 3 + i][0] for i in range(3)) / 3 for j in range(4)]
print("   mean by input %s   mean by output %s" % (["%.2f" % v for v in by_in], ["%.2f" % v for v in by_out]))
del pairs, outs
torch.cuda.empty_cache()

# ---- B: one arena, outputs at chosen offsets
MiB = 1 << 20
arena = torch.empty(6 * 1024 * MiB, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
pad = (-base) % (256 * MiB)                                  # work relative to a 256 MiB boundary
in_off = [pad + k * 256 * MiB for k in range(3)]             # 115 MB each
ains = []
for k, off in enumerate(in_off):
    v = view_f32(arena, off, (1, NS))
    v.copy_(ins[k])
    ains.append(v)
out_bytes = FR * BINS * 4
deltas = [0, 256, 4096, 65536, MiB, MiB + 4096, 2 * MiB, 3 * MiB + 65536, 17 * MiB, 64 * MiB + 4096, 128 * MiB]
print("B: arena at %s (+%d to the 256 MiB boundary); inputs at +0, +256, +512 MiB; outputs of %d bytes" % (hex(base), pad, out_bytes))
for d in deltas:
    o_off = [pad + (3 + k) * 256 * MiB * 1 + d for k in range(3)]       # three outputs 256 MiB apart, all moved by d
    o_off = [pad + 768 * MiB + k * 512 * MiB + d for k in range(3)]
    aouts = [view_f32(arena, off, (1, FR, BINS)) for off in o_off]
    settle(ains, aouts, 200)
    r = measure([(ains[k], aouts[k]) for k in range(3)], REPS * 2)
    print("   output base = 512 MiB multiple + %-10d: %s   mean %.2f us" % (d, "  ".join("%.2f" % v[0] for v in r), sum(v[0] for v in r) / 3))
# inputs moved instead, outputs fixed
for d in [0, 4096, 65536, MiB, 3 * MiB + 65536]:
    a2 = []
    for k in range(3):
        v = view_f32(arena, pad + k * 256 * MiB + d, (1, NS))
        v.copy_(ins[k])
        a2.append(v)
    aouts = [view_f32(arena, pad + 768 * MiB + k * 512 * MiB, (1, FR, BINS)) for k in range(3)]
    settle(a2, aouts, 200)
    r = measure([(a2[k], aouts[k]) for k in range(3)], REPS * 2)
    print("   input base = 256 MiB multiple + %-10d: %s   mean %.2f us" % (d, "  ".join("%.2f" % v[0] for v in r), sum(v[0] for v in r) / 3))
eng.close()
