#!/usr/bin/env python3
"""Round 6: the launch time of config 2 for every pairing of eight input with eight output allocations (inputs cycle through all
eight, so nothing of an input survives in the Infinity Cache): does what is left of the allocation dependence follow the input,
the output, or the pair?   usage (GPU box): tools/addr_grid.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
dev = torch.device("cuda", 0)
eng = S.Engine(fft_size=1024, hop=256)
K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]
outs = [torch.empty(1, FR, BINS, device=dev) for _ in range(K)]
for i in range(800):
    eng.stft(ins[i % K], out=outs[i % K])
torch.cuda.synchronize()
REPS = 12
# launch i: input i mod 8, output (i + i // 8) mod 8 -- every pairing once per 64 launches, inputs always cycling
seq = [(i % K, (i + i // K) % K) for i in range(K * K * REPS)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
ev[0].record()
for j, (a, b) in enumerate(seq):
    eng.stft(ins[a], out=outs[b])
    ev[j + 1].record()
torch.cuda.synchronize()
t = [[[] for _ in range(K)] for _ in range(K)]
for j, (a, b) in enumerate(seq):
    t[a][b].append(ev[j].elapsed_time(ev[j + 1]) * 1e3)
med = [[sorted(t[a][b])[len(t[a][b]) // 2] for b in range(K)] for a in range(K)]
print("inputs  " + " ".join(hex(x.data_ptr()) for x in ins))
print("outputs " + " ".join(hex(x.data_ptr()) for x in outs))
print("median us per launch; rows = input allocation, columns = output allocation")
for a in range(K):
    print("in %d: " % a + " ".join("%5.1f" % v for v in med[a]) + "   | mean %.2f" % (sum(med[a]) / K))
print("out mean " + " ".join("%5.1f" % (sum(med[a][b] for a in range(K)) / K) for b in range(K)))
g = sum(sum(r) for r in med) / (K * K)
ri = [sum(med[a]) / K - g for a in range(K)]
co = [sum(med[a][b] for a in range(K)) / K - g for b in range(K)]
res = max(abs(med[a][b] - g - ri[a] - co[b]) for a in range(K) for b in range(K))
print("grand mean %.2f; input effects %s; output effects %s; largest residual of the additive model %.2f us"
      % (g, " ".join("%+.1f" % v for v in ri), " ".join("%+.1f" % v for v in co), res))
eng.close()
