#!/usr/bin/env python3
"""Per frame-wave instruction and cycle budget from a tools/pmc_sizes.sh summary (r03gen mode): instructions by class,
active-instruction cycles (SQ_ACTIVE_INST_* are quad-cycles), SIMD issue occupancy, LDS array cycles, HBM traffic
against the algorithmic bytes."""
import re, sys
t = open(sys.argv[1]).read()
# frames per launch and algorithmic bytes per frame of tools/bench_sizes.py r03gen, waves per frame
cases = {"<10,0,0>": (341120, 1, 300 * 4 + 2048), "<10,1,0>": (341120, 1, 300 * 4 + 2048), "<10,2,0>": (399808, 1, 3072),
         "<11,0,3>": (230272, 2, 735 * 4 + 1024), "<11,1,3>": (230272, 2, 735 * 4 + 1024), "<11,2,3>": (330496, 2, 512 * 4 + 1024)}
if len(sys.argv) > 2 and sys.argv[2] == "sizes":   # the default mode of tools/bench_sizes.py: (frames per launch, waves per frame, bytes per frame)
    cases = {"<10,2,0>": (2547712, 1, 3072), "<11,2,0>": (719104, 2, 6144), "<12,2,0>": (510464, 4, 12288), "<13,2,0>": (127808, 8, 24576),
             "<8,2,0>": (2399936, 0.25, 1024), "<9,2,0>": (1199808, 0.5, 1536), "<10,4,0>": (199936, 1, 4096), "<10,2,2>": (399808, 1, 3072),
             "<10,2,3>": (399808, 1, 1536), "<11,2,3>": (719104, 2, 3072)}
for k, (frames, wpf, bpf) in cases.items():
    m = re.search(r"stft_kernel" + re.escape(k) + r".*?(?=stft_kernel<|\Z)", t, re.S)
    if not m: continue
    b = m.group(0)
    g = lambda n: float(re.search(n + r"\s+mean=([\d.e+]+)", b).group(1))
    d = float(re.search(r"avg=([\d.]+) us", b).group(1))
    fw = frames * wpf
    rd, wr = g("FETCH_SIZE") * 2048, g("WRITE_SIZE") * 1024
    print("%s  %.1f us/launch = %.3ge9 frames/s | per frame-wave: VALU %.0f SALU %.0f LDS %.0f VMEM rd %.1f wr %.1f | active cycles %.0f "
          "(VALU %.0f, LDS %.0f) | SIMD issue busy %.2f | waves parked %.2f | LDS array cycles %.0f (conflicts %.0f) | HBM read %.3g + write %.3g "
          "= %.3fx algorithmic" % (k, d, frames / d / 1e3, g("SQ_INSTS_VALU") / fw, g("SQ_INSTS_SALU") / fw, g("SQ_INSTS_LDS") / fw,
          g("SQ_INSTS_VMEM_RD") / fw, g("SQ_INSTS_VMEM_WR") / fw, 4 * g("SQ_ACTIVE_INST_ANY") / fw, 4 * g("SQ_ACTIVE_INST_VALU") / fw,
          4 * g("SQ_ACTIVE_INST_LDS") / fw, 4 * g("SQ_ACTIVE_INST_ANY") / 1024 / (d * 2.0e3), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
          g("SQ_LDS_IDX_ACTIVE") / fw, g("SQ_LDS_BANK_CONFLICT") / fw, rd, wr, (rd + wr) / (frames * bpf)))
