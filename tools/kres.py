#!/usr/bin/env python3
"""Compile one size's kernel unit for gfx950 and print each instantiation's register / LDS / scratch use
(hipcc -Rpass-analysis=kernel-resource-usage).  usage: tools/kres.py <log2n> [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
l2 = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       "-DSG_LOG2N=" + l2, "-mllvm", "-structurizecfg-skip-uniform-regions", "-c", os.path.join(ROOT, "spectrogram_amd/csrc/stft_inst.hip"), "-o", "/tmp/kres_%s.o" % l2,
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
if r.returncode:
    print(r.stdout[-4000:]); sys.exit(1)
cur = {}
rows = []
for line in r.stdout.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for c in rows:
    m = re.search(r"stft_kernelILi(\d+)ELi(\d+)ELi(\d+)E", c["name"])
    nm = "<%s,%s,%s>" % m.groups() if m else c["name"]
    print("%-12s vgpr=%-4s agpr=%-3s sgpr=%-3s scratch=%-4s occ=%-2s lds=%s" % (
        nm, c.get("VGPRs"), c.get("AGPRs"), c.get("TotalSGPRs"), c.get("ScratchSize [bytes/lane]"),
        c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
