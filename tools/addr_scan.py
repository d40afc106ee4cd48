#!/usr/bin/env python3
"""Round 6: the stores-only skeleton (4,017 streams x 28 rows of 2 KiB) at successive offsets of large allocations: where in an
allocation is memory of the fast kind?   usage (GPU box): tools/addr_scan.py [GiB per allocation] [allocations]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libstreams2.so",
                       os.path.join(ROOT, "tools", "streams_lib2.hip")])
import torch
GIB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NALLOC = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
sb = C.CDLL("/tmp/libstreams2.so")
stream = torch.cuda.current_stream(dev).cuda_stream
OUT_B = 4017 * 28 * 2048
STEP = 256 << 20
arenas = [torch.empty(GIB << 30, dtype=torch.uint8, device=dev) for _ in range(NALLOC)]
small = [torch.empty(OUT_B, dtype=torch.uint8, device=dev) for _ in range(12)]


def t_at(ptrs, reps=9):
    n = len(ptrs) * reps
    for i in range(len(ptrs)):
        sb.sb2_launch(0, C.c_void_p(ptrs[i]), 4017, 28, C.c_void_p(stream))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        sb.sb2_launch(0, C.c_void_p(ptrs[i % len(ptrs)]), 4017, 28, C.c_void_p(stream))
        ev[i + 1].record()
    torch.cuda.synchronize()
    acc = [[] for _ in ptrs]
    for i in range(n):
        acc[i % len(ptrs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
    return [sorted(a)[len(a) // 2] for a in acc]


# warm the clock
for _ in range(3):
    t_at([s.data_ptr() for s in small])
print("twelve separate allocations of %d bytes: %s" % (OUT_B, " ".join("%.1f" % v for v in t_at([s.data_ptr() for s in small]))))
for k, a in enumerate(arenas):
    ptrs = [a.data_ptr() + off for off in range(0, (GIB << 30) - OUT_B, STEP)]
    t = t_at(ptrs)
    print("allocation %d of %d GiB at %#x, every 256 MiB:" % (k, GIB, a.data_ptr()))
    for r in range(0, len(t), 16):
        print("   +%5d MiB: %s" % (r * 256, " ".join("%.1f" % v for v in t[r:r + 16])))
