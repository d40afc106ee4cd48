#!/usr/bin/env python3
"""Same-PROCESS A/B of builds of libspectro.so on the SAME buffers (round 6: a launch's duration depends on where the driver
put its output allocation by 5-10 %, so builds compared in processes of their own -- tools/lib_ab.py -- carry that lottery
into the comparison).  Every library is dlopen'ed side by side (their symbols stay local), gets a handle of its own, and the
launches alternate build by build over the same rotating buffer sets.
    tools/lib_ab_same.py [--shape n,hop,channels,samples[,out]]... <name>=<lib.so|in-tree> ...
default shapes: the bench launch (config 2), a 4096-utterance batch, the same with byte output"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from spectrogram_amd import _lib as L
from spectrogram_amd.engine import make_config

args = sys.argv[1:]
per_set = "--per-set" in args
args = [a for a in args if a != "--per-set"]
shapes = []
while args and args[0] == "--shape":
    f = args[1].split(",")
    shapes.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4] if len(f) > 4 else "power"))
    args = args[2:]
if not shapes:
    shapes = [(1024, 256, 1, 28_800_000, "power"), (1024, 256, 4096, 160_000, "power"), (1024, 256, 4096, 160_000, "db_u8")]
L.load()                                               # the in-tree library (and the runtime policy) first
libs = []
for a in args:
    name, path = a.split("=", 1)
    path = L.LIB_PATH if path == "in-tree" else os.path.abspath(path)
    lib = C.CDLL(path)
    for sym, (res, at) in L.SYMBOLS.items():
        fn = getattr(lib, sym)
        fn.restype, fn.argtypes = res, at
    libs.append((name, lib))
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
for n, hop, ch, ns, out in shapes:
    frames = (ns - n) // hop + 1
    sets = 8 if ch * ns * 4 < (1 << 30) else 2
    xs = [(torch.rand(ch, ns, device=dev) * 2 - 1) for _ in range(sets)]
    os_ = [torch.empty(ch, frames, n // 2, device=dev, dtype=torch.uint8 if out == "db_u8" else torch.float32) for _ in range(sets)]
    hs = []
    for name, lib in libs:
        cfg = make_config(fft_size=n, hop=hop, out=out)
        h = C.c_void_p()
        L.check(lib.sg_create(C.byref(cfg), C.byref(h)))
        hs.append(h)

    def launch(k, i):
        rc = libs[k][1].sg_stft(hs[k], xs[i % sets].data_ptr(), ch, ns, ns, 1, os_[i % sets].data_ptr(), frames * (n // 2), L.PTR_DEVICE, stream)
        assert rc == 0, libs[k][1].sg_last_error()

    iters = max(8, min(240, int(0.02 / (ch * frames * 3072 / 4.5e12))))     # ~20 ms per block
    for k in range(len(libs)):
        for i in range(iters):
            launch(k, i)
    torch.cuda.synchronize()
    # bit-equality of the builds on this shape
    ref = None
    for k in range(len(libs)):
        launch(k, 0)
        torch.cuda.synchronize()
        cur = os_[0].clone()
        same = True if ref is None else bool(torch.equal(cur, ref))
        ref = cur if ref is None else ref
        if not same:
            print("   !! %s differs from %s on this shape" % (libs[k][0], libs[0][0]))
    res = [[] for _ in libs]
    for rnd in range(7):
        for k in (range(len(libs)) if rnd % 2 == 0 else reversed(range(len(libs)))):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(iters):
                launch(k, i)
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / iters * 1e3)
    print("N=%d hop=%d ch=%d samples=%d %s  (%d launches per block, %d buffer sets)" % (n, hop, ch, ns, out, iters, sets))
    base = sorted(res[0])[3]
    for (name, _), r in zip(libs, res):
        med = sorted(r)[3]
        print("   %-12s median %.2f us  (%+.2f %%)   blocks: %s" % (name, med, 100 * (base / med - 1), " ".join("%.2f" % v for v in r)))
    if per_set and sets > 2:
        # per buffer set (the output allocation decides 5-10 % of a launch): per-launch events, builds alternating pass by pass
        reps = 12
        seq = [(k, i) for r in range(reps) for k in range(len(libs)) for i in range(sets)]   # a build walks all sets: cold inputs
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
        ev[0].record()
        for j, (k, i) in enumerate(seq):
            launch(k, i)
            ev[j + 1].record()
        torch.cuda.synchronize()
        acc = {}
        for j, (k, i) in enumerate(seq):
            acc.setdefault((k, i), []).append(ev[j].elapsed_time(ev[j + 1]) * 1e3)
        print("   per buffer set (median of %d per-launch events, us):" % reps)
        for k, (name, _) in enumerate(libs):
            print("   %-12s %s" % (name, " ".join("%6.1f" % sorted(acc[(k, i)])[reps // 2] for i in range(sets))))
    for (name, lib), h in zip(libs, hs):
        lib.sg_destroy(h)
    del xs, os_
    torch.cuda.empty_cache()
