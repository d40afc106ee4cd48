#!/usr/bin/env python3
"""PCIe-inclusive rate of sg_stft's host-pointer path on BASELINE config 2 (10 min mono 48 kHz,
N=1024, hop 256): H2D 115 MB + kernel + D2H 230 MB per call, pageable vs pinned host buffers."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spectrogram_amd as S
from spectrogram_amd import _lib as L

n, hop, ns = 1024, 256, 28_800_000
frames = (ns - n) // hop + 1
res = {}
with S.Engine(fft_size=n, hop=hop) as e:
    for kind in ("pageable", "pinned", "pinned_in_place"):
        if kind == "pageable":
            x = np.random.default_rng(0).uniform(-1, 1, ns).astype(np.float32)
            o = np.empty((frames, 512), np.float32)
            xp, op = x.ctypes.data, o.ctypes.data
        else:
            xt = (torch.rand(ns) * 2 - 1).pin_memory(); ot = torch.empty(frames, 512).pin_memory()
            xp, op = xt.data_ptr(), ot.data_ptr()
        if kind == "pinned_in_place":   # SG_IN_PINNED | SG_OUT_PINNED: the kernel reads / writes the page-locked buffers itself
            def call():
                L.check(e._lib.sg_stft(e._h, xp, 1, ns, ns, 1, op, frames * 512, L.IN_PINNED | L.OUT_PINNED, None))
                torch.cuda.synchronize()
        else:
            call = lambda: L.check(e._lib.sg_stft(e._h, xp, 1, ns, ns, 1, op, frames * 512, L.PTR_HOST, None))
        call(); call()
        t0 = time.perf_counter()
        for _ in range(5): call()
        dt = (time.perf_counter() - t0) / 5
        res[kind] = {"ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(frames / dt), "GBps_moved": round(345.6e6 / dt / 1e9, 2)}
# Page-locked INPUT read in place by the generic loads, or staged by an async copy first?  (the generic loads re-read
# the overlap between frames; if those re-reads cross PCIe, in place loses.)  sg_stft decides by shape (spectro_api.hip): in
# place for hop >= N and for N <= 512 at unit stride, staged otherwise; "pinned_direct_generic" forces in place for this A/B.
for n2, hop2, ns2 in ((512, 100, 8_000_000), (512, 16, 2_000_000), (256, 8, 1_000_000), (16384, 1000, 16_000_000), (2048, 2100, 16_000_000)):
    frames2 = (ns2 - n2) // hop2 + 1
    xt = (torch.rand(ns2) * 2 - 1).pin_memory()
    ot = torch.empty(frames2, n2 // 2).pin_memory()
    do = torch.empty(frames2, n2 // 2, device="cuda")
    for outk, op, oflag in (("pinned out", ot.data_ptr(), L.OUT_PINNED), ("device out", do.data_ptr(), L.OUT_DEVICE)):
        row = {}
        for label, dbg in (("library's choice", {}), ("staged", {"no_pinned_direct": 1}), ("in place", {"pinned_direct_generic": 1})):
            with S.Engine(fft_size=n2, hop=hop2, debug=dbg) as e2:
                def call():
                    L.check(e2._lib.sg_stft(e2._h, xt.data_ptr(), 1, ns2, ns2, 1, op, frames2 * (n2 // 2), L.IN_PINNED | oflag, None))
                    torch.cuda.synchronize()
                for _ in range(6): call()   # the first calls on freshly page-locked buffers pay for their mapping
                t0 = time.perf_counter()
                for _ in range(5): call()
                row[label] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        res["generic loads N=%d hop=%d (%.1fx overlap), %s" % (n2, hop2, n2 / hop2, outk)] = row
    del xt, ot, do
print(json.dumps(res))
