#!/usr/bin/env python3
"""Does bench.py's one-wave clock probe (libsgdiag.so sg_diag_clock_probe) see the clock the bench kernel runs at?
Idle, beside the float32 / byte-output STFT launches, at several points of a long queue; host time of the probe's
return against the queue's end shows whether it really ran beside the launches."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
import bench

lib = C.CDLL(os.path.join("spectrogram_amd", "lib", "libsgdiag.so"))
lib.sg_diag_clock_probe.restype = C.c_int
lib.sg_diag_clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]


lib.sg_diag_realtime_stamp.restype = C.c_int
lib.sg_diag_realtime_stamp.argtypes = [C.c_void_p, C.c_void_p]
stamps = torch.zeros(2, dtype=torch.int64, device="cuda")


def stamp(i):
    lib.sg_diag_realtime_stamp(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(stamps[i:].data_ptr()))


def probe(ev, segs=8, us=25, waves=1):
    m = (C.c_double * 13)()
    rc = lib.sg_diag_clock_probe(C.c_void_p(ev.cuda_event) if ev is not None else None, waves, segs, us, m)
    return rc, [round(v, 1) for v in m[:3]], (m[3], m[4])


x = torch.rand(1, bench.N_SAMPLES, device="cuda") * 2 - 1
print("idle", probe(None), flush=True)
for out in ("power", "db_u8"):
    with S.Engine(fft_size=1024, hop=256, out=out) as e:
        o = e.stft(x)
        for _ in range(700):          # ~50 ms: governor settled
            e.stft(x, out=o)
        torch.cuda.synchronize()
        for before, after in ((12, 48), (100, 100), (300, 100)):
            for _ in range(300):
                e.stft(x, out=o)
            ev = torch.cuda.Event()
            t0 = time.perf_counter()
            for _ in range(before):
                e.stft(x, out=o)
            stamp(0)
            ev.record()
            for _ in range(after):
                e.stft(x, out=o)
            stamp(1)
            t1 = time.perf_counter()
            r = probe(ev)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            q0, q1 = (int(v) for v in stamps.cpu())
            print(out, "probe behind launch %d of %d:" % (before, before + after), r[:2],
                  "probe ran %.0f .. %.0f us after the event; the %d launches behind it took %.0f us;" % ((r[2][0] - q0) / 100, (r[2][1] - q0) / 100, after, (q1 - q0) / 100),
                  "enqueue %.2f ms, probe returned at %.2f ms, queue drained at %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t0) * 1e3), flush=True)
        # longer segments
        for _ in range(400):
            e.stft(x, out=o)
        ev = torch.cuda.Event(); ev.record()
        for _ in range(200):
            e.stft(x, out=o)
        print(out, "8 x 200 us:", probe(ev, 8, 200), flush=True)
        torch.cuda.synchronize()
        for _ in range(400):
            e.stft(x, out=o)
        ev = torch.cuda.Event(); ev.record()
        for _ in range(200):
            e.stft(x, out=o)
        print(out, "64 waves, 8 x 25 us:", probe(ev, 8, 25, 64), flush=True)
        torch.cuda.synchronize()
