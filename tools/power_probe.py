#!/usr/bin/env python3
"""Clock and power while the STFT kernel runs back to back for a few seconds (rocm-smi sampled from a side thread).
usage: tools/power_probe.py [out_kind] [fft] [hop] [channels] [samples]"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S

out = sys.argv[1] if len(sys.argv) > 1 else "power"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
hop = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ch = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
ns = int(sys.argv[5]) if len(sys.argv) > 5 else 160_000
stop = False
samples = []

def smi():
    while not stop:
        t = time.time()
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "fclk", "junction", "Temperature (Sensor memory"))]
        samples.append((t, keep))
        time.sleep(0.3)

if out == "stream":   # the box's stream kernel (bench.py's roofline denominator) under the same sampling
    import ctypes as C
    lib = C.CDLL(os.path.join(os.path.dirname(S.__file__), "lib", "libsgdiag.so"))
    lib.sg_diag_stream_rate.restype = C.c_int
    lib.sg_diag_stream_rate.argtypes = [C.c_longlong, C.c_int, C.POINTER(C.c_double)]
    torch.cuda.init(); torch.zeros(1, device="cuda")
    th = threading.Thread(target=smi); th.start()
    time.sleep(1.0)
    t0 = time.time()
    g = C.c_double(0.0)
    rates = []
    for rep in range(6):
        lib.sg_diag_stream_rate(3 << 30, 400, C.byref(g))
        rates.append((time.time() - t0, g.value))
    time.sleep(1.0)
    stop = True; th.join()
    print("stream_mix<1,2> GB/s by batch:", " ".join("%.1fs:%.0f" % r for r in rates))
    for t, keep in samples:
        print("%6.2f  %s" % (t - t0, " | ".join(k.split(":", 1)[-1].strip() if "GPU[0]" in k else "" for k in keep)))
    sys.exit(0)

x = torch.rand(ch, ns, device="cuda") * 2 - 1
with S.Engine(fft_size=n, hop=hop, out=out) as e:
    o = e.stft(x)
    fr = ch * e.num_frames(ns)
    torch.cuda.synchronize()
    th = threading.Thread(target=smi); th.start()
    time.sleep(1.0)
    t0 = time.time()
    marks = []
    for rep in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            e.stft(x, out=o)
        b.record(); torch.cuda.synchronize()
        marks.append((time.time() - t0, a.elapsed_time(b) / 200))
    time.sleep(1.0)
    stop = True; th.join()
print("load starts at t=0; per-launch ms by batch:", " ".join("%.1fs:%.4f" % m for m in marks))
print("frames/s last batch: %.4g" % (fr / marks[-1][1] * 1e3))
for t, keep in samples:
    print("%6.2f  %s" % (t - t0, " | ".join(k.split(":", 1)[-1].strip() if "GPU[0]" in k else "" for k in keep)))
