#!/usr/bin/env python3
"""One box's character in a minute: the bench launch (config 2) in its single-round geometry and in 16-frame runs, a
large batch, the byte-output variant and the bare stream kernel -- each with its per-launch time (HIP events over
back-to-back launches after 60 ms of the same load), the shader clock a one-wave probe sees beside it, and rocm-smi's
clocks / power / temperatures under that load.  Boxes of this pool run the same float32 launch in 65 us or in 80 us;
this is what tells them apart (profiles/r03_box_check.txt)."""
import ctypes as C, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import torch
import spectrogram_amd as S
import bench

lib = bench.clock_probe_lib()


def smi():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    d = {}
    pw = re.search(r"Power \(W\): ([\d.]+)", r)
    if pw: d["W"] = float(pw.group(1))
    for k, v in re.findall(r"(\w+) clock level: \d+: \((\d+)Mhz\)", r): d.setdefault(k, int(v))
    for k, v in re.findall(r"Temperature \(Sensor (\w+)\) \(C\): ([\d.]+)", r): d.setdefault("T_" + k, float(v))
    return d


def measure(tag, launch, frames, bytes_per_frame, load_s=1.2):
    for _ in range(3): launch()
    torch.cuda.synchronize()
    # rocm-smi beside ~1.2 s of load
    samples, stop = [], []
    def sampler():
        t0 = time.perf_counter()
        while not stop:
            s = smi()
            if time.perf_counter() - t0 > 0.4: samples.append(s)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < load_s:
        for _ in range(32): launch(); n += 1
        torch.cuda.synchronize()
    stop.append(1); th.join()
    # events over back-to-back launches, the probe beside them
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k = max(40, int(8e-3 / max((time.perf_counter() - t0) / n, 1e-6)))
    ev = torch.cuda.Event()
    a.record()
    h0 = time.perf_counter()
    for i in range(k):
        launch()
        if i == k // 4: ev.record()
    h1 = time.perf_counter()
    b.record()
    mhz = (C.c_double * 13)()
    rc = lib.sg_diag_clock_probe(C.c_void_p(ev.cuda_event), 16, 8, 25, mhz) if lib else -1
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / k * 1e3
    med = lambda key: (sorted(x[key] for x in samples if key in x) or [None])[len([x for x in samples if key in x]) // 2]
    print("%-34s %8.1f us/launch %7.1f Mframes/s %6.0f GB/s alg | probe %s MHz | smi sclk %s mclk %s fclk %s  %s W  Tj %s Thbm %s" % (
        tag, us, frames / us, frames * bytes_per_frame / us / 1e3, "%.0f [%s] host %.1f us/enqueue" % (mhz[0], " ".join("%.0f" % mhz[5 + x] for x in range(8)), (h1 - h0) / k * 1e6) if rc == 0 else "-",
        med("sclk"), med("mclk"), med("fclk"), med("W"), med("T_junction"), med("T_memory")), flush=True)


print("idle", smi(), flush=True)
x1 = torch.rand(1, bench.N_SAMPLES, device="cuda") * 2 - 1
for out, bpf in (("power", 3072), ("db_u8", 1536)):
    with S.Engine(fft_size=1024, hop=256, out=out) as e:
        o = e.stft(x1)
        measure("config 2 %s (run_len %d)" % (out, e.last_launch()[1]), lambda: e.stft(x1, out=o), bench.FRAMES, bpf)
os.environ["SG_NO_SINGLE_ROUND"] = "1"     # read when the handle is made
e16 = S.Engine(fft_size=1024, hop=256)
del os.environ["SG_NO_SINGLE_ROUND"]
with e16 as e:
    o = e.stft(x1)
    measure("config 2 power (run_len %d)" % e.last_launch()[1], lambda: e.stft(x1, out=o), bench.FRAMES, 3072)
# frames per run against the launch time: a regular stride between thousands of concurrent runs could alias on memory
# channels (a run writes run_len x 2 KiB, the next run starts right behind it)
if "--sweep" in sys.argv:
    for rl in (16, 20, 24, 26, 27, 28, 29, 30, 32, 36, 40, 48, 56, 64):
        os.environ["SG_RUN_LEN"] = str(rl)
        er = S.Engine(fft_size=1024, hop=256)
        del os.environ["SG_RUN_LEN"]
        with er as e:
            e.stft(x1, out=o)
            for _ in range(300): e.stft(x1, out=o)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(200): e.stft(x1, out=o)
            b.record(); torch.cuda.synchronize()
            print("  SG_RUN_LEN=%d: %.1f us/launch (%d runs)" % (rl, a.elapsed_time(b) / 200 * 1e3, e.last_launch()[3]), flush=True)
del o
xb = torch.rand(2048, 480_000, device="cuda") * 2 - 1
with S.Engine(fft_size=1024, hop=256) as e:
    o = e.stft(xb)
    fr = 2048 * e.num_frames(480_000)
    measure("2048 ch x 10 s power (run_len %d)" % e.last_launch()[1], lambda: e.stft(xb, out=o), fr, 3072)
del o, xb
x2 = torch.rand(64, 44100 * 64, device="cuda") * 2 - 1
with S.Engine(fft_size=2048, hop=735, out="db_u8", window="blackman", scale="inv_n") as e:
    o = e.stft(x2)
    fr = 64 * e.num_frames(44100 * 64)
    measure("N=2048 hop 735 bytes, 64 ch", lambda: e.stft(x2, out=o), fr, 735 * 4 + 1024)
g = bench.measured_stream_gbps()
print("stream_mix<1,2>: %.0f GB/s" % g, flush=True)
