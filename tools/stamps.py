#!/usr/bin/env python3
"""Timeline of ONE launch of the bench workload from per-run start / end stamps (diagnostic build -DSG_STAMP=1):
how many runs are resident over time, when the first round ends, how long the tail is.
usage (GPU box): tools/stamps.py [run_len ...]      rebuilds libspectro.so with the stamp code, restores it afterwards"""
import os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def build(flags):
    e = dict(os.environ); e["SG_EXTRA_HIPCC_FLAGS"] = flags
    subprocess.check_call([sys.executable, "-m", "spectrogram_amd.build", "--force"], cwd=ROOT, env=e, stdout=subprocess.DEVNULL)

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import spectrogram_amd as S
x = torch.rand(1, 28_800_000, device="cuda") * 2 - 1
f = os.environ.pop("SG_STAMP_FILE")
with S.Engine(fft_size=1024, hop=256) as e:
    o = e.stft(x)
    for _ in range(600):            # settle the clock (docs/EXPERIMENTS.md 5.1)
        e.stft(x, out=o)
    torch.cuda.synchronize()
    os.environ["SG_STAMP_FILE"] = f
    e.stft(x, out=o)
    torch.cuda.synchronize()
''' % ROOT

def one(run_len):
    path = "/tmp/stamps_%d.bin" % run_len
    e = dict(os.environ, SG_STAMP_FILE=path)
    if run_len:
        e["SG_RUN_LEN"] = str(run_len)
    subprocess.check_call([sys.executable, "-c", CHILD], env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    raw = open(path, "rb").read()
    v = struct.unpack("<%dQ" % (len(raw) // 8), raw)
    t0s, t1s = v[0::4], v[1::4]            # per run: realtime start / end, shader clock start / end
    base = min(t0s)
    us = lambda t: (t - base) / 100.0          # s_memrealtime: 100 MHz
    start = sorted(us(t) for t in t0s); end = sorted(us(t) for t in t1s)
    n = len(start); total = end[-1]
    dur = sorted((b - a) / 100.0 for a, b in zip(t0s, t1s))
    print("run_len=%d: %d runs, launch %.1f us (first start -> last end); run duration median %.1f us (p5 %.1f, p95 %.1f)"
          % (run_len, n, total, dur[n // 2], dur[n // 20], dur[-n // 20]))
    # resident runs over time
    import bisect
    line = []
    for t in range(0, int(total) + 5, 5):
        res = bisect.bisect_right(start, t) - bisect.bisect_right(end, t)
        line.append("%d:%d" % (t, res))
    print("  resident runs at t us: " + " ".join(line))
    # by XCD (the kernel gives each XCD a contiguous range of runs) and by slot within the workgroup
    nblocks = (n + 3) // 4
    per = nblocks >> 3
    d_us = [(b - a) / 100.0 for a, b in zip(t0s, t1s)]
    e_us = [us(b) for b in t1s]
    if per:
        parts = []
        for x in range(8):
            idx = range(x * per * 4, min((x + 1) * per * 4, n))
            dd = sorted(d_us[i] for i in idx); ee = max(e_us[i] for i in idx)
            parts.append("X%d med %.1f p95 %.1f last-end %.1f" % (x, dd[len(dd) // 2], dd[-max(1, len(dd) // 20)], ee))
        print("  per XCD run duration: " + " | ".join(parts))
    for sl in range(4):
        dd = sorted(d_us[i] for i in range(sl, n, 4))
        print("  slot %d: median %.1f p95 %.1f" % (sl, dd[len(dd) // 2], dd[-len(dd) // 20]), end=";")
    print()
    # duration against start time (first round vs later rounds)
    first = sorted(d_us[i] for i in range(n) if us(t0s[i]) < 5.0)
    later = sorted(d_us[i] for i in range(n) if us(t0s[i]) >= 5.0)
    if first and later:
        print("  runs started < 5 us: n=%d median %.1f; later: n=%d median %.1f" % (len(first), first[len(first) // 2], len(later), later[len(later) // 2]))
    print("  starts: first 4096 by %.1f us; last start %.1f us; ends: first %.1f us, 50%% %.1f, 90%% %.1f, 99%% %.1f, last %.1f"
          % (start[min(4095, n - 1)], start[-1], end[0], end[n // 2], end[int(n * 0.9)], end[int(n * 0.99)], end[-1]))

if __name__ == "__main__":
    lens = [int(a) for a in sys.argv[1:]] or [0]
    pre = os.path.join(ROOT, "gpurun_stage", "libspectro_stamp.so")   # tools/inkernel_clock.py --build (on the CPU box)
    if os.path.exists(pre) and not os.environ.get("SG_STAMP_FLAGS"):
        os.environ["SPECTRO_LIB"] = pre
        for rl in lens:
            one(rl)
        sys.exit(0)
    build("-DSG_STAMP=1 " + os.environ.get("SG_STAMP_FLAGS", ""))
    try:
        for rl in lens:
            one(rl)
    finally:
        build("")
