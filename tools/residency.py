#!/usr/bin/env python3
"""Round 6: how many waves of stft_kernel<11,2,0> are resident per SIMD, and at what clock -- from the kernel's own per-run
stamps (diagnostic build -DSG_STAMP=1: s_memrealtime and s_memtime at the start and end of every run), to settle what
profiles/r06_n2048_stalls.txt left open (the stall probes' wave lifetimes added up to 2.7 waves per SIMD, SQ_WAVE_CYCLES to 3.65).

  CPU box:  SG_VARIANT_API=1 tools/build_variant.sh stamp "-DSG_STAMP=1" 11     -> gpurun_stage/libspectro_stamp.so
  GPU box:  tools/residency.py [channels]
"""
import bisect, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import spectrogram_amd as S
ch = int(sys.argv[1])
x = torch.rand(ch, 1_440_000, device="cuda") * 2 - 1
f = os.environ.pop("SG_STAMP_FILE")
with S.Engine(fft_size=2048, hop=512) as e:
    o = e.stft(x)
    for _ in range(300):            # settle the clock
        e.stft(x, out=o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        e.stft(x, out=o)
    b.record(); torch.cuda.synchronize()
    print("MS %%.5f" %% (a.elapsed_time(b) / 20))
    os.environ["SG_STAMP_FILE"] = f
    e.stft(x, out=o)
    torch.cuda.synchronize()
    print("LAUNCH %%d %%d %%d %%d" %% tuple(e.last_launch()))
''' % ROOT


def main():
    ch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    path = "/tmp/residency_%d.bin" % ch
    env = dict(os.environ, SPECTRO_LIB=os.path.join(ROOT, "gpurun_stage", "libspectro_stamp.so"), SG_STAMP_FILE=path)
    out = subprocess.run([sys.executable, "-c", CHILD, str(ch)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if out.returncode:
        sys.exit("child failed:\n" + out.stderr[-2000:])
    ms = float([l for l in out.stdout.splitlines() if l.startswith("MS")][0].split()[1])
    run_len = int([l for l in out.stdout.splitlines() if l.startswith("LAUNCH")][0].split()[2])
    raw = open(path, "rb").read()
    v = struct.unpack("<%dQ" % (len(raw) // 8), raw)
    r0, r1, c0, c1 = v[0::4], v[1::4], v[2::4], v[3::4]     # per run: 100 MHz clock start / end, shader clock start / end
    n = len(r0)
    base = min(r0)
    total_us = (max(r1) - base) / 100.0
    dur_us = [(b - a) / 100.0 for a, b in zip(r0, r1)]
    clk = sorted((d - c) / max(1, (b - a)) * 100.0 for a, b, c, d in zip(r0, r1, c0, c1))      # MHz per run
    slots = 1024 * 4 // 2                                   # 4 waves per SIMD, 2 waves per run (one 128-lane workgroup)
    print("N = 2048 hop 512, %d channels: %d runs (workgroups of two waves), %.4f ms per launch by HIP events (stamp build)" % (ch, n, ms))
    print("the stamped launch: %.1f us from the first run's start to the last run's end; run duration p5 / p50 / p95 = %.1f / %.1f / %.1f us"
          % (total_us, sorted(dur_us)[n // 20], sorted(dur_us)[n // 2], sorted(dur_us)[-max(1, n // 20)]))
    print("shader clock per run (s_memtime ticks per s_memrealtime tick x 100 MHz): p5 / p50 / p95 = %.0f / %.0f / %.0f MHz"
          % (clk[n // 20], clk[n // 2], clk[-max(1, n // 20)]))
    res_avg = sum(dur_us) / total_us
    print("resident runs, time average: %.0f of %d slots = %.2f waves per SIMD (4 allowed by 128 VGPRs and 13.3 KB of LDS)"
          % (res_avg, slots, res_avg * 2 / 1024))
    start, end = sorted((t - base) / 100.0 for t in r0), sorted((t - base) / 100.0 for t in r1)
    step = max(10, int(total_us / 24))
    line = []
    for t in range(0, int(total_us) + step, step):
        line.append("%d:%d" % (t, bisect.bisect_right(start, t) - bisect.bisect_right(end, t)))
    print("resident runs at t us: " + " ".join(line))
    # the steady part of the launch (between the end of the ramp and the start of the drain): residency there
    t_lo, t_hi = 0.15 * total_us, 0.75 * total_us
    mid = sum(max(0.0, min(b, t_hi) - max(a, t_lo)) for a, b in zip(((t - base) / 100.0 for t in r0), ((t - base) / 100.0 for t in r1)))
    print("resident runs between 15 %% and 75 %% of the launch: %.0f = %.2f waves per SIMD" % (mid / (t_hi - t_lo), mid / (t_hi - t_lo) * 2 / 1024))
    cyc = sorted((d - c) for c, d in zip(c0, c1))
    print("shader cycles per run p50: %d (= %.0f per frame at %d frames per run)" % (cyc[n // 2], cyc[n // 2] / float(run_len), run_len))


if __name__ == "__main__":
    main()
