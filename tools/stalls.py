#!/usr/bin/env python3
"""Round 6: where the waves of stft_kernel<11,2,0> (N = 2048, hop 512: config 3, the reference's fftSize) wait.

The SQ counters give the split of a wave's resident time into issuing / ready-but-not-picked / parked (SQ_ACTIVE_INST_ANY,
SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES: tools/n2048_stalls.sh) but not WHAT a parked wave waits for, and PC
sampling is refused on this pool.  So the kernel says it itself: diagnostic builds (-DSG_STALL=k, stft_kernel.hpp) put ONE
wait of the frame between two s_memtime stamps and every wave sums the interval over its run.  One probe per build, so a
frame carries two extra scalar memory reads and nothing else; build 10 brackets nothing (the stamp pair's own cost, taken
off every other probe), build 9 the whole frame.

  CPU box:  tools/stalls.py --build        -> gpurun_stage/libspectro_stall<k>.so   (k = 1 .. 10)
  GPU box:  tools/stalls.py [k ...]         one child process per build; prints the table profiles/r06_n2048_stalls.txt holds
"""
import json, os, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = {1: "head: s_waitcnt vmcnt(8) -- the frame's two PCM loads (and everything older: the stores of frame f-2)",
          2: "in-wave 8x8 transpose through LDS: 4 ds_write2_b64 + 4 ds_read2_b64 issued and landed (no barrier)",
          3: "wave-bit exchange: 8 ds_write_b64 issued and landed",
          4: "wave-bit exchange: s_barrier",
          5: "wave-bit exchange: 8 ds_read_b64 issued and landed",
          6: "partner exchange: 4 ds_write_b64 issued and landed",
          7: "partner exchange: s_barrier",
          8: "partner exchange: 4 ds_read_b64 issued and landed",
          9: "whole frame (top of the frame to behind its stores' issue)",
          10: "nothing (cost of the stamp pair itself)"}

CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
import spectrogram_amd as S
ch, ns = 256, 1_440_000
x = torch.rand(ch, ns, device="cuda") * 2 - 1
f = os.environ.pop("SG_STAMP_FILE", None)
with S.Engine(fft_size=2048, hop=512) as e:
    o = e.stft(x)
    for _ in range(300):            # settle the clock
        e.stft(x, out=o)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        e.stft(x, out=o)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 30
    info = e.last_launch()
    frames = ch * e.num_frames(ns)
    if f:
        os.environ["SG_STAMP_FILE"] = f
        e.stft(x, out=o)
        torch.cuda.synchronize()
print(json.dumps({"ms": ms, "frames": frames, "run_len": info[1], "total_runs": info[3]}))
''' % ROOT


def run(k):
    env = dict(os.environ)
    path = None
    if k is not None:
        env["SPECTRO_LIB"] = os.path.join(ROOT, "gpurun_stage", "libspectro_stall%d.so" % k)
        path = "/tmp/stall_%d.bin" % k
        env["SG_STAMP_FILE"] = path
    else:
        env.pop("SPECTRO_LIB", None)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    if out.returncode != 0:
        raise RuntimeError("probe %s failed:\n%s" % (k, out.stderr[-2000:]))
    res = json.loads(out.stdout.strip().splitlines()[-1])
    if path:
        raw = open(path, "rb").read()
        v = struct.unpack("<%dQ" % (len(raw) // 8), raw)
        res["total_cycles"] = sum(v[0::2])      # per wave: cycles of the run, cycles between the stamps
        res["probe_cycles"] = sum(v[1::2])
        res["waves"] = len(v) // 2
        per = sorted(p / t for t, p in zip(v[0::2], v[1::2]) if t)
        res["share_p5_p50_p95"] = [per[len(per) // 20], per[len(per) // 2], per[-max(1, len(per) // 20)]]
    return res


def main():
    if "--build" in sys.argv:
        for k in PROBES:
            subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), "stall%d" % k, "-DSG_STALL=%d" % k, "11"],
                                  env=dict(os.environ, SG_VARIANT_API="1"))
        return
    ks = [int(a) for a in sys.argv[1:]] or sorted(PROBES)
    base = run(None)
    print("product build: %.4f ms per launch of %d frames (%.3fe9 frames/s), runs of %d frames"
          % (base["ms"], base["frames"], base["frames"] / base["ms"] / 1e6, base["run_len"]))
    res = {}
    for k in ([10] if 10 not in ks else []) + ks:
        res[k] = run(k)
    fw = 2 * base["frames"]                       # frame-waves per launch (two waves per frame)
    pair = res[10]["probe_cycles"] / fw           # cycles one stamp pair costs by itself
    print("stamp pair alone: %.1f cycles per frame-wave (taken off every probe below); shader clock per run from the same stamps"
          % pair)
    print("%-3s %-9s %-10s %-12s %-12s %s" % ("k", "ms/launch", "slowdown", "cyc/frame-w", "share of run", "what is between the stamps"))
    for k in sorted(res):
        r = res[k]
        cyc = r["probe_cycles"] / fw - (pair if k != 10 else 0.0)
        tot = r["total_cycles"] / fw
        print("%-3d %-9.4f %-10.3f %-12.1f %-12.4f %s" % (k, r["ms"], r["ms"] / base["ms"], cyc, cyc / tot, PROBES[k]))
    print(json.dumps({"base": base, "probes": res}))


if __name__ == "__main__":
    main()
