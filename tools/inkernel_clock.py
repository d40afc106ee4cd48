#!/usr/bin/env python3
"""The clock the chip really holds under each kernel (MI355X_MICROARCH.md, DVFS give-back item 6): diagnostic build
-DSG_STAMP=1 (gpurun_stage/libspectro_stamp.so, built beforehand with tools/inkernel_clock.py --build), >= 2 s of
back-to-back launches, then one launch whose runs stamp s_memtime (shader clock) and s_memrealtime (100 MHz) at their
start and end: clock = median over runs of d(s_memtime) / d(s_memrealtime) x 100 MHz.  Beside it: the average launch
duration of the 2 s run (HIP events), rocm-smi's sclk / board power sampled during the run (for comparison only --
the guide: "board power and pp_dpm_sclk are not the test"), and the same for the bare stream kernel of libsgdiag.so."""
import json, os, re, struct, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP_LIB = os.path.join(ROOT, "gpurun_stage", "libspectro_stamp.so")

if "--build" in sys.argv:
    e = dict(os.environ, SG_EXTRA_HIPCC_FLAGS="-DSG_STAMP=1")
    keep = os.path.join(ROOT, "spectrogram_amd", "lib", "libspectro.so")
    os.makedirs(os.path.dirname(STAMP_LIB), exist_ok=True)
    os.rename(keep, keep + ".keep")
    try:
        subprocess.check_call([sys.executable, "-m", "spectrogram_amd.build", "--force"], cwd=ROOT, env=e, stdout=subprocess.DEVNULL)
        os.replace(keep, STAMP_LIB)
    finally:
        os.rename(keep + ".keep", keep)
    subprocess.check_call([sys.executable, "-m", "spectrogram_amd.build", "--force"], cwd=ROOT, stdout=subprocess.DEVNULL)
    print("built", STAMP_LIB)
    sys.exit(0)

CHILD = r'''
import json, os, re, struct, subprocess, sys, threading, time
sys.path.insert(0, %r)
import torch
import spectrogram_amd as S
n, hop, ch, ns, out, secs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], float(sys.argv[6])
stamp_file = os.environ.pop("SG_STAMP_FILE")
x = torch.rand(ch, ns, device="cuda") * 2 - 1
smi = []
stop = []
def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20).stdout
            pw = re.search(r"Power \(W\): ([\d.]+)", r); ck = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r)
            if pw and ck: smi.append((float(pw.group(1)), int(ck.group(1))))
        except Exception:
            return
with S.Engine(fft_size=n, hop=hop, out=out) as e:
    o = e.stft(x); torch.cuda.synchronize()
    th = threading.Thread(target=sampler); th.start()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); launches = 0
    a.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(64): e.stft(x, out=o)
        launches += 64
        torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / launches
    os.environ["SG_STAMP_FILE"] = stamp_file          # the next launch stamps its runs
    e.stft(x, out=o); torch.cuda.synchronize()
    del os.environ["SG_STAMP_FILE"]
    info = e.last_launch()
    stop.append(1); th.join()
raw = open(stamp_file, "rb").read()
v = struct.unpack("<%%dQ" %% (len(raw) // 8), raw)
r = sorted((v[i + 3] - v[i + 2]) / (v[i + 1] - v[i]) * 100.0 for i in range(0, len(v), 4) if v[i + 1] - v[i] > 50)
frames = ch * e.num_frames(ns) if False else None
pw = sorted(s[0] for s in smi[len(smi) // 3:]); ck = sorted(s[1] for s in smi[len(smi) // 3:])
print(json.dumps({"fft": n, "hop": hop, "channels": ch, "out": out, "kernel_variant": info[0], "run_len": info[1], "runs_stamped": len(r),
                  "inkernel_clock_MHz": {"median": r[len(r) // 2], "p5": r[len(r) // 20], "p95": r[-max(1, len(r) // 20)]},
                  "avg_launch_ms": ms, "launches": launches,
                  "rocm_smi_sclk_MHz": ck[len(ck) // 2] if ck else None, "rocm_smi_board_W": pw[len(pw) // 2] if pw else None}))
''' % ROOT

def main():
    secs = 2.2
    cases = [(1024, 256, 1, 28_800_000, "power"), (1024, 256, 1, 28_800_000, "db_u8"), (1024, 256, 4096, 160_000, "power"),
             (2048, 512, 256, 1_440_000, "power"), (2048, 512, 256, 1_440_000, "db_u8"), (1024, 300, 64, 1_600_000, "power"),
             (2048, 735, 64, 2_646_000, "db_u8")]
    for c in cases:
        env = dict(os.environ, SPECTRO_LIB=STAMP_LIB, SG_STAMP_FILE="/tmp/stamps_clock.bin")
        out = subprocess.run([sys.executable, "-c", CHILD] + [str(v) for v in c] + [str(secs)], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines else json.dumps({"case": c, "error": out.stderr[-400:]}), flush=True)
    # the bare stream kernel (roofline denominator): stamps are always compiled into libsgdiag.so
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "spectrogram_amd", "lib", "libsgdiag.so"))
    lib.sg_diag_stream_clock.restype = C.c_int
    lib.sg_diag_stream_clock.argtypes = [C.c_longlong, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    mhz, gbps = C.c_double(0), C.c_double(0)
    rc = lib.sg_diag_stream_clock(3 << 30, secs, C.byref(mhz), C.byref(gbps))
    print(json.dumps({"kernel": "libsgdiag stream_mix<1,2> (1 KiB read + 2 KiB written per unit, 3 GiB)", "rc": rc,
                      "inkernel_clock_MHz": {"median": mhz.value}, "GBps": gbps.value}), flush=True)

if __name__ == "__main__":
    main()
