#!/usr/bin/env python3
"""Same-box, interleaved A/B of builds of libspectro.so (each build in a process of its own, rounds round-robin):
    tools/lib_ab.py <name>=<lib.so|in-tree> ...                    the bench launch (three rotating buffer pairs, as bench.py)
    tools/lib_ab.py --modes r04,ref <name>=<lib.so|in-tree> ...    tools/bench_sizes.py modes instead (the former ab_lib.sh)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
import spectrogram_amd as S
import bench
x = [torch.rand(1, bench.N_SAMPLES, device="cuda") * 2 - 1 for _ in range(bench.N_SETS)]
with S.Engine(fft_size=1024, hop=256, out=os.environ.get("AB_OUT", "power")) as e:
    o = [e.stft(x[0]).clone() for _ in range(bench.N_SETS)]
    i = 0
    for _ in range(800):
        e.stft(x[i %% bench.N_SETS], out=o[i %% bench.N_SETS]); i += 1
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(300):
        e.stft(x[i %% bench.N_SETS], out=o[i %% bench.N_SETS]); i += 1
    b.record(); torch.cuda.synchronize()
    print("%%.2f" %% (a.elapsed_time(b) / 300 * 1e3))
''' % ROOT
args = sys.argv[1:]
modes = []
if args and args[0] == "--modes":
    modes, args = args[1].split(","), args[2:]
libs = [a.split("=", 1) for a in args]
if modes:
    for rnd in range(2):
        for m in modes:
            for n, lib in libs:
                env = dict(os.environ)
                if lib != "in-tree":
                    env["SPECTRO_LIB"] = lib
                print("== %s %s" % (n, m), flush=True)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_sizes.py"), m], env=env, stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, text=True).stdout
                print("\n".join(l for l in r.splitlines() if l.startswith("{")), flush=True)
    sys.exit(0)
res = {n: [] for n, _ in libs}
for rnd in range(3):
    for n, lib in libs:
        env = dict(os.environ)
        if lib != "in-tree":
            env["SPECTRO_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.strip().splitlines()
        res[n].append(float(out[-1]) if out else float("nan"))
for n, _ in libs:
    print("%-12s %s  median %.2f us/launch" % (n, " ".join("%.2f" % v for v in res[n]), sorted(res[n])[1]), flush=True)
