#!/usr/bin/env python3
"""Board power and core clock while ONE instruction type runs back to back on every SIMD (tools/valubench <op> <secs>):
dynamic energy per wave-instruction = (P_load - P_idle) / rate.  Diagnostic; DESIGN.md section 5.5 uses the table."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def smi():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    p = re.search(r"Power \(W\): ([\d.]+)", r)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r)
    return (float(p.group(1)) if p else float("nan"), int(c.group(1)) if c else 0)

def main():
    ops = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [19, 5, 4, 3, 1, 0, 9, 12, 13, 18, 20, 21]
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
    time.sleep(1.0)
    idle = [smi() for _ in range(3)]
    p_idle = sorted(x[0] for x in idle)[1]
    print("idle: %.0f W, sclk %d MHz" % (p_idle, idle[-1][1]))
    for op in ops:
        samples = []
        pr = subprocess.Popen([os.path.join(ROOT, "tools", "valubench"), str(op), str(secs)], stdout=subprocess.PIPE, text=True)
        t0 = time.time()
        while pr.poll() is None:
            s = smi()
            if time.time() - t0 > 1.0:
                samples.append(s)
        out = pr.stdout.read().strip()
        m = re.search(r"([\d.e+]+) wave-instr/s", out)
        rate = float(m.group(1)) if m else float("nan")
        if samples:
            pw = sorted(x[0] for x in samples)[len(samples) // 2]
            ck = sorted(x[1] for x in samples)[len(samples) // 2]
            print("%-44s P %.0f W  sclk %d MHz  %.3g wave-instr/s  dynamic %.2f nJ per wave-instr (%.1f pJ per lane)" %
                  (out.split(":")[0], pw, ck, rate, (pw - p_idle) / rate * 1e9, (pw - p_idle) / rate / 64 * 1e12), flush=True)
        time.sleep(1.0)

if __name__ == "__main__":
    main()
