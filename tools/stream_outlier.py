#!/usr/bin/env python3
"""Where does config 4's one slow hop come from?  Each variant in a FRESH process (the slow hop shows once per
process): 0 / 20 / 40 / 200 untimed warm-up hops, and a 50 ms pause between context creation and the run."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
warm, pause = int(sys.argv[1]), float(sys.argv[2])
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
if pause: 
    # keep the GPU busy for `pause` seconds first (clocks up, every engine awake), then run
    x = torch.rand(1 << 26, device=dev); y = torch.empty_like(x).pin_memory()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < pause:
        y.copy_(x[: y.numel()], non_blocking=True); x.mul_(1.0001); torch.cuda.synchronize()
t_start = time.perf_counter()
r = bench.streaming_report(dev, hops=400, warm=warm)
print(json.dumps({"warm": warm, "busy_before_s": pause, "host_wall": r["host_wall"], "d2h": r["d2h"], "h2d": r["h2d"], "slowest": r["slowest_hops"][:2]}))
''' % ROOT
for warm, pause in ((20, 0), (0, 0), (40, 0), (200, 0), (20, 0.5)):
    out = subprocess.run([sys.executable, "-c", CHILD, str(warm), str(pause)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            j = json.loads(l)
            print(j["warm"], j["busy_before_s"], "wall max %.3f" % j["host_wall"]["max"], "d2h max %.3f h2d max %.3f" % (j["d2h"]["max"], j["h2d"]["max"]),
                  [(s["hop"], round(s["h2d"], 3), round(s["kernel"], 3), round(s["d2h"], 3)) for s in j["slowest"]], flush=True)
