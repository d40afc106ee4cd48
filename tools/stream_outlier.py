#!/usr/bin/env python3
"""Where does config 4's one slow hop come from?  Each variant in a FRESH process (the slow hop shows once per
process): untimed warm-up hops 0 / 20 / 200; no per-hop HIP events at all (host wall clock only); and a 64-hop
throwaway pass with events in front of the measured one."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch, bench
warm, events, pre = int(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
dev = torch.device("cuda", 0)
if pre:
    bench.streaming_report(dev, hops=pre, warm=4)
r = bench.streaming_report(dev, hops=400, warm=warm, events=events)
wall = r["host_wall"]
print(json.dumps({"warm": warm, "events": events, "throwaway_hops": pre, "host_wall": wall, "d2h": r["d2h"], "h2d": r["h2d"],
                  "slowest": r["slowest_hops"][:2]}))
''' % ROOT
for warm, events, pre in ((20, 1, 0), (0, 1, 0), (200, 1, 0), (20, 0, 0), (200, 0, 0), (20, 1, 64)):
    out = subprocess.run([sys.executable, "-c", CHILD, str(warm), str(events), str(pre)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    for l in out.stdout.splitlines():
        if l.startswith("{"):
            j = json.loads(l)
            print("warm-up %3d  per-hop events %d  throwaway pass %2d hops | host wall max %.3f ms p99.9 %.3f | d2h max %.3f h2d max %.3f | slowest (hop, h2d, kernel, d2h): %s"
                  % (j["warm"], j["events"], j["throwaway_hops"], j["host_wall"]["max"], j["host_wall"]["p99_9"], j["d2h"]["max"], j["h2d"]["max"],
                     [(s["hop"], round(s["h2d"], 3), round(s["kernel"], 3), round(s["d2h"], 3)) for s in j["slowest"]]), flush=True)
