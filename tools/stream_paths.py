import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
bench.streaming_report(dev, hops=64, warm=4)
def show(tag, r):
    print(tag, {k: round(r[k]["p50"], 4) for k in ("device_total", "h2d", "kernel", "d2h", "host_wall")}, "max", round(r["device_total"]["max"], 3), flush=True)
for lf in (0, 256):
    for pcm in ("f32", "s16"):
        for op in (False, True):
            show("lf=%d %s out_pinned=%s direct-in" % (lf, pcm, op), bench.streaming_report(dev, hops=1000, logfreq_cols=lf, pcm=pcm, out_pinned=op))
os.environ["SG_NO_PINNED_DIRECT"] = "1"
for lf in (0, 256):
    show("lf=%d f32 copy-engine in, DMA out" % lf, bench.streaming_report(dev, hops=1000, logfreq_cols=lf))
