#!/usr/bin/env python3
"""bench.py -- STFT frames/s on MI355X for BASELINE.json's metric.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (window -> framing -> FFT -> |X|^2) over
BASELINE.json configs[1]: 10 min of mono 48 kHz float32 PCM, N=1024, hop=256 =
112,497 frames, input already resident in HBM.  With N ranks every rank owns its
own 10-minute stream (independent shards, no collective on the data path:
"scaling": "weak"); value = frames all ranks produced / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline     -- dominant kernel vs the HBM roofline: algorithmic bytes per launch
                  (hop*4 in + N/2*4 out = 3,072 B/frame x frames/launch) / average
                  launch duration from HIP events on the launch stream.
  cpu_baseline -- the oracle's float32 radix-2 restatement timed on this box's host
                  cores on the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT, HOP, SR, SECONDS = 1024, 256, 48000, 600
N_SAMPLES = SR * SECONDS                      # 28,800,000
FRAMES = (N_SAMPLES - FFT) // HOP + 1         # 112,497
BINS = FFT // 2
BYTES_PER_FRAME = HOP * 4 + BINS * 4          # 3,072
HBM_PEAK_GBPS = 8000.0                        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SETS = 3                                    # rotated in/out buffer sets (> 256 MiB Infinity Cache)


def host_cores():
    cores = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        cores = len(os.sched_getaffinity(0))
    try:  # container CPU share, if the cgroup states one
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return min(cores, 64)


def js_baseline(cores, budget_s=8.0):
    """The plain-JS radix-2 STFT (oracle/stft_oracle.js) under node worker_threads: the "reference
    JS path" of BASELINE.md section 4.  None when node is not installed."""
    import shutil
    import subprocess
    node = shutil.which("node")
    if node is None:
        return None
    try:
        out = subprocess.check_output([node, os.path.join(ROOT, "oracle", "js_baseline.js"), str(cores),
                                       str(budget_s), str(FFT), str(HOP), str(N_SAMPLES)], text=True, timeout=120)
        j = json.loads(out.strip().splitlines()[-1])
        return {"value": j["frames_per_s"], "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": "plain-JS Float32Array radix-2 STFT (oracle/stft_oracle.js) under node %s, %d worker_threads, "
                          "%d frames of the 10-min stream shape in %.1f s" %
                          (subprocess.check_output([node, "--version"], text=True).strip(), cores, j["frames"], j["seconds"])}
    except Exception as e:  # the baseline is informational; never fail the bench on it
        return {"value": None, "unit": "frames/s", "cores": cores, "kind": "port", "sample": "node failed: %r" % (e,)}


def cpu_baseline(budget_s=8.0):
    """CPU baselines on this box's host cores, same workload shape (N=1024, hop=256, 28.8 M samples):
    the JS path (headline, the reference's language) and the oracle's float32 C twin."""
    import numpy as np
    import oracle as O
    cores = host_cores()
    rng = np.random.default_rng(0x5EED0002)
    x = rng.uniform(-1, 1, N_SAMPLES).astype(np.float32)
    cfg = O.OracleConfig.make(FFT, HOP)
    O.lib()
    # each thread takes a contiguous range of frames of the one stream (halo = N - hop samples)
    per = (FRAMES + cores - 1) // cores
    done = [0] * cores
    stop_at = time.perf_counter() + budget_s
    outs = [np.empty((per, BINS), np.float32) for _ in range(cores)]

    def work(i):
        f0 = i * per
        nf = min(per, FRAMES - f0)
        if nf <= 0:
            return
        seg = x[f0 * HOP: f0 * HOP + (nf - 1) * HOP + FFT]
        while True:
            O.stft_f32(cfg, seg, out=outs[i][:nf])
            done[i] += nf
            if time.perf_counter() >= stop_at:
                return

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    c_port = {"value": sum(done) / dt, "unit": "frames/s", "cores": cores, "kind": "port",
              "sample": "oracle float32 radix-2 C port (gcc -O2), %d threads, %d passes over the 10-min stream "
                        "(%d frames) in %.1f s" % (cores, min(done) // max(per, 1), sum(done), dt)}
    js = js_baseline(cores, budget_s)
    if js is None or js["value"] is None:
        c_port["note"] = "node not available: scalar C port stands in for the JS path"
        return c_port
    js["c_port"] = {k: c_port[k] for k in ("value", "cores", "sample")}
    one = js_baseline(1, 3.0)  # the same JS path on one thread, for the per-core figure
    if one and one.get("value"):
        js["single_thread_value"] = one["value"]
    return js


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out-kind", default="power", choices=["power", "mag", "db", "db_u8"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("SG_BENCH_FORCE_DIST") == "1"  # the latter: 1-GPU rehearsal
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL: used only for the barrier + max
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    import spectrogram_amd as S
    eng = S.Engine(fft_size=FFT, hop=HOP, out=args.out_kind, device_id=local)
    out_dtype = torch.uint8 if args.out_kind == "db_u8" else torch.float32
    bytes_per_frame = HOP * 4 + BINS * (1 if args.out_kind == "db_u8" else 4)

    # synthetic PCM generated on device: uniform(-1, 1), seed per (config, rank, set)
    ins, outs = [], []
    for s in range(N_SETS):
        g = torch.Generator(device=dev).manual_seed(0x5EED0002 + 1000 * rank + s)
        ins.append((torch.rand(1, N_SAMPLES, device=dev, generator=g) * 2 - 1).contiguous())
        outs.append(torch.empty(1, FRAMES, BINS, device=dev, dtype=out_dtype))
    stream = torch.cuda.current_stream(dev)

    # measured copy rate on this box (the practical 100 % mark; reported beside the spec peak)
    a = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.float32)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        b.copy_(a)
    e1.record(stream)
    torch.cuda.synchronize()
    copy_gbps = 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b

    def step(i):
        eng.stft(ins[i % N_SETS], out=outs[i % N_SETS])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # timed region: EXACTLY K steps, barrier + synchronize on both sides.  One HIP-event pair on the
    # launch stream brackets the K launches: average launch duration = elapsed / K, launch boundaries
    # included.  (Marker packets around every launch, or around a sample of them, themselves open
    # gaps between ~80 us kernels and read 85-92 us)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(i)
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    avg_ms = ev0.elapsed_time(ev1) / args.steps

    # after the region: individually bracketed launches on an otherwise idle stream = the kernel's own
    # duration, the quantity rocprofv3's kernel trace reports (informational, not part of `value`)
    kern_ms = []
    for i in range(min(20, args.steps)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        step(i)
        b.record(stream)
        torch.cuda.synchronize()
        kern_ms.append(a.elapsed_time(b))
    kern_ms.sort()

    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        frames_total = FRAMES * args.steps * world
        achieved = FRAMES * bytes_per_frame / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")  # written from rocprofv3 --pmc passes
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.out_kind, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "STFT frames/sec (1024-pt, hop 256)",
            "value": frames_total / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "10 min mono 48 kHz PCM, 1024-pt FFT, hop 256, float32 (BASELINE.json configs[1])",
                       "fft_size": FFT, "hop": HOP, "window": "hann", "out": args.out_kind,
                       "frames_per_step_per_gpu": FRAMES, "bins": BINS,
                       "sharding": "one independent 10-min stream per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "sg::stft_kernel<10,2,%d>" % ["power", "mag", "db", "db_u8"].index(args.out_kind),
                         "bytes_per_frame": bytes_per_frame, "frames_per_launch": FRAMES,
                         "avg_launch_ms": avg_ms, "isolated_launch_ms_median": kern_ms[len(kern_ms) // 2],
                         "isolated_launch_ms_min": kern_ms[0],
                         "measured_copy_GBps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
