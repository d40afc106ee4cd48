"""CPU: bench.py's workload constants are the ones BASELINE.json / SURVEY.md section 8(d) fix."""
import importlib.util
import json
import os

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_workload_is_config_2():
    b = _load_bench()
    assert (b.FFT, b.HOP, b.N_SAMPLES) == (1024, 256, 28_800_000)   # 10 min mono 48 kHz
    assert b.FRAMES == 112_497                                       # floor((n - N)/hop) + 1
    assert b.BYTES_PER_FRAME == 3072                                 # hop*4 + N/2*4
    assert b.HBM_PEAK_GBPS == 8000.0
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "1024-pt, hop 256" in base["metric"] and base["published"] == {}


def test_traffic_file_shape():
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return
    t = json.load(open(p))["power"]
    assert t["algorithmic_bytes_per_launch"] == 112_497 * 3072
    # PMC traffic may exceed the algorithmic bytes (halo re-reads) but not wildly
    assert 1.0 <= t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"] < 1.25
    u8 = json.load(open(p)).get("db_u8")      # bench.py --out-kind db_u8 reads its roofline.traffic here (it was null)
    if u8:
        assert u8["algorithmic_bytes_per_launch"] == 112_497 * 1536
        assert 1.0 <= u8["hbm_bytes_per_launch"] / u8["algorithmic_bytes_per_launch"] < 1.25


def test_config5_constants():
    b = _load_bench()
    assert (b.UTT_SAMPLES, b.UTT_FRAMES) == (160_000, 622)            # 10 s at 16 kHz, N=1024, hop 256
    assert 1_000_000 * b.UTT_FRAMES == 622_000_000                     # 6.22e8 frames


def test_gpus_without_launcher_spawns_fresh_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE must start N child processes (one per GPU, rendezvous on
    127.0.0.1) from a parent that never touched the GPU, instead of asserting.  The children are
    replaced here by a stub interpreter that records what it was given."""
    import subprocess
    import sys
    b = _load_bench()
    seen = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
            self.returncode = 0
            self._rank0 = env["RANK"] == "0"

        def communicate(self):
            return ('{"rank0": true}\n' if self._rank0 else ""), None

        def wait(self):
            return 0

    monkeypatch.setattr(b.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    ns = type("A", (), {"gpus": 4})()
    rc = b.spawn_ranks(ns, ["--gpus", "4", "--steps", "3"])
    assert rc == 0 and len(seen) == 4
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2", "3"] == [e["LOCAL_RANK"] for _, e in seen]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in seen)
    assert len({e["MASTER_PORT"] for _, e in seen}) == 1
    assert all(cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"]
               for cmd, _ in seen)
    # and the parent path is taken before torch is imported: main() with --gpus 2 and no WORLD_SIZE exits via spawn_ranks
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('sys.exit(spawn_ranks(args, sys.argv[1:]))') < src.index("    import torch\n    import torch.distributed as dist")


def test_timed_region_excludes_the_closing_collective():
    """the closing barrier (an RCCL all-reduce at N > 1) must not sit inside the interval `value` is
    computed from.  A barrier that takes 50 ms must leave the measured time of a 10 ms body at ~10 ms, the order of
    calls must be barrier, body, synchronize, (stamp), barrier, max-reduce, and one rank keeps its meaning."""
    import time
    b = _load_bench()
    calls = []

    class FakeCuda:
        @staticmethod
        def synchronize():
            calls.append("sync")

    fake_torch = type("T", (), {"cuda": FakeCuda})

    def barrier():
        calls.append("barrier")
        time.sleep(0.05)

    def body():
        calls.append("body")
        time.sleep(0.01)

    def max_over_ranks(dt):
        calls.append("max")
        return dt

    dt = b.timed_region(body, barrier, max_over_ranks, fake_torch)
    assert calls == ["barrier", "body", "sync", "barrier", "max"]
    assert 0.009 <= dt < 0.04, dt   # the 50 ms barriers on either side are not billed
    src = open(os.path.join(ROOT, "bench.py")).read()
    fn = src[src.index("def timed_region"):src.index("def spawn_ranks")]
    assert fn.index("t0 = time.perf_counter()") < fn.index("body()") < fn.index("torch.cuda.synchronize()") \
        < fn.index("dt = time.perf_counter() - t0") < fn.rindex("barrier()")
    # every timed interval in the file goes through it: no other perf_counter() pair brackets a barrier()
    assert src.count("timed_region(") >= 4
